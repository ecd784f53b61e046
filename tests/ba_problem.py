"""Synthetic bundle-adjustment problems shared by the CPU (oracle) and GPU (kernel) tests."""
import numpy as np


def rot(axis, ang):
    axis = np.asarray(axis, float) / np.linalg.norm(axis)
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * K @ K


def make_problem(N=5, K=8, seed=0, noise_px=0.0, outlier_frac=0.0, perturb=0.0, W=406, H=308):
    """Ground-truth cameras on an arc looking at a point cloud; dense observations with the pipeline's visibility
    pattern (a track of frame s is seen by every earlier frame and the next two).  Returns the PERTURBED start values
    (R, C, X), intrinsics, uv, valid and the ground truth."""
    rng = np.random.default_rng(seed)
    R_gt = np.stack([rot([0, 1, 0], 0.06 * t) @ rot([1, 0, 0], 0.02 * t) for t in range(N)])      # world -> camera
    C_gt = np.stack([[0.25 * t, 0.02 * t, 0.05 * t] for t in range(N)]).astype(float)
    intr = np.tile(np.array([[320.0, 330.0, W / 2.0, H / 2.0]]), (N, 1)) * (1 + 0.01 * rng.standard_normal((N, 1)))
    # like the pipeline: a track is a keypoint pixel of its own frame lifted along its ray (always visible there)
    X_gt = np.zeros((N * K, 3))
    for s in range(N):
        px = np.stack([rng.uniform(20, W - 20, K), rng.uniform(20, H - 20, K)], 1)
        d = rng.uniform(3.0, 7.0, K)
        ray = np.stack([(px[:, 0] - intr[s, 2]) / intr[s, 0], (px[:, 1] - intr[s, 3]) / intr[s, 1], np.ones(K)], 1)
        X_gt[s * K:(s + 1) * K] = C_gt[s] + (ray * d[:, None]) @ R_gt[s]
    uv = np.zeros((N, N, K, 2), np.float32)
    valid = np.zeros((N, N, K), np.uint8)
    for s in range(N):
        for t in range(N):
            if not (t <= s or t <= s + 2):
                continue
            for k in range(K):
                p = R_gt[t] @ (X_gt[s * K + k] - C_gt[t])
                if p[2] <= 0.1:
                    continue
                u = intr[t, 0] * p[0] / p[2] + intr[t, 2]
                v = intr[t, 1] * p[1] / p[2] + intr[t, 3]
                if 0 <= u < W and 0 <= v < H:
                    uv[s, t, k] = (u, v)
                    valid[s, t, k] = 1
    if noise_px > 0:
        uv += (noise_px * rng.standard_normal(uv.shape)).astype(np.float32) * valid[..., None]
    if outlier_frac > 0:
        bad = (rng.random(valid.shape) < outlier_frac) & (valid > 0)
        uv[bad] += rng.uniform(-40, 40, size=(int(bad.sum()), 2)).astype(np.float32)
    R0 = np.stack([rot(rng.standard_normal(3), perturb * 0.02 * rng.standard_normal()) @ R_gt[t] for t in range(N)])
    C0 = C_gt + perturb * 0.03 * rng.standard_normal(C_gt.shape)
    X0 = X_gt + perturb * 0.05 * rng.standard_normal(X_gt.shape)
    return dict(R=R0, C=C0, X=X0, intr=intr, uv=uv, valid=valid, R_gt=R_gt, C_gt=C_gt, X_gt=X_gt, W=W, H=H)
