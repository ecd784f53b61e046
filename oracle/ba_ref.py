"""ORACLE — test infrastructure only.  CPU (numpy, float64) restatement of the bundle adjustment the reference runs
through pytheia / Ceres: per chunk (utils/chunk_reconstruction.py:188-219: 10 iterations, Huber 2.0, DENSE_SCHUR,
SetOutlierTracksToUnestimated(tracks, 2, 0.25)) and after each alignment with pose priors on the overlap views
(utils/reconstruction_alignment.py:107-171: orientation prior covariance 2 I, position prior covariance 25 I,
50 iterations, Huber 3.0, outliers (3, 0.25)).

PARITY UNPINNED: pytheia 0.2.9 (C++ TheiaSfM + Ceres Solver, requirements.txt:10) is not under /root/reference and
cannot be installed offline, and the reference holds no vectors for this stage.  What is restated is the published
method those libraries implement:
  * reprojection residual of a pinhole camera with fixed intrinsics (Theia's default intrinsics_to_optimize = NONE),
    camera = (world->camera rotation R, centre C), r = (fx x/z + cx - u, fy y/z + cy - v), (x, y, z) = R (X - C);
  * Huber loss rho(s) = s (s <= a^2), 2 a sqrt(s) - a^2 otherwise, applied per observation as iteratively re-weighted
    least squares (weight rho'(s));
  * Levenberg-Marquardt as Ceres' trust-region strategy states it: (J^T W J + diag(J^T W J) / radius) d = -J^T W r with
    the diagonal clamped to [1e-6, 1e32]; step quality rho = (cost - cost_new) / model_decrease; accepted when
    rho > 1e-3 with radius /= max(1/3, 1 - (2 rho - 1)^3), otherwise radius /= decrease_factor (2, doubling);
    function tolerance 1e-6; initial radius 1e4;
  * the Schur complement on the points (DENSE_SCHUR), which is algebra, not an approximation: this file solves the full
    normal equations densely, the device eliminates the points first - both must produce the same step.
`solve_with_scipy` is an independent check of the OPTIMUM (scipy.optimize.least_squares with loss='huber').
`bundle_adjust_schur` is the same LM loop with the step computed the way DENSE_SCHUR does (per-track 3x3 elimination,
reduced 6N x 6N camera system by Cholesky, back-substitution): it scales to a whole chunk (100 cameras, 20 000 tracks,
~1 M observations: seconds per iteration) where the dense normal equations (60 600^2 doubles) do not, and
tests/test_ba_oracle.py checks that both forms take the same steps.
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import numpy as np


def huber(s: np.ndarray, a: float) -> Tuple[np.ndarray, np.ndarray]:
    small = s <= a * a
    r = np.sqrt(np.where(small, 1.0, s))
    return np.where(small, s, 2 * a * r - a * a), np.where(small, 1.0, a / r)


def skew(p: np.ndarray) -> np.ndarray:
    z = np.zeros(p.shape[:-1])
    return np.stack([np.stack([z, -p[..., 2], p[..., 1]], -1), np.stack([p[..., 2], z, -p[..., 0]], -1),
                     np.stack([-p[..., 1], p[..., 0], z], -1)], -2)


def exp_so3(w: np.ndarray) -> np.ndarray:
    th = np.linalg.norm(w)
    K = skew(w)
    if th < 1e-8:
        a, b = 1.0 - th * th / 6.0, 0.5 - th * th / 24.0
    else:
        a, b = np.sin(th) / th, (1 - np.cos(th)) / th ** 2
    return np.eye(3) + a * K + b * K @ K


def log_so3(R: np.ndarray) -> np.ndarray:
    c = np.clip(0.5 * (np.trace(R) - 1.0), -1.0, 1.0)
    th = np.arccos(c)
    v = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    return (0.5 if th < 1e-8 else th / (2 * np.sin(th))) * v


def observations(uv: np.ndarray, valid: np.ndarray):
    """Dense [N][N][K] layout -> lists: track index i = s*K + k, camera t, pixel."""
    s, t, k = np.nonzero(valid)
    K = valid.shape[2]
    return s * K + k, t, uv[s, t, k].astype(np.float64)


def residuals(R, C, intr, X, trk, cam, px):
    """-> r (M,2), Jc (M,2,6), Jp (M,2,3), front (M,) bool."""
    d = X[trk] - C[cam]
    p = np.einsum("mij,mj->mi", R[cam], d)
    front = p[:, 2] > 1e-9
    z = np.where(front, p[:, 2], 1.0)
    fx, fy, cx, cy = (intr[cam, i] for i in range(4))
    r = np.stack([fx * p[:, 0] / z + cx - px[:, 0], fy * p[:, 1] / z + cy - px[:, 1]], -1)
    Jpi = np.zeros((len(trk), 2, 3))
    Jpi[:, 0, 0], Jpi[:, 0, 2] = fx / z, -fx * p[:, 0] / z ** 2
    Jpi[:, 1, 1], Jpi[:, 1, 2] = fy / z, -fy * p[:, 1] / z ** 2
    Jp = np.einsum("mij,mjk->mik", Jpi, R[cam])
    Jw = -np.einsum("mij,mjk->mik", Jpi, skew(p))
    Jc = np.concatenate([Jw, -Jp], axis=2)
    return r, Jc, Jp, front


def prior_terms(R, C, prior):
    """-> (cost, H_diag (N,6), g (N,6)) of the pose priors (first-order Jacobian sr I / sp I)."""
    N = len(R)
    Hd, g, cost = np.zeros((N, 6)), np.zeros((N, 6)), 0.0
    if prior is None:
        return cost, Hd, g
    sr2, sp2 = prior["sqrt_info_rot"] ** 2, prior["sqrt_info_pos"] ** 2
    for t in np.nonzero(prior["flag"])[0]:
        w3 = log_so3(R[t] @ prior["R"][t].T)
        dc = C[t] - prior["C"][t]
        Hd[t, :3], Hd[t, 3:] = sr2, sp2
        g[t, :3], g[t, 3:] = sr2 * w3, sp2 * dc
        cost += 0.5 * sr2 * w3 @ w3 + 0.5 * sp2 * dc @ dc
    return cost, Hd, g


def total_cost(R, C, intr, X, trk, cam, px, a, prior) -> float:
    r, _, _, front = residuals(R, C, intr, X, trk, cam, px)
    rho, _ = huber((r ** 2).sum(1), a)
    return 0.5 * rho[front].sum() + prior_terms(R, C, prior)[0]


def householder(x: np.ndarray):
    """(v, beta) with (I - beta v v^T) x = +-|x| e_last: ceres::internal::ComputeHouseholderVector, the basis of
    ceres::HomogeneousVectorParameterization (what Theia's use_homogeneous_point_parametrization puts on a track's
    4-vector; the reference sets that option to True: utils/reconstruction_alignment.py:147-152)."""
    sigma = float(x[:-1] @ x[:-1])
    v = x.astype(np.float64).copy()
    v[-1] = 1.0
    pivot = float(x[-1])
    if sigma <= 1e-300:
        return v, (2.0 if pivot < 0.0 else 0.0)
    mu = np.sqrt(pivot * pivot + sigma)
    vp = pivot - mu if pivot <= 0.0 else -sigma / (pivot + mu)
    beta = 2.0 * vp * vp / (sigma + vp * vp)
    v[:-1] /= vp
    return v, beta


def homogeneous_plus(h: np.ndarray, delta: np.ndarray) -> np.ndarray:
    """HomogeneousVectorParameterization::Plus: move the 4-vector h along its unit sphere by the tangent step delta (3)."""
    nd = float(np.linalg.norm(delta))
    if nd == 0.0:
        return h.copy()
    y = np.concatenate([delta * (np.sin(0.5 * nd) / nd), [np.cos(0.5 * nd)]])
    v, beta = householder(h)
    return np.linalg.norm(h) * (y - v * (beta * (v @ y)))


def homogeneous_plus_jacobian(h: np.ndarray) -> np.ndarray:
    """d Plus(h, delta) / d delta at delta = 0: 0.5 |h| (I - beta v v^T)[:, :3]  (4 x 3)."""
    v, beta = householder(h)
    Hh = np.eye(4) - beta * np.outer(v, v)
    return 0.5 * np.linalg.norm(h) * Hh[:, :3]


def point_frames(X: np.ndarray) -> np.ndarray:
    """T (P, 3, 3) = d X / d delta at delta = 0 for every track, h = [X, 1] / |[X, 1]| RE-DERIVED from X (the device does
    the same each iteration; identical to carrying h while its last component stays positive, which the tests check):
    T = (d X / d h) (d Plus / d delta) = (1 / w) [I | -X] . 0.5 (I - beta v v^T)[:, :3]."""
    P = len(X)
    h = np.concatenate([X, np.ones((P, 1))], 1)
    h /= np.linalg.norm(h, axis=1, keepdims=True)
    T = np.zeros((P, 3, 3))
    for i in range(P):
        T[i] = (np.concatenate([np.eye(3), -X[i][:, None]], 1) / h[i, 3]) @ homogeneous_plus_jacobian(h[i])
    return T


def points_plus(X: np.ndarray, delta: np.ndarray) -> np.ndarray:
    """X after the tangent steps delta (P, 3) on the spheres of h = [X, 1] / |[X, 1]|."""
    h = np.concatenate([X, np.ones((len(X), 1))], 1)
    h /= np.linalg.norm(h, axis=1, keepdims=True)
    hn = np.stack([homogeneous_plus(h[i], delta[i]) for i in range(len(X))])
    return hn[:, :3] / hn[:, 3:4]


def bundle_adjust(R, C, intr, X, uv, valid, huber_width: float, max_iters: int, prior: Optional[Dict] = None,
                  homogeneous: bool = False, carry_h: bool = False):
    """R (N,3,3) world->camera, C (N,3), intr (N,4), X (N*K,3); uv (N,N,K,2), valid (N,N,K).
    Returns refined (R, C, X) and a summary dict.  prior: {'R','C','flag','sqrt_info_rot','sqrt_info_pos'}.
    homogeneous=True: every track is the unit 4-vector h = [X, 1] / |[X, 1]| stepped in the 3-dimensional tangent space
    of its sphere (Theia's default point parametrization, see householder()); the objective is the same function of the
    same geometry - X = h[:3] / h[3] - so the optimum is the same and only the LM path (its diagonal damping acts on
    different coordinates) differs.  The device adjuster optimises Euclidean points; tests/test_ba_oracle.py measures
    what that changes after the reference's 10 / 50 iterations."""
    R, C, X = R.copy(), C.copy(), X.copy()
    N, P = len(R), len(X)
    hvec = None              # carry_h: keep the 4-vectors from iteration to iteration as Ceres does (default: re-derive)
    if homogeneous and carry_h:
        hvec = np.concatenate([X, np.ones((P, 1))], 1)
        hvec /= np.linalg.norm(hvec, axis=1, keepdims=True)
    trk, cam, px = observations(uv, valid)
    cost = total_cost(R, C, intr, X, trk, cam, px, huber_width, prior)
    summary = {"initial_cost": cost, "iterations": 0, "accepted_steps": 0}
    radius, decrease = 1e4, 2.0
    nc = 6 * N
    for _ in range(max_iters):
        r, Jc, Jp, front = residuals(R, C, intr, X, trk, cam, px)
        if homogeneous:      # d r / d delta = (d r / d X) (d X / d h) (d h / d delta),  X = h[:3] / h[3]
            if carry_h:
                T = np.zeros((P, 3, 3))
                for i in range(P):
                    dX_dh = np.concatenate([np.eye(3), -X[i][:, None]], 1) / hvec[i, 3]
                    T[i] = dX_dh @ homogeneous_plus_jacobian(hvec[i])
            else:
                T = point_frames(X)
            Jp = np.einsum("mij,mjk->mik", Jp, T[trk])
        _, w = huber((r ** 2).sum(1), huber_width)
        w = w * front
        H = np.zeros((nc + 3 * P, nc + 3 * P))
        g = np.zeros(nc + 3 * P)
        ci = (6 * cam[:, None] + np.arange(6)[None]).astype(int)
        pi = (nc + 3 * trk[:, None] + np.arange(3)[None]).astype(int)
        J = np.zeros((len(trk), 2, 9))
        J[:, :, :6], J[:, :, 6:] = Jc, Jp
        idx = np.concatenate([ci, pi], 1)
        blocks = np.einsum("m,mia,mib->mab", w, J, J)
        np.add.at(H, (idx[:, :, None], idx[:, None, :]), blocks)
        np.add.at(g, idx, np.einsum("m,mia,mi->ma", w, J, r))
        _, Hd, gprior = prior_terms(R, C, prior)
        H[np.arange(nc), np.arange(nc)] += Hd.reshape(-1)
        g[:nc] += gprior.reshape(-1)
        D = np.clip(np.diag(H), 1e-6, 1e32) / radius
        ok = True
        try:
            L = np.linalg.cholesky(H + np.diag(D))
            d = -np.linalg.solve(L.T, np.linalg.solve(L, g))
        except np.linalg.LinAlgError:
            ok, d = False, np.zeros_like(g)
        model = -0.5 * g @ d + 0.5 * d @ (D * d)
        Rn = np.stack([exp_so3(d[6 * t:6 * t + 3]) @ R[t] for t in range(N)])
        Cn = C + d[:nc].reshape(N, 6)[:, 3:]
        if homogeneous and carry_h:
            hn = np.stack([homogeneous_plus(hvec[i], d[nc + 3 * i: nc + 3 * i + 3]) for i in range(P)])
            Xn = hn[:, :3] / hn[:, 3:4]
        elif homogeneous:
            Xn = points_plus(X, d[nc:].reshape(P, 3))
        else:
            Xn = X + d[nc:].reshape(P, 3)
        cnew = total_cost(Rn, Cn, intr, Xn, trk, cam, px, huber_width, prior)
        summary["iterations"] += 1
        rho = (cost - cnew) / model if (ok and model > 0) else -1.0
        if rho > 1e-3 and np.isfinite(cnew):
            radius = min(radius / max(1.0 / 3.0, 1.0 - (2 * rho - 1) ** 3), 1e16)
            decrease = 2.0
            rel = abs(cost - cnew) / max(cost, 1e-300)
            R, C, X, cost = Rn, Cn, Xn, cnew
            if homogeneous and carry_h:
                hvec = hn
            summary["accepted_steps"] += 1
            if rel < 1e-6:
                break
        else:
            radius /= decrease
            decrease *= 2.0
            if radius < 1e-32:
                break
    summary["final_cost"] = cost
    summary["radius"] = radius
    return R, C, X, summary


# ------------------------------------------------------------------------------------------------ inverse depth
# The reference's --use-inverse-depth (utils/chunk_reconstruction.py:187-188,199-204; utils/reconstruction_alignment.py:
# 147-152): reconstruction.InitializeInverseDepth() + ba_options.use_inverse_depth_parametrization = True.  TheiaSfM then
# holds, per track, ONE parameter - the inverse depth rho along the bearing of the track's feature in its REFERENCE view
# (the first view that observed it; a track of this pipeline is created with its own frame's keypoint first,
# chunk_reconstruction.py:142-160, so the reference view of track (s, k) is frame s):
#     X = C_s + R_s^T (b / rho),   b = ((u - cx) / fx, (v - cy) / fy, 1) of the keypoint in frame s,
# and every OTHER observation's reprojection error is a function of (pose_s, pose_t, rho); the reference view's own
# observation is met exactly by construction and carries no residual.  PARITY UNPINNED like the rest of this file.
def inverse_depth_state(R, C, intr, X, uv, valid):
    """(b (P, 3), rho (P,), anchor (P,)) from Euclidean points: InitializeInverseDepth - the depth of every track in its
    reference view; the point is thereby snapped onto the reference keypoint's ray."""
    N, _, K = valid.shape
    P = N * K
    anchor = np.arange(P) // K
    kk = np.arange(P) % K
    px = uv[anchor, anchor, kk].astype(np.float64)
    ia = intr[anchor]
    b = np.stack([(px[:, 0] - ia[:, 2]) / ia[:, 0], (px[:, 1] - ia[:, 3]) / ia[:, 1], np.ones(P)], 1)
    z = np.einsum("pj,pj->p", R[anchor][:, 2, :], X - C[anchor])
    return b, 1.0 / z, anchor


def inverse_depth_points(R, C, b, rho, anchor):
    return C[anchor] + np.einsum("pji,pj->pi", R[anchor], b / rho[:, None])


def invdepth_observations(uv, valid):
    """observations() without every track's reference-view observation."""
    trk, cam, px = observations(uv, valid)
    K = valid.shape[2]
    keep = cam != trk // K
    return trk[keep], cam[keep], px[keep]


def invdepth_jacobians(R, C, intr, b, rho, anchor, trk, cam, px):
    """-> r (M,2), Jt (M,2,6) w.r.t. the observing camera, Ja (M,2,6) w.r.t. the reference camera, Jr (M,2) w.r.t. rho,
    front.  Chain rule through X(pose_a, rho): dX/dC_a = I, dX/dw_a = R_a^T [p_a]x (R_a <- exp([w]x) R_a), dX/drho =
    -R_a^T p_a / rho with p_a = b / rho."""
    X = inverse_depth_points(R, C, b, rho, anchor)
    r, Jt, Jp, front = residuals(R, C, intr, X, trk, cam, px)
    a = anchor[trk]
    pa = b[trk] / rho[trk][:, None]
    RaT = np.transpose(R[a], (0, 2, 1))
    dX_dw = np.einsum("mij,mjk->mik", RaT, skew(pa))
    Ja = np.concatenate([np.einsum("mij,mjk->mik", Jp, dX_dw), Jp], axis=2)
    Jr = -np.einsum("mij,mj->mi", Jp, np.einsum("mij,mj->mi", RaT, pa)) / rho[trk][:, None]
    return r, Jt, Ja, Jr, front


def invdepth_cost(R, C, intr, b, rho, anchor, trk, cam, px, a, prior) -> float:
    X = inverse_depth_points(R, C, b, rho, anchor)
    r, _, _, front = residuals(R, C, intr, X, trk, cam, px)
    rh, _ = huber((r ** 2).sum(1), a)
    return 0.5 * rh[front].sum() + prior_terms(R, C, prior)[0]


def bundle_adjust_inverse_depth(R, C, intr, X, uv, valid, huber_width: float, max_iters: int,
                                prior: Optional[Dict] = None):
    """bundle_adjust() with one inverse-depth parameter per track (dense normal equations over 6 N + P unknowns, the
    same LM rules).  Returns refined (R, C, X, summary); X are the Euclidean points of the refined state."""
    R, C = R.copy(), C.copy()
    N = len(R)
    b, rho, anchor = inverse_depth_state(R, C, intr, X, uv, valid)
    P = len(rho)
    trk, cam, px = invdepth_observations(uv, valid)
    cost = invdepth_cost(R, C, intr, b, rho, anchor, trk, cam, px, huber_width, prior)
    summary = {"initial_cost": cost, "iterations": 0, "accepted_steps": 0}
    radius, decrease = 1e4, 2.0
    nc = 6 * N
    for _ in range(max_iters):
        r, Jt, Ja, Jr, front = invdepth_jacobians(R, C, intr, b, rho, anchor, trk, cam, px)
        _, w = huber((r ** 2).sum(1), huber_width)
        w = w * front
        a = anchor[trk]
        n = nc + P
        H, g = np.zeros((n, n)), np.zeros(n)
        it = (6 * cam[:, None] + np.arange(6)[None]).astype(int)
        ia = (6 * a[:, None] + np.arange(6)[None]).astype(int)
        ir = (nc + trk)[:, None].astype(int)
        J = np.concatenate([Jt, Ja, Jr[:, :, None]], 2)                  # (M, 2, 13)
        idx = np.concatenate([it, ia, ir], 1)
        np.add.at(H, (idx[:, :, None], idx[:, None, :]), np.einsum("m,mia,mib->mab", w, J, J))
        np.add.at(g, idx, np.einsum("m,mia,mi->ma", w, J, r))
        _, Hd, gprior = prior_terms(R, C, prior)
        H[np.arange(nc), np.arange(nc)] += Hd.reshape(-1)
        g[:nc] += gprior.reshape(-1)
        D = np.clip(np.diag(H), 1e-6, 1e32) / radius
        ok = True
        try:
            L = np.linalg.cholesky(H + np.diag(D))
            d = -np.linalg.solve(L.T, np.linalg.solve(L, g))
        except np.linalg.LinAlgError:
            ok, d = False, np.zeros_like(g)
        model = -0.5 * g @ d + 0.5 * d @ (D * d)
        Rn = np.stack([exp_so3(d[6 * t:6 * t + 3]) @ R[t] for t in range(N)])
        Cn = C + d[:nc].reshape(N, 6)[:, 3:]
        rn = rho + d[nc:]
        cnew = invdepth_cost(Rn, Cn, intr, b, rn, anchor, trk, cam, px, huber_width, prior)
        summary["iterations"] += 1
        q = (cost - cnew) / model if (ok and model > 0) else -1.0
        if q > 1e-3 and np.isfinite(cnew):
            radius = min(radius / max(1.0 / 3.0, 1.0 - (2 * q - 1) ** 3), 1e16)
            decrease = 2.0
            rel = abs(cost - cnew) / max(cost, 1e-300)
            R, C, rho, cost = Rn, Cn, rn, cnew
            summary["accepted_steps"] += 1
            if rel < 1e-6:
                break
        else:
            radius /= decrease
            decrease *= 2.0
            if radius < 1e-32:
                break
    summary["final_cost"] = cost
    summary["radius"] = radius
    return R, C, inverse_depth_points(R, C, b, rho, anchor), summary


def _segsum(idx: np.ndarray, vals: np.ndarray, n: int) -> np.ndarray:
    """out[j] = sum of vals[m] over idx[m] == j, in observation order (np.bincount adds sequentially), any trailing
    shape; the fixed order makes the oracle itself reproducible."""
    flat = vals.reshape(len(idx), -1)
    out = np.stack([np.bincount(idx, weights=flat[:, c], minlength=n) for c in range(flat.shape[1])], 1)
    return out.reshape((n,) + vals.shape[1:])


def schur_step(R, C, intr, X, trk, cam, px, K, huber_width, radius, prior, homogeneous: bool = False):
    """One damped Gauss-Newton step through the Schur complement on the points.
    -> (ok, dc (N,6), dp (P,3), model_decrease).  ok = False: the reduced system was not positive definite."""
    N, P = len(R), len(X)
    r, Jc, Jp, front = residuals(R, C, intr, X, trk, cam, px)
    if homogeneous:          # the point columns in the tangent coordinates of the tracks' spheres; dp is then a tangent step
        Jp = np.einsum("mij,mjk->mik", Jp, point_frames(X)[trk])
    _, w = huber((r ** 2).sum(1), huber_width)
    w = w * front
    B = _segsum(cam, np.einsum("m,mia,mib->mab", w, Jc, Jc), N)            # camera blocks
    gc = _segsum(cam, np.einsum("m,mia,mi->ma", w, Jc, r), N)
    Cb = _segsum(trk, np.einsum("m,mia,mib->mab", w, Jp, Jp), P)           # point blocks
    gp = _segsum(trk, np.einsum("m,mia,mi->ma", w, Jp, r), P)
    E = np.einsum("m,mia,mib->mab", w, Jc, Jp)                             # (M, 6, 3) off-diagonal blocks
    _, Hd, gprior = prior_terms(R, C, prior)
    i6, i3 = np.arange(6), np.arange(3)
    B[:, i6, i6] += Hd
    gc = gc + gprior
    Dc = np.clip(B[:, i6, i6], 1e-6, 1e32) / radius
    Dp = np.clip(Cb[:, i3, i3], 1e-6, 1e32) / radius
    Cd = Cb.copy()
    Cd[:, i3, i3] += Dp
    Cinv = np.linalg.inv(Cd)
    n6 = 6 * N
    S = np.zeros((n6, n6))
    for t in range(N):
        S[6 * t:6 * t + 6, 6 * t:6 * t + 6] = B[t] + np.diag(Dc[t])
    v = np.zeros(n6)
    src, kk = trk // K, trk % K
    bounds = np.searchsorted(src, np.arange(N + 1))                        # observations are sorted by source frame
    for s_ in range(N):
        sl = slice(bounds[s_], bounds[s_ + 1])
        if sl.start == sl.stop:
            continue
        tmax = int(cam[sl].max()) + 1
        A = np.zeros((tmax, 6, K, 3))
        A[cam[sl], :, kk[sl], :] = E[sl]
        Y = np.einsum("takb,kbc->takc", A, Cinv[s_ * K:(s_ + 1) * K]).reshape(6 * tmax, 3 * K)
        S[:6 * tmax, :6 * tmax] -= Y @ A.reshape(6 * tmax, 3 * K).T
        v[:6 * tmax] += Y @ gp[s_ * K:(s_ + 1) * K].reshape(3 * K)
    try:
        L = np.linalg.cholesky(S)
        if not np.isfinite(L).all():          # LAPACK lets NaN pivots through; a pivot that is not > 0 is a failure
            raise np.linalg.LinAlgError("non-finite factor")
    except np.linalg.LinAlgError:
        return False, np.zeros((N, 6)), np.zeros((P, 3)), 0.0
    dc = np.linalg.solve(L.T, np.linalg.solve(L, -(gc.reshape(-1) - v))).reshape(N, 6)
    dp = -np.einsum("pab,pb->pa", Cinv, gp + _segsum(trk, np.einsum("mab,ma->mb", E, dc[cam]), P))
    model = -0.5 * ((gc * dc).sum() + (gp * dp).sum()) + 0.5 * ((Dc * dc * dc).sum() + (Dp * dp * dp).sum())
    return True, dc, dp, model


def bundle_adjust_schur(R, C, intr, X, uv, valid, huber_width: float, max_iters: int, prior: Optional[Dict] = None,
                        trace: Optional[list] = None, homogeneous: bool = False):
    """bundle_adjust() with the linear solve of every iteration done by schur_step; same trust-region rules, same
    summary.  trace (a list) receives one dict per iteration: cost before, candidate cost, accepted, radius, ok."""
    R, C, X = R.copy(), C.copy(), X.copy()
    N, P, K = len(R), len(X), valid.shape[2]
    trk, cam, px = observations(uv, valid)
    cost = total_cost(R, C, intr, X, trk, cam, px, huber_width, prior)
    summary = {"initial_cost": cost, "iterations": 0, "accepted_steps": 0, "chol_failures": 0}
    radius, decrease = 1e4, 2.0
    for _ in range(max_iters):
        ok, dc, dp, model = schur_step(R, C, intr, X, trk, cam, px, K, huber_width, radius, prior, homogeneous)
        Rn = np.stack([exp_so3(dc[t, :3]) @ R[t] for t in range(N)])
        Cn, Xn = C + dc[:, 3:], (points_plus(X, dp) if homogeneous else X + dp)
        cnew = total_cost(Rn, Cn, intr, Xn, trk, cam, px, huber_width, prior)
        summary["iterations"] += 1
        summary["chol_failures"] += 0 if ok else 1
        rho = (cost - cnew) / model if (ok and model > 0) else -1.0
        accepted = bool(rho > 1e-3 and np.isfinite(cnew))
        if trace is not None:
            trace.append(dict(cost=cost, cost_new=cnew, accepted=accepted, radius=radius, ok=ok, model=model))
        if accepted:
            radius = min(radius / max(1.0 / 3.0, 1.0 - (2 * rho - 1) ** 3), 1e16)
            decrease = 2.0
            rel = abs(cost - cnew) / max(cost, 1e-300)
            R, C, X, cost = Rn, Cn, Xn, cnew
            summary["accepted_steps"] += 1
            if rel < 1e-6:
                break
        else:
            radius /= decrease
            decrease *= 2.0
            if radius < 1e-32:
                break
    summary["final_cost"] = cost
    summary["radius"] = radius
    return R, C, X, summary


def outlier_tracks(R, C, intr, X, uv, valid, max_px: float, min_angle_deg: float) -> np.ndarray:
    """SetOutlierTracksToUnestimated semantics (TheiaSfM): a track stays estimated iff every observation is in front of
    its camera and within max_px, and some pair of viewing rays subtends more than min_angle_deg."""
    N, _, K = valid.shape
    est = np.zeros(N * K, dtype=bool)
    cosmin = np.cos(np.deg2rad(min_angle_deg))
    for s in range(N):
        for k in range(K):
            i = s * K + k
            cams = np.nonzero(valid[s, :, k])[0]
            if len(cams) == 0:
                continue
            trk = np.full(len(cams), i)
            r, _, _, front = residuals(R, C, intr, X, trk, cams, uv[s, cams, k].astype(np.float64))
            if not front.all() or ((r ** 2).sum(1) > max_px ** 2).any():
                continue
            rays = X[i] - C[cams]
            rays = rays / np.linalg.norm(rays, axis=1, keepdims=True)
            cs = rays @ rays.T
            est[i] = bool((cs[np.triu_indices(len(cams), 1)] < cosmin).any())
    return est


def solve_with_scipy(R, C, intr, X, uv, valid, huber_width: float, prior: Optional[Dict] = None, max_nfev: int = 200):
    """Independent optimiser for the same objective (scipy trust-region reflective, loss='huber' on the 2-vector
    residual norm implemented through per-observation re-weighting): returns the final cost only."""
    from scipy.optimize import minimize
    N, P = len(R), len(X)
    trk, cam, px = observations(uv, valid)

    def unpack(z):
        Rn = np.stack([exp_so3(z[6 * t:6 * t + 3]) @ R[t] for t in range(N)])
        Cn = C + z[:6 * N].reshape(N, 6)[:, 3:]
        return Rn, Cn, X + z[6 * N:].reshape(P, 3)

    def fun(z):
        Rn, Cn, Xn = unpack(z)
        return total_cost(Rn, Cn, intr, Xn, trk, cam, px, huber_width, prior)

    res = minimize(fun, np.zeros(6 * N + 3 * P), method="L-BFGS-B", options={"maxiter": 3000, "maxfun": 200000,
                                                                           "ftol": 1e-15, "gtol": 1e-10})
    return float(res.fun)
