"""Seeded random-shape fuzz of the post-network entry points against the oracle: the Sim(3) match +
filter + closed form, the keypoint gather / fp16 pack, the observation projection and the bundle adjuster, at shapes the
fixed tests do not visit (odd overlap counts, K not a multiple of anything, ragged matches, duplicate keypoints, masks,
weights, 1-pixel-wide maps, ...).  Every case is compared with oracle/ exactly as the fixed-shape tests compare: indices
and fp16 samples bit for bit, the similarity to 1e-11, bundle-adjustment iterates step for step.
FUZZ_SECONDS (default 40) bounds the wall time; the seed is fixed, so a failure reproduces."""
import os
import time

import numpy as np
import pytest
import torch

from ba_problem import make_problem

pytestmark = pytest.mark.gpu
BUDGET = float(os.environ.get("FUZZ_SECONDS", "40"))


@pytest.fixture(scope="module")
def dev(built_lib):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def test_fuzz_sim3_against_oracle(dev):
    from oracle import post_ref
    from pi3_slam_amd import ops
    rng = np.random.default_rng(1234)
    t_end, n = time.time() + BUDGET / 4, 0
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)            # noqa: E731
    while time.time() < t_end or n < 20:
        n += 1
        ov = int(rng.integers(1, 25))
        K = int(rng.choice([1, 2, 3, 7, 63, 64, 65, 200, 257, 400, 450]))
        world = rng.standard_normal((ov, K, 3)) * rng.uniform(0.5, 3.0) + rng.standard_normal(3) + [0, 0, 5.0]
        ang = rng.uniform(-1, 1)
        R = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]])
        s, tr = float(np.exp(rng.uniform(-0.5, 0.5))), rng.standard_normal(3)
        qry = ((world - tr) @ R) / s + rng.choice([0.0, 1e-3, 5e-2]) * rng.standard_normal(world.shape)
        kr = (rng.random((ov, K, 2)) * 400).astype(np.float16)
        kq = kr.copy()
        pq = qry
        mode = rng.integers(0, 4)
        if mode >= 1 and K > 1:                                               # permuted qry keypoints
            perm = np.stack([rng.permutation(K) for _ in range(ov)])
            kq = np.take_along_axis(kq, perm[..., None], 1)
            pq = np.take_along_axis(pq, perm[..., None], 1)
        if mode >= 2:                                                          # unmatched + duplicated keypoints
            drop = rng.random((ov, K)) < 0.2
            kq[drop] = (kq[drop].astype(np.float32) + 1000).astype(np.float16)
            if K > 4:
                kr[:, K // 2] = kr[:, 1]
        dtype = np.float16 if rng.random() < 0.6 else np.float32
        pr_, pq_ = world.astype(dtype), pq.astype(dtype)
        pose = np.eye(4, dtype=np.float32)
        pose[:3, 3] = rng.standard_normal(3).astype(np.float32)
        use_filter = bool(rng.integers(0, 2))
        kind = rng.integers(0, 3)
        okw, wr, wq = {}, None, None
        if kind == 1:
            wr, wq = (rng.random((ov, K)) < 0.8).astype(np.uint8), (rng.random((ov, K)) < 0.8).astype(np.uint8)
            okw = dict(w_ref=wr, w_qry=wq)
        elif kind == 2:
            wr, wq = (rng.random((ov, K)) * (rng.random((ov, K)) < 0.9)).astype(np.float32), rng.random((ov, K)).astype(np.float32)
            okw = dict(weights_ref=wr, weights_qry=wq)
        ref = post_ref.align_chunks(pr_, pq_, kr, kq, pose, use_filter, **okw)
        idx = ops.sim3_match_keypoints(t(kr), t(kq))
        assert np.array_equal(idx.cpu().numpy(), ref["idx"]), (n, ov, K, mode)
        out = ops.sim3_umeyama(t(pr_), t(pq_), idx, t(pose), None if wr is None else t(wr), None if wq is None else t(wq),
                               use_filter).cpu().numpy()
        tag = (n, ov, K, int(mode), dtype.__name__, use_filter, int(kind))
        assert int(out[29]) == ref["n_used"] and int(out[30]) == ref["n_common"], tag
        if use_filter and ref["n_common"]:
            assert out[31] == ref["median"], tag
        if ref["n_used"] >= 3:
            # a handful of points can be (nearly) coplanar / collinear: the closed form is then ill-conditioned and the two
            # implementations may differ beyond 1e-11 although both are minimisers; compare the residual there
            cond_ok = ref["n_used"] >= 8
            if cond_ok:
                np.testing.assert_allclose(out[13:29].reshape(4, 4), ref["M"], atol=1e-9, rtol=1e-9, err_msg=str(tag))
            np.testing.assert_allclose(out[32], ref["rms"], rtol=1e-6, atol=1e-9, err_msg=str(tag))
        else:
            assert np.array_equal(out[13:29].reshape(4, 4), np.eye(4)), tag
    print(f"fuzz sim3: {n} cases")


def test_fuzz_gather_and_projection_against_oracle(dev):
    from oracle import post_ref
    from pi3_slam_amd import ops
    g = torch.Generator().manual_seed(99)
    rng = np.random.default_rng(99)
    t_end, n = time.time() + BUDGET / 4, 0
    while time.time() < t_end or n < 12:
        n += 1
        F = int(rng.integers(1, 5))
        H, W = int(rng.choice([2, 3, 14, 28, 57, 308])), int(rng.choice([2, 5, 14, 42, 91, 406]))
        K = int(rng.choice([1, 3, 17, 64, 200, 333]))
        pts = torch.randn(F, H, W, 3, generator=g) * 3
        lp = torch.randn(F, H, W, 3, generator=g).abs() + 0.1
        conf = torch.randn(F, H, W, 1, generator=g) * 3
        imgs = torch.rand(F, 3, H, W, generator=g)
        masks = torch.rand(F, H, W, generator=g) > 0.4
        kp = torch.rand(F, K, 2, generator=g) * torch.tensor([W - 1.0, H - 1.0])
        kp[:, 0] = 0.0                                                        # corners and the far edge
        if K > 2:
            kp[:, 1] = torch.tensor([W - 1.0, H - 1.0])
            kp[:, 2] = torch.tensor([W - 1.0, H - 1.0]) * 1.01                # just outside: border clamp
        out = ops.gather_keypoints(pts.to(dev), lp.to(dev), conf.to(dev), masks.to(torch.uint8).to(dev), imgs.to(dev),
                                   kp.to(dev))
        exp = post_ref.interpolate_at_keypoints(pts, lp, conf, masks, kp, H, W)
        tag = (n, F, H, W, K)
        assert torch.equal(out["masks"].bool().cpu().reshape(F, K), exp["masks"].reshape(F, K)), tag
        assert torch.equal(out["conf"].cpu().reshape(F, K), exp["conf"].to(torch.float16).reshape(F, K)), tag
        assert torch.equal(out["keypoints"].cpu(), kp.to(torch.float16)), tag
        for k in ("points", "local_points"):
            a, b = out[k].cpu().view(torch.int16), exp[k].to(torch.float16).view(torch.int16)
            assert torch.equal(a, b), (tag, k, float((a != b).float().mean()))
        col = post_ref.keypoint_colors(imgs, kp)
        assert torch.equal(out["colors"].cpu().float(), col.float()), tag
        # observation projection on the gathered chunk
        N = F
        poses = torch.eye(4).repeat(N, 1, 1)
        poses[:, :3, 3] = torch.randn(N, 3, generator=g) * 0.2
        intr = torch.zeros(N, 3, 3)
        intr[:, 0, 0], intr[:, 1, 1], intr[:, 0, 2], intr[:, 1, 2], intr[:, 2, 2] = 300.0, 310.0, W / 2, H / 2, 1.0
        p16 = (torch.randn(N, K, 3, generator=g) + torch.tensor([0.0, 0.0, 4.0])).half()
        uv, valid = ops.project_observations(p16.to(dev), poses.to(dev), intr.to(dev), W, H, 2)
        uv_ref, valid_ref = post_ref.project_observations(p16.numpy(), poses.numpy(), intr.numpy(), W, H, 2)
        got_valid = valid.cpu().numpy().astype(bool)
        # the reference inverts the fp32 pose with LAPACK in fp32, the device in fp64: a projection within 0.02 px of the
        # image border may land on the other side; everywhere else the in-bounds flags must be equal
        near = (np.abs(uv_ref[..., 0]) < 0.02) | (np.abs(uv_ref[..., 0] - W) < 0.02) | (np.abs(uv_ref[..., 1]) < 0.02) | \
            (np.abs(uv_ref[..., 1] - H) < 0.02)
        assert np.array_equal(got_valid[~near], valid_ref[~near]), tag
        m = valid_ref & got_valid
        np.testing.assert_allclose(uv.cpu().numpy()[m], uv_ref[m], rtol=2e-4, atol=2e-3, err_msg=str(tag))
    print(f"fuzz gather / projection: {n} cases")


def test_fuzz_bundle_adjust_against_schur_oracle(dev):
    from oracle import ba_ref
    from pi3_slam_amd import ops
    rng = np.random.default_rng(5)
    t_end, n = time.time() + BUDGET / 2, 0
    while time.time() < t_end or n < 4:
        n += 1
        N, K = int(rng.integers(3, 15)), int(rng.integers(4, 20))
        pb = make_problem(N=N, K=K, seed=int(rng.integers(0, 10 ** 6)), noise_px=float(rng.choice([0.0, 0.3, 1.0])),
                          outlier_frac=float(rng.choice([0.0, 0.05])), perturb=float(rng.uniform(0.2, 1.0)))
        huber = float(rng.choice([2.0, 3.0]))
        iters = int(rng.integers(1, 5))           # the well-conditioned phase: both take the same steps (test_ba_gpu.py)
        prior = None
        pr = pc = pf = None
        if rng.random() < 0.5:
            flag = (rng.random(N) < 0.4).astype(np.uint8)
            prior = dict(R=pb["R_gt"], C=pb["C_gt"] + 0.02, flag=flag, sqrt_info_rot=0.5 ** 0.5, sqrt_info_pos=0.2)
            pr, pc, pf = (torch.from_numpy(pb["R_gt"].reshape(N, 9)).to(dev), torch.from_numpy(prior["C"]).to(dev),
                          torch.from_numpy(flag).to(dev))
        form = int(rng.integers(0, 3))            # Euclidean steps / homogeneous point parametrization / inverse depth
        hom, invd = form == 1, form == 2
        if invd:
            R, C, X, s = ba_ref.bundle_adjust_inverse_depth(pb["R"], pb["C"], pb["intr"], pb["X"], pb["uv"], pb["valid"], huber,
                                                            iters, prior)
        else:
            R, C, X, s = ba_ref.bundle_adjust_schur(pb["R"], pb["C"], pb["intr"], pb["X"], pb["uv"], pb["valid"], huber, iters,
                                                    prior, homogeneous=hom)
        to = lambda a, dt=None: torch.from_numpy(np.ascontiguousarray(a)).to(dev) if dt is None else \
            torch.from_numpy(np.ascontiguousarray(a)).to(dev, dt)             # noqa: E731
        pts = to(pb["X"])
        rc = to(np.concatenate([pb["R"].reshape(N, 9), pb["C"]], 1))
        intr, uv, valid = to(pb["intr"]), to(pb["uv"]), to(pb["valid"])
        out = ops.bundle_adjust(pts, rc, intr, uv, valid, huber, iters, pr, pc, pf, 0.5 ** 0.5 if prior else 0.0,
                                0.2 if prior else 0.0, homogeneous=hom, inverse_depth=invd).cpu().numpy()
        tag = (n, N, K, huber, iters, prior is not None, form)
        assert abs(out[8] - s["initial_cost"]) <= 1e-9 * max(s["initial_cost"], 1e-12), tag
        assert (int(out[5]), int(out[6])) == (s["iterations"], s["accepted_steps"]), (tag, out[:10], s)
        assert abs(out[0] - s["final_cost"]) <= 1e-7 * max(s["final_cost"], 1e-9) + 1e-12, (tag, out[0], s["final_cost"])
        rcn = rc.cpu().numpy()
        np.testing.assert_allclose(rcn[:, 9:], C, atol=1e-6, err_msg=str(tag))
        np.testing.assert_allclose(pts.cpu().numpy(), X, atol=1e-5, err_msg=str(tag))
        est = ops.ba_outlier_tracks(pts, rc, intr, uv, valid, 2.0, 0.25).cpu().numpy().reshape(-1)
        ref = ba_ref.outlier_tracks(rcn[:, :9].reshape(N, 3, 3), rcn[:, 9:], pb["intr"], pts.cpu().numpy(), pb["uv"],
                                    pb["valid"], 2.0, 0.25)
        assert np.array_equal(est, ref), tag
    print(f"fuzz bundle adjustment: {n} cases")
