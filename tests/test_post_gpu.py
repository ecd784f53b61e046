"""GPU parity of the per-chunk post-processing and the Sim(3) alignment kernels against the oracle and the vectors
produced by the reference's own functions (tests/golden/post_*.npz).  Integer / index / boolean outputs are bit-exact;
fp16-packed values must equal the reference's fp16 values except where an fp32 ulp flips the fp16 rounding (bounded)."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev(built_lib):
    assert torch.cuda.is_available()
    from pi3_slam_amd import lib
    lib.load(require_gpu=True)
    return torch.device("cuda:0")


def _case(name):
    from oracle.gen_golden_post import CASES, synthetic_chunk
    N, H, W = CASES[name]
    return np.load(os.path.join(GOLDEN, name + ".npz")), synthetic_chunk(name, N, H, W), (N, H, W)


def _fp16_close(a, b, max_ulp=1, max_frac=2e-3):
    """fp16 tensors equal up to max_ulp fp16 ulps, and different in at most max_frac of the elements."""
    ai = a.cpu().view(torch.int16).to(torch.int32)
    bi = b.cpu().view(torch.int16).to(torch.int32)
    d = (ai - bi).abs()
    return int(d.max()) <= max_ulp and float((d > 0).float().mean()) <= max_frac


@pytest.mark.parametrize("name", ["post_a", "post_b"])
def test_masks_scale_gather_against_reference_vectors(dev, name):
    from pi3_slam_amd import ops
    g, d, (N, H, W) = _case(name)
    lp, conf, pts = d["local_points"].to(dev), d["conf"].to(dev), d["points"].to(dev)
    masks = ops.compute_masks(conf, lp)
    assert np.array_equal(masks.bool().cpu().numpy(), g["masks"]), "masks must be bit-exact on this fixture"
    med = ops.masked_ratio_median(d["moge_depth"].to(dev), lp[0][..., 2], 3, masks[0].contiguous(), H * W)
    assert np.float32(med[0].item()) == g["scale"] and int(med[1].item()) == int(g["masks"][0].sum())
    for tag in ("full", "sub"):
        kp = torch.from_numpy(g[f"kp_{tag}"]).to(dev)
        out = ops.gather_keypoints(pts, lp, conf, masks, d["images"].to(dev), kp)
        assert np.array_equal(out["masks"].cpu().numpy(), g[f"imasks_{tag}"])                   # nearest: exact
        assert np.array_equal(out["conf"].cpu().numpy(), g[f"iconf_{tag}"])                     # nearest: exact
        assert np.array_equal(out["keypoints"].cpu().numpy(), g[f"kp_{tag}"].astype(np.float16))
        # bilinear: the kernel restates ATen's FMA order, so fp16 values and uint8 colours are bit-exact as well
        assert _fp16_close(out["points"], torch.from_numpy(g[f"ipoints_{tag}"]), max_ulp=0, max_frac=0.0)
        assert _fp16_close(out["local_points"], torch.from_numpy(g[f"ilocal_{tag}"]), max_ulp=0, max_frac=0.0)
        col = out["colors"].cpu().float().numpy()
        ref = g[f"colors_{tag}"].astype(np.float32)
        assert np.array_equal(col, ref), ((col != ref).mean(), np.abs(col - ref).max())


def test_masks_edge_cases(dev):
    from oracle import post_ref
    from pi3_slam_amd import ops
    torch.manual_seed(3)
    F, H, W = 2, 9, 11
    lp = torch.rand(F, H, W, 3) + 0.5
    lp[0, 0, 0, 2] = 0.0                # z = 0 -> ratio inf -> nan_to_num -> edge
    lp[0, 4, 4, 2] = float("nan")
    lp[1, H - 1, W - 1, 2] = 10.0       # border pixel: only valid neighbours take part
    conf = torch.randn(F, H, W, 1) * 3
    got = ops.compute_masks(conf.to(dev), lp.to(dev)).bool().cpu()
    assert torch.equal(got, post_ref.compute_masks(conf, lp))


def _adversarial_maps(case, F, H, W, g):
    lp = torch.randn(F, H, W, 3, generator=g)
    z = torch.exp(0.3 * torch.randn(F, H, W, generator=g))
    if case in ("smooth", "conf_at_crossing"):          # depth ratios of the 3x3 window around rtol = 0.03
        z = (1.0 + torch.arange(H).view(1, H, 1) * 0.0148 + torch.arange(W).view(1, 1, W) * 0.0151
             + 1e-4 * torch.randn(F, H, W, generator=g))
    if case in ("specials", "specials_no_nan"):
        r = torch.rand(F, H, W, generator=g)
        if case == "specials":
            z = torch.where(r < 0.01, torch.full_like(z, float("nan")), z)
        z = torch.where((r > 0.01) & (r < 0.02), torch.full_like(z, float("inf")), z)
        z = torch.where((r > 0.02) & (r < 0.03), torch.zeros_like(z), z)
        z = torch.where((r > 0.03) & (r < 0.04), -z, z)
        z = torch.where((r > 0.04) & (r < 0.05), z * 1e-42, z)
        z = torch.where((r > 0.05) & (r < 0.06), z * 1e35, z)
        z = torch.where((r > 0.06) & (r < 0.07), torch.full_like(z, float("-inf")), z)
    lp[..., 2] = z
    conf = torch.randn(F, H, W, 1, generator=g) * 3
    if case == "conf_at_crossing":                       # confidences within 200 ulp of sigmoid(c) = 0.1
        c0 = torch.full((F, H, W, 1), -float(np.log(9.0)))
        conf = (c0.view(torch.int32) + torch.randint(-200, 200, (F, H, W, 1), generator=g, dtype=torch.int32)).view(torch.float32)
    return conf.contiguous(), lp.contiguous()


@pytest.mark.parametrize("case", ["random", "smooth", "specials", "specials_no_nan", "conf_at_crossing"])
@pytest.mark.parametrize("shape", [(6, 61, 406), (3, 17, 5), (2, 9, 1024), (1, 1, 1), (2, 14, 342)])
def test_masks_fast_decisions_equal_the_exact_arithmetic(dev, case, shape, monkeypatch):
    """masks_kernel decides most pixels without the IEEE division and the expf (post.hip: mask_pixel_fast) and falls
    back to them inside a band around the thresholds / for strips holding a NaN.  Both forms must give the decisions of
    the oracle (= the reference's arithmetic); with confidences packed within ulps of the sigmoid's crossing, where the
    device expf and the CPU exp may round differently, the two device forms are compared with each other."""
    from oracle import post_ref
    from pi3_slam_amd import ops
    g = torch.Generator().manual_seed(hash((case,) + shape) % (1 << 31))
    conf, lp = _adversarial_maps(case, *shape, g)
    for thr, rtol in [(0.1, 0.03), (0.5, 0.0), (0.9, 1.0), (-1.0, 0.03), (1.0, 0.03), (1e-30, 3e38), (0.1, float("inf"))]:
        monkeypatch.delenv("PI3_MASKS_EXACT_ONLY", raising=False)
        fast = ops.compute_masks(conf.to(dev), lp.to(dev), thr, rtol).cpu()
        monkeypatch.setenv("PI3_MASKS_EXACT_ONLY", "1")
        exact = ops.compute_masks(conf.to(dev), lp.to(dev), thr, rtol).cpu()
        monkeypatch.delenv("PI3_MASKS_EXACT_ONLY", raising=False)
        assert torch.equal(fast, exact), (case, shape, thr, rtol, int((fast != exact).sum()))
        if case != "conf_at_crossing":
            want = post_ref.compute_masks(conf, lp, conf_thr=thr, rtol=rtol)
            assert torch.equal(fast.bool(), want), (case, shape, thr, rtol, int((fast.bool() != want).sum()))


def test_ratio_median_semantics(dev):
    from pi3_slam_amd import ops
    torch.manual_seed(0)
    for n in (1, 2, 7, 1000, 125048):
        num, den = torch.rand(n) + 0.1, torch.rand(n) + 0.1
        mask = torch.rand(n) > 0.3
        mask[0] = True
        ref = (num[mask] / den[mask]).median()          # torch: LOWER median
        got = ops.masked_ratio_median(num.to(dev), den.to(dev), 1, mask.to(dev).view(torch.uint8), n).cpu()
        assert got[0].item() == ref.item() and int(got[1]) == int(mask.sum())
    num = torch.tensor([1.0, float("inf"), 3.0, -2.0, 5.0])
    got = ops.masked_ratio_median(num.to(dev), torch.ones(5, device=dev), 1, torch.ones(5, device=dev, dtype=torch.uint8), 5)
    assert got[0].item() == 3.0
    got = ops.masked_ratio_median(num.to(dev), torch.ones(5, device=dev), 1, torch.zeros(5, device=dev, dtype=torch.uint8), 5)
    assert np.isnan(got[0].item()) and got[1].item() == 0         # empty selection: NaN + count 0
    num[2] = float("nan")
    got = ops.masked_ratio_median(num.to(dev), torch.ones(5, device=dev), 1, torch.ones(5, device=dev, dtype=torch.uint8), 5)
    assert np.isnan(got[0].item())                                # NaN propagates like torch.median


def test_apply_scale(dev):
    from pi3_slam_amd import ops
    lp, pts, poses = torch.rand(2, 5, 7, 3, device=dev), torch.rand(2, 5, 7, 3, device=dev), torch.rand(2, 4, 4, device=dev)
    lp0, pts0, poses0 = lp.clone(), pts.clone(), poses.clone()
    ops.apply_scale(torch.tensor([1.37], device=dev), lp, pts, poses)
    assert torch.equal(lp, lp0 * 1.37) and torch.equal(pts, pts0 * 1.37)
    exp = poses0.clone(); exp[:, :3, 3] *= 1.37
    assert torch.equal(poses, exp)


@pytest.mark.parametrize("name", ["post_a", "post_b"])
def test_focal_shift_against_scipy_lm_vectors(dev, name):
    """The device LM restates MINPACK lmdif (what scipy's method='lm' runs); the reference stops at ftol = 1e-3, so
    the iterate sequence must be the same, not just the optimum: focal/shift agree to 1e-5 relative."""
    from pi3_slam_amd import ops
    from pi3_slam_amd.chunk_creator import _uv_tables
    g, d, (N, H, W) = _case(name)
    uvx, uvy = _uv_tables(H, W, dev)
    r = ops.focal_shift(d["local_points"].to(dev), d["conf"].to(dev), uvx, uvy)
    np.testing.assert_allclose(r["focal"].cpu().numpy(), g["focal"], rtol=1e-5)
    np.testing.assert_allclose(r["shift"].cpu().numpy(), g["shift"], rtol=1e-5, atol=1e-6)
    from oracle import post_ref
    ref = post_ref.estimate_camera_parameters(d["local_points"], d["conf"])
    np.testing.assert_allclose(r["intrinsics"].cpu().numpy(), ref["intrinsics"].numpy(), rtol=1e-5)
    assert torch.equal(r["fxfycxcy"][:, 2].cpu(), ref["cx"][0]) and torch.equal(r["fxfycxcy"][:, 3].cpu(), ref["cy"][0])


def test_focal_shift_degenerate_mask(dev):
    from pi3_slam_amd import ops
    from pi3_slam_amd.chunk_creator import _uv_tables
    H, W = 28, 42
    uvx, uvy = _uv_tables(H, W, dev)
    lp = torch.rand(1, H, W, 3, device=dev) + 1
    conf = torch.full((1, H, W, 1), -20.0, device=dev)           # nothing passes sigmoid > 0.1
    r = ops.focal_shift(lp, conf, uvx, uvy)
    assert r["focal"].item() == 1.0 and r["shift"].item() == 0.0  # geometry_torch.py:152-155


def _two_chunks(ov=6, K=40, noise=1e-3, seed=1, permute=False):
    rng = np.random.default_rng(seed)
    world = rng.standard_normal((ov, K, 3)) * 2 + np.array([0, 0, 5.0])
    ang = 0.4
    R = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]])
    s, t = 1.3, np.array([0.5, -0.2, 1.0])
    qry = ((world - t) @ R) / s + noise * rng.standard_normal(world.shape)       # world = s R qry + t
    kp = (rng.random((ov, K, 2)) * 300).astype(np.float16)
    kq = kp.copy()
    pq = qry.astype(np.float16)
    if permute:
        perm = rng.permutation(K)
        kq, pq = kq[:, perm], pq[:, perm]
        kq[0, 3] = [1234.0, 1234.0]           # one unmatched keypoint
    pose = np.eye(4, dtype=np.float32); pose[:3, 3] = [0.1, 0.0, 0.3]
    return world.astype(np.float16), pq, kp, kq, pose, (s, R, t)


@pytest.mark.parametrize("permute,use_filter", [(False, True), (True, True), (True, False)])
def test_sim3_against_oracle(dev, permute, use_filter):
    from oracle import post_ref
    from pi3_slam_amd import ops
    pr, pq, kr, kq, pose, (s, R, t) = _two_chunks(permute=permute)
    ref = post_ref.align_chunks(pr, pq, kr, kq, pose, use_filter)
    idx = ops.sim3_match_keypoints(torch.from_numpy(kr).to(dev), torch.from_numpy(kq).to(dev))
    assert np.array_equal(idx.cpu().numpy(), ref["idx"]), "match indices must be bit-exact"
    out = ops.sim3_umeyama(torch.from_numpy(pr).to(dev), torch.from_numpy(pq).to(dev), idx,
                           torch.from_numpy(pose).to(dev), None, None, use_filter).cpu().numpy()
    assert int(out[29]) == ref["n_used"] and int(out[30]) == ref["n_common"]
    if use_filter:
        assert out[31] == ref["median"]                      # exact order statistics (np.median)
    np.testing.assert_allclose(out[0], ref["s"], rtol=1e-12)
    np.testing.assert_allclose(out[1:10].reshape(3, 3), ref["R"], atol=1e-12)
    np.testing.assert_allclose(out[10:13], ref["t"], atol=1e-11)
    np.testing.assert_allclose(out[13:29].reshape(4, 4), ref["M"], atol=1e-11)
    np.testing.assert_allclose(out[32], ref["rms"], rtol=1e-9)
    assert abs(out[0] - s) < 5e-3 and np.abs(out[1:10].reshape(3, 3) - R).max() < 5e-3      # recovers the truth


def _solve_pairs(dev, x, y, use_filter=False, pose=None, kp=None):
    """Push explicit point pairs (qry x -> ref y) through the match + filter + closed-form kernels."""
    from pi3_slam_amd import ops
    n = len(x)
    if kp is None:
        kp = (np.arange(2 * n, dtype=np.float32).reshape(1, n, 2) * 0.5).astype(np.float16)
    kr = kq = kp
    pose = np.eye(4, dtype=np.float32) if pose is None else pose
    idx = ops.sim3_match_keypoints(torch.from_numpy(kr).to(dev), torch.from_numpy(kq).to(dev))
    out = ops.sim3_umeyama(torch.from_numpy(y.astype(np.float16)[None]).to(dev),
                           torch.from_numpy(x.astype(np.float16)[None]).to(dev), idx,
                           torch.from_numpy(pose).to(dev), None, None, use_filter).cpu().numpy()
    return idx.cpu().numpy(), out


@pytest.mark.parametrize("kind", ["mirrored", "planar", "collinear", "three_points"])
def test_sim3_degenerate_configurations(dev, kind):
    """Parity unpinned (pytheia absent): the kernel (Horn quaternion, Jacobi) must agree with BOTH oracle closed forms
    (SVD Umeyama and numpy Horn) where the optimum is unique, and return a proper rotation with the minimal residual
    where it is not (collinear)."""
    from oracle import post_ref
    rng = np.random.default_rng(11)
    n = 3 if kind == "three_points" else 60
    x = (rng.standard_normal((n, 3)) * np.array([2.0, 1.0, 0.5]) + np.array([0.0, 0.0, 4.0]))
    ang = 0.5
    R0 = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]])
    if kind == "planar":
        x[:, 2] = 4.0 + 0.25 * x[:, 0]
    if kind == "collinear":
        x = np.outer(rng.standard_normal(n), [1.0, 0.5, -0.25]) + np.array([0.0, 0.0, 4.0])
    y = 1.25 * x @ R0.T + np.array([0.25, -0.5, 1.0])
    if kind == "mirrored":
        y = y * np.array([1.0, -1.0, 1.0])
    x16, y16 = x.astype(np.float16).astype(np.float64), y.astype(np.float16).astype(np.float64)   # what the kernel sees
    _, out = _solve_pairs(dev, x, y)
    s, R, t, M = out[0], out[1:10].reshape(3, 3), out[10:13], out[13:29].reshape(4, 4)
    assert int(out[29]) == n and np.isfinite(out[:31]).all() and np.isfinite(out[32])    # out[31] = median: inf, no filter
    assert np.abs(R @ R.T - np.eye(3)).max() < 1e-12 and abs(np.linalg.det(R) - 1.0) < 1e-12
    s1, R1, t1, M1 = post_ref.umeyama(x16, y16)
    s2, R2, t2, M2 = post_ref.horn_sim3(x16, y16)
    rms = lambda s_, R_, t_: np.sqrt((((s_ * x16 @ R_.T) + t_ - y16) ** 2).sum(1).mean())
    if kind == "collinear":       # rotation about the line is free: same residual, same mapped points
        np.testing.assert_allclose(out[32], rms(s1, R1, t1), atol=1e-9)
        np.testing.assert_allclose(s * x16 @ R.T + t, s1 * x16 @ R1.T + t1, atol=1e-6)
    else:
        np.testing.assert_allclose(M, M1, atol=1e-9, rtol=1e-9)
        np.testing.assert_allclose(M, M2, atol=1e-9, rtol=1e-9)
        np.testing.assert_allclose(out[32], rms(s1, R1, t1), rtol=1e-7, atol=1e-12)


def test_sim3_too_few_pairs_and_everything_filtered(dev):
    """< 3 usable pairs -> identity + the count (the host turns that into (False, {...}) like
    reconstruction_alignment.py:99-101); the strict '<' median filter (:78-86) drops EVERY pair when all reference
    points are equidistant from the last camera."""
    from pi3_slam_amd.alignment import align_and_refine_reconstructions
    rng = np.random.default_rng(2)
    x = rng.standard_normal((2, 3)) + [0, 0, 3.0]
    _, out = _solve_pairs(dev, x, x * 2.0)
    assert int(out[29]) == 2 and np.array_equal(out[13:29].reshape(4, 4), np.eye(4)) and out[0] == 1.0
    # every pair filtered: reference points on a sphere of radius exactly 2 (fp16-representable) around the camera
    dirs = np.array([[1, 0, 0], [0, 1, 0], [0, 0, 1], [-1, 0, 0], [0, -1, 0], [0, 0, -1]], np.float64)
    y = 2.0 * dirs
    _, out = _solve_pairs(dev, dirs, y, use_filter=True)
    assert int(out[30]) == 6 and int(out[29]) == 0 and out[31] == 2.0
    assert np.array_equal(out[13:29].reshape(4, 4), np.eye(4))
    # no common keypoint at all
    from pi3_slam_amd import ops
    kr = np.zeros((1, 4, 2), np.float16); kq = np.ones((1, 4, 2), np.float16)
    idx = ops.sim3_match_keypoints(torch.from_numpy(kr).to(dev), torch.from_numpy(kq).to(dev))
    assert (idx.cpu().numpy() == -1).all()
    # host surface: failure is reported, not raised
    K = 5
    mk = lambda pts: {"points": torch.from_numpy(np.tile(pts.astype(np.float16), (3, 1, 1))),
                      "keypoints": torch.from_numpy(np.tile((np.arange(2 * K).reshape(K, 2) + 1000 * (pts[0, 0] > 0)).astype(np.float16), (3, 1, 1))),
                      "masks": torch.ones(3, K, 1, dtype=torch.bool), "camera_poses": torch.eye(4).repeat(3, 1, 1)}
    a, b = mk(rng.standard_normal((K, 3)) - 5.0), mk(np.abs(rng.standard_normal((K, 3))) + 5.0)
    ok, info = align_and_refine_reconstructions(a, b, [(1, 0), (2, 1)], device=str(dev))
    assert ok is False and "error" in info


def test_sim3_duplicated_keypoints_take_the_first_reference_track(dev):
    """Two reference keypoints with identical pixels: the qry keypoint pairs with the FIRST one (oracle semantics)."""
    from oracle import post_ref
    from pi3_slam_amd import ops
    rng = np.random.default_rng(4)
    ov, K = 2, 16
    kr = (rng.random((ov, K, 2)) * 200).astype(np.float16)
    kr[:, 9] = kr[:, 2]                      # duplicate of keypoint 2 later in the list
    kr[:, 12] = kr[:, 2]
    kq = kr[:, ::-1].copy()
    pr = rng.standard_normal((ov, K, 3)).astype(np.float16)
    pq = rng.standard_normal((ov, K, 3)).astype(np.float16)
    idx = ops.sim3_match_keypoints(torch.from_numpy(kr).to(dev), torch.from_numpy(kq).to(dev)).cpu().numpy()
    ref = post_ref.align_chunks(pr, pq, kr, kq, np.eye(4, dtype=np.float32), False)
    assert np.array_equal(idx, ref["idx"])
    assert idx[0, K - 1 - 9] == 2 and idx[0, K - 1 - 12] == 2 and idx[0, K - 1 - 2] == 2
    out = ops.sim3_umeyama(torch.from_numpy(pr).to(dev), torch.from_numpy(pq).to(dev),
                           torch.from_numpy(idx).to(dev), torch.eye(4, device=dev), None, None, False).cpu().numpy()
    np.testing.assert_allclose(out[13:29].reshape(4, 4), ref["M"], atol=1e-10)
    _, _, _, M2 = post_ref.horn_sim3(*[np.array(v, np.float64) for v in _pairs_of(ref["idx"], pq, pr)])
    np.testing.assert_allclose(out[13:29].reshape(4, 4), M2, atol=1e-9)


def _pairs_of(idx, pq, pr):
    xs, ys = [], []
    for v in range(idx.shape[0]):
        for j in range(idx.shape[1]):
            if idx[v, j] >= 0:
                xs.append(pq[v, j].astype(np.float64)); ys.append(pr[v, idx[v, j]].astype(np.float64))
    return xs, ys


def test_sim3_apply_and_prefix(dev):
    from oracle import post_ref
    from pi3_slam_amd import ops
    rng = np.random.default_rng(0)
    _, _, _, _, _, (s, R, t) = _two_chunks()
    M = np.eye(4); M[:3, :3] = s * R; M[:3, 3] = t
    pts = rng.standard_normal((1000, 3)).astype(np.float32)
    poses = np.tile(np.eye(4, dtype=np.float32), (5, 1, 1)); poses[:, :3, 3] = rng.standard_normal((5, 3))
    p_ref, P_ref = post_ref.apply_sim3(M, pts, poses)
    pd, Pd = torch.from_numpy(pts).to(dev), torch.from_numpy(poses).to(dev)
    ops.sim3_apply(torch.from_numpy(M).to(dev), pd, Pd)
    np.testing.assert_allclose(pd.cpu().numpy(), p_ref, rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(Pd.cpu().numpy(), P_ref, rtol=1e-6, atol=1e-6)
    T = np.stack([np.eye(4), M, M, np.linalg.inv(M)]).reshape(4, 16)
    G = ops.sim3_compose_prefix(torch.from_numpy(T).to(dev)).cpu().numpy().reshape(4, 4, 4)
    np.testing.assert_allclose(G[2], M @ M, rtol=1e-12)
    np.testing.assert_allclose(G[3], M, rtol=1e-10, atol=1e-12)


def test_alignment_is_similarity_equivariant(dev):
    """Chunk-parallel == sequential for the closed-form step: aligning to a transformed reference composes."""
    from pi3_slam_amd import ops
    pr, pq, kr, kq, pose, _ = _two_chunks(noise=1e-3)
    rng = np.random.default_rng(5)
    ang = 1.1
    Rg = np.array([[np.cos(ang), -np.sin(ang), 0], [np.sin(ang), np.cos(ang), 0], [0, 0, 1.0]])
    G = np.eye(4); G[:3, :3] = 2.0 * Rg; G[:3, 3] = [3.0, -1.0, 0.5]
    def solve(pref, posem):
        idx = ops.sim3_match_keypoints(torch.from_numpy(kr).to(dev), torch.from_numpy(kq).to(dev))
        return ops.sim3_umeyama(torch.from_numpy(pref).to(dev), torch.from_numpy(pq).to(dev), idx,
                                torch.from_numpy(posem).to(dev), None, None, True).cpu().numpy()
    a = solve(pr, pose)
    pr_g = ((G[:3, :3] @ pr.astype(np.float64).reshape(-1, 3).T).T + G[:3, 3]).reshape(pr.shape).astype(np.float16)
    pose_g = pose.copy(); pose_g[:3, 3] = G[:3, :3] @ pose[:3, 3] + G[:3, 3]
    b = solve(pr_g, pose_g)
    np.testing.assert_allclose(b[13:29].reshape(4, 4), G @ a[13:29].reshape(4, 4), rtol=2e-2, atol=2e-2)   # fp16 re-rounding of the transformed reference


def test_offline_reconstructor_roundtrip(dev, tmp_path):
    """chunk files -> OfflineReconstructor.run(): overlapping chunks that differ by known similarities come back on
    one trajectory; TUM file format and first-occurrence de-duplication as offline_reconstructor.py:218-255."""
    import json
    from pi3_slam_amd.reconstructor import OfflineReconstructor
    rng = np.random.default_rng(0)
    cl, ov, K, nchunks = 8, 3, 30, 3
    n_frames = cl + (nchunks - 1) * (cl - ov)
    gt_pos = np.stack([np.array([0.1 * i, 0.02 * i, 0.0]) for i in range(n_frames)])
    world_pts = {i: rng.standard_normal((K, 3)) + np.array([0.1 * i, 0, 4.0]) for i in range(n_frames)}
    kp = (rng.random((K, 2)) * 300).astype(np.float16)
    os.makedirs(tmp_path / "chunks")
    for c in range(nchunks):
        start = c * (cl - ov)
        ang, s = 0.3 * c, 1.0 + 0.2 * c
        R = np.array([[np.cos(ang), -np.sin(ang), 0], [np.sin(ang), np.cos(ang), 0], [0, 0, 1.0]])
        t = np.array([0.5 * c, -0.3 * c, 0.1 * c])
        inv = lambda X: ((X - t) @ R) / s                        # chunk frame = S^-1 (world)
        poses = np.tile(np.eye(4, dtype=np.float32), (cl, 1, 1))
        pts = np.zeros((cl, K, 3), np.float16)
        for j in range(cl):
            poses[j, :3, :3] = R.T
            poses[j, :3, 3] = inv(gt_pos[start + j])
            pts[j] = inv(world_pts[start + j])
        torch.save({"points": torch.from_numpy(pts), "keypoints": torch.from_numpy(np.tile(kp, (cl, 1, 1))),
                    "masks": torch.ones(cl, K, 1, dtype=torch.bool), "colors": torch.full((cl, K, 3), 128.0).half(),
                    "camera_poses": torch.from_numpy(poses), "image_paths": [[f"img_{start + j:04d}.png"] for j in range(cl)],
                    "original_width": 406, "original_height": 308, "chunk_index": c},
                   tmp_path / "chunks" / f"chunk_{c:06d}.pt")
    json.dump({"chunk_length": cl, "overlap": ov, "target_size": [308, 406]}, open(tmp_path / "chunk_metadata.json", "w"))
    # random keypoints / points / poses with no camera model behind them: the closed-form chain is what this test is
    # about (bundle adjustment of physically inconsistent data is meaningless; it has its own tests in test_ba_gpu.py)
    rec = OfflineReconstructor(str(tmp_path), str(tmp_path / "out"), bundle_adjust=False)
    assert rec.chunk_length == cl and rec.overlap == ov
    rec.run()
    lines = open(tmp_path / "out" / "trajectory_tum.txt").read().strip().split("\n")
    assert lines[0] == "# timestamp tx ty tz qx qy qz qw" and len(lines) == n_frames + 1
    traj = np.array([[float(v) for v in l.split()[1:4]] for l in lines[1:]])
    assert np.abs(traj - gt_pos).max() < 2e-2                    # chunk 0 is the world frame; fp16 point storage
    assert [l.split()[0] for l in lines[1:4]] == ["0", "1", "2"]
    assert os.path.exists(tmp_path / "out" / "final_points.ply") and os.path.exists(tmp_path / "out" / "final_camera_poses.ply")


def test_apply_scale_ignores_invalid_median(dev):
    from pi3_slam_amd import ops
    lp, pts, poses = torch.rand(1, 3, 4, 3, device=dev), torch.rand(1, 3, 4, 3, device=dev), torch.rand(1, 4, 4, device=dev)
    for bad in (float("nan"), 0.0, -1.0, float("inf")):
        a, b, c = lp.clone(), pts.clone(), poses.clone()
        ops.apply_scale(torch.tensor([bad], device=dev), a, b, c)
        assert torch.equal(a, lp) and torch.equal(b, pts) and torch.equal(c, poses)


def test_project_observations_against_reference_vectors(dev):
    """§8f rank 1: observation projection kernel vs vectors produced by the reference's ChunkPTRecon methods."""
    from pi3_slam_amd import ops
    from pi3_slam_amd.observations import project_chunk_observations
    g = np.load(os.path.join(GOLDEN, "post_proj.npz"))
    N, K, W, H, max_after = [int(v) for v in g["shape"]]
    pts = torch.from_numpy(g["points"]).to(dev)
    poses = torch.from_numpy(g["poses"]).to(dev)
    intr = torch.from_numpy(g["intrinsics"]).to(dev)
    uv, valid = ops.project_observations(pts, poses, intr, W, H, max_after)
    torch.cuda.synchronize()
    uv, valid = uv.cpu().numpy(), valid.cpu().numpy()
    ref_uv, ref_valid = g["uv"], g["valid"]
    pair = np.zeros((N, N), dtype=bool)
    for s in range(N):
        pair[s, :s] = True
        pair[s, s + 1:s + 1 + max_after] = True
    assert not valid[~pair].any()
    # tolerance: the reference inverts the fp32 pose with LAPACK sgetri, the kernel in fp64 -> ~1e-6 relative on the
    # camera-frame point, amplified by 1/z for points near the camera plane.  Compare where the reference value is
    # finite and moderately sized; require the in-bounds flag to agree except within 1e-2 px of a border.
    sel = pair[:, :, None] & np.isfinite(ref_uv).all(-1) & (np.abs(ref_uv).max(-1) < 1e4)
    err = np.abs(uv - ref_uv)[sel]
    scale = np.maximum(1.0, np.abs(ref_uv)[sel])
    assert (err / scale).max() < 2e-4, (err / scale).max()
    flips = (valid != ref_valid) & pair[:, :, None]
    if flips.any():
        u, v = ref_uv[..., 0][flips], ref_uv[..., 1][flips]
        d = np.minimum.reduce([np.abs(u), np.abs(u - W), np.abs(v), np.abs(v - H)])
        assert d.max() < 1e-2
    assert flips.mean() < 1e-3
    # host mirror: same observations, in the reference's loop order
    chunk = {"points": pts, "camera_poses": poses, "intrinsics": intr}
    obs = project_chunk_observations(chunk, W, H, max_observations_per_track=2 * max_after + 1)
    s, t, k = np.nonzero(valid)
    assert np.array_equal(obs["source_frame"].cpu().numpy(), s)
    assert np.array_equal(obs["target_frame"].cpu().numpy(), t)
    assert np.array_equal(obs["keypoint"].cpu().numpy(), k)
    assert np.array_equal(obs["uv"].cpu().numpy(), uv[s, t, k])
    # full chunk size (cl=100, K=200 keypoints): count agrees with the oracle
    from oracle import post_ref
    gen = torch.Generator().manual_seed(5)
    Nf, Kf = 100, 200
    P = torch.eye(4).repeat(Nf, 1, 1)
    P[:, 0, 3] = torch.arange(Nf) * 0.05
    X = (torch.randn(Nf, Kf, 3, generator=gen) * torch.tensor([2.0, 1.5, 1.0]) + torch.tensor([2.5, 0.0, 4.0])).half()
    Kf33 = torch.tensor([[400.0, 0, 320], [0, 400.0, 240], [0, 0, 1]]).repeat(Nf, 1, 1)
    uvf, vf = ops.project_observations(X.to(dev), P.to(dev), Kf33.to(dev), 640, 480, 2)
    ouv, ov_ = post_ref.project_observations(X.numpy(), P.numpy(), Kf33.numpy(), 640, 480, 2)
    vf = vf.cpu().numpy()
    assert (vf != ov_).mean() < 1e-4
    both = vf & ov_
    assert np.abs(uvf.cpu().numpy()[both] - ouv[both]).max() < 1e-2


@pytest.mark.parametrize("name", ["ingest_down", "ingest_up", "ingest_mixed"])
def test_ingest_frames_bit_exact_with_pillow_vectors(dev, name):
    """§8f rank 2: device Resize + ToTensor vs vectors produced by Pillow (what transforms.Resize runs on a PIL image)."""
    from pi3_slam_amd.image_io import ingest_frames_device
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    H1, W1 = [int(v) for v in g["target"]]
    out = ingest_frames_device(torch.from_numpy(g["frames"]).to(dev), (H1, W1))
    torch.cuda.synchronize()
    assert out.shape == (g["frames"].shape[0], 3, H1, W1) and out.dtype == torch.float32
    assert np.array_equal(out.cpu().numpy(), g["tensor"])                # bit-exact, including the /255


def test_ingest_full_size_against_oracle_and_dataset_item(dev, tmp_path):
    """Full ingest size (EuRoC-like 480x752 -> 308x406, 100 frames would be 108 MB: 12 frames here) against the oracle,
    and ChunkImageDataset.load_chunk_device against the host loader item on real PNG files."""
    from PIL import Image
    from oracle import ingest_ref
    from pi3_slam_amd.image_io import ChunkImageDataset, ingest_frames_device, target_size_for
    rng = np.random.default_rng(3)
    frames = rng.integers(0, 256, (12, 480, 752, 3), dtype=np.uint8)
    H1, W1 = target_size_for(752, 480)
    out = ingest_frames_device(torch.from_numpy(frames).to(dev), (H1, W1)).cpu().numpy()
    assert np.array_equal(out[:3], ingest_ref.ingest_frames(frames[:3], (H1, W1)))
    assert np.array_equal(out[11], ingest_ref.ingest_frames(frames[11:], (H1, W1))[0])
    paths = []
    for i in range(5):
        p = str(tmp_path / f"f{i:03d}.png")
        Image.fromarray(frames[i, :120, :160]).save(p)
        paths.append(p)
    ds = ChunkImageDataset(paths, 4, 1, (56, 70))
    for idx in range(len(ds)):
        host, devi = ds[idx], ds.load_chunk_device(idx, dev)
        assert torch.equal(host["chunk"], devi["chunk"].cpu())
        assert host["chunk_paths"] == devi["chunk_paths"] and int(host["start_idx"]) == int(devi["start_idx"])


@pytest.mark.parametrize("calib,target", [("euroc_cam0_calib.json", (308, 406)), ("euroc_cam0_calib.json", (480, 752)),
                                          ("cam_calib.json", (378, 504))])
def test_undistortion_maps_and_remap_against_oracle(dev, calib, target):
    """§8f rank 4: map builder (4 camera models) and cv2.remap restatement, kernels vs the numpy oracle."""
    import json
    from oracle import undistort_ref as U
    from pi3_slam_amd.undistortion import Camera, UndistortionMaps
    cal = json.load(open(os.path.join(GOLDEN, "calib_" + calib)))
    cam = Camera()
    cam.load_camera_calibration_json(cal, 1.0)
    maps = UndistortionMaps(cam, device=str(dev))
    mx, my = maps.get_maps(target)
    rx, ry = U.undistort_maps(cal, target)
    # fp64 formulas on both sides, one rounding to fp32: at most 1 ulp apart (sqrt / fma contraction differences)
    assert np.abs(mx.cpu().numpy() - rx).max() <= 6.2e-5 and np.abs(my.cpu().numpy() - ry).max() <= 6.2e-5
    rng = np.random.default_rng(11)
    frames = rng.integers(0, 256, (3, cal["image_height"], cal["image_width"], 3), dtype=np.uint8)
    out = maps.undistort_frames_device(torch.from_numpy(frames).to(dev), target)
    # remap on the DEVICE maps (integer arithmetic from there on: bit-exact)
    ref = np.stack([U.remap_bilinear_u8(f, mx.cpu().numpy(), my.cpu().numpy()) for f in frames])
    ref = (ref.astype(np.float32) / np.float32(255.0)).transpose(0, 3, 1, 2)
    assert np.array_equal(out.cpu().numpy(), ref)
    with pytest.raises(NotImplementedError):
        maps.undistort_frames_device(torch.zeros(1, 100, 100, 3, dtype=torch.uint8, device=dev), target)


def test_undistortion_models_synthetic(dev):
    """PINHOLE and FISHEYE (no example calibration in the reference): kernels vs oracle on synthetic parameters, maps
    partly outside the image (border taps) and a skewed camera."""
    from oracle import undistort_ref as U
    from pi3_slam_amd.undistortion import Camera, UndistortionMaps
    base = {"image_height": 240, "image_width": 320}
    cals = [
        {**base, "intrinsic_type": "PINHOLE", "intrinsics": {"aspect_ratio": 1.02, "focal_length": 210.0,
         "principal_pt_x": 158.3, "principal_pt_y": 121.9, "radial_distortion_1": -0.21, "radial_distortion_2": 0.05,
         "skew": 0.4}},
        {**base, "intrinsic_type": "FISHEYE", "intrinsics": {"aspect_ratio": 0.99, "focal_length": 150.0,
         "principal_pt_x": 160.0, "principal_pt_y": 120.0, "radial_distortion_1": -0.03, "radial_distortion_2": 0.01,
         "radial_distortion_3": -0.002, "radial_distortion_4": 0.0003, "skew": 0.0}},
    ]
    img = np.random.default_rng(5).integers(0, 256, (2, 240, 320, 3), dtype=np.uint8)
    for cal in cals:
        cam = Camera()
        cam.load_camera_calibration_json(cal, 1.0)
        maps = UndistortionMaps(cam, device=str(dev))
        mx, my = maps.get_maps((240, 320))
        rx, ry = U.undistort_maps(cal, (240, 320))
        assert np.abs(mx.cpu().numpy() - rx).max() <= 6.2e-5 and np.abs(my.cpu().numpy() - ry).max() <= 6.2e-5
        out = maps.undistort_frames_device(torch.from_numpy(img).to(dev), (240, 320)).cpu().numpy()
        ref = np.stack([U.remap_bilinear_u8(f, mx.cpu().numpy(), my.cpu().numpy()) for f in img])
        assert np.array_equal(out, (ref.astype(np.float32) / np.float32(255.0)).transpose(0, 3, 1, 2))
