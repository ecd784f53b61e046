"""Sim(3) overlap alignment (utils/reconstruction_alignment.py:74-105) against the oracle AT THE SIZE
IT SHIPS: ov = 20 views x K = 200 / 400 keypoints = 4 000 / 8 000 pairs (two chunks related by a known
(s, R, t), Gaussian noise sigma = 1 mm).  sim3_umeyama_kernel is one 1 024-thread workgroup walking the pairs with a
stride of 1 024, so anything below 1 025 pairs never takes the second trip of its loops (count, radix select, means,
covariance, residual); here every loop runs 4-8 trips.  Covered: filter on / off, fp16 (chunk-file) and fp32
(bundle-adjusted) points, the validity-mask variant, the real-valued weighted variant, a ragged match (two independent
per-frame random subsets of the grid, as two chunk creations give: ~170 of 200 common), a short previous chunk through
both host entry points.  Match indices bit-exact; median exact; s, R, t, M to 1e-11.

Parity status: the arithmetic lives in pytheia 0.2.9 (absent offline) - the oracle restates it from the call sites and
the published closed form: "parity unpinned" (DESIGN.md §2); this file pins the KERNEL to that oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

H, W, OV = 308, 406, 20


@pytest.fixture(scope="module")
def dev(built_lib):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def _overlap(K, seed, ragged, noise=1e-3, ov=OV):
    """Two chunks' overlap blocks.  Every grid position g of view v carries one world point X[v, g]; the ref chunk sees
    it as is, the qry chunk in its own frame (world = s R qry + t) plus noise.  ragged: each chunk keeps its own random
    K-subset of the grid per view (keypoint_extraction.py:140-143), in its own order."""
    from oracle import post_ref
    rng = np.random.default_rng(seed)
    sp = post_ref.grid_spacing(H, W, K)
    gx, gy = np.arange(min(H, W) * 0.05, W - min(H, W) * 0.05, sp), np.arange(min(H, W) * 0.05, H - min(H, W) * 0.05, sp)
    full = np.stack(np.meshgrid(gx, gy), -1).reshape(-1, 2).astype(np.float32)
    G = len(full)
    assert G >= K
    X = rng.standard_normal((ov, G, 3)) * np.array([2.0, 1.5, 1.0]) + np.array([0.3, -0.2, 6.0])
    ang, ax = 0.35, np.array([0.2, 1.0, -0.1]) / np.linalg.norm([0.2, 1.0, -0.1])
    Kx = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
    R = np.eye(3) + np.sin(ang) * Kx + (1 - np.cos(ang)) * Kx @ Kx
    s, t = 1.23, np.array([0.4, -0.3, 0.8])
    Xq = ((X - t) @ R) / s + noise * rng.standard_normal(X.shape)
    if ragged:
        sel_r = np.stack([rng.permutation(G)[:K] for _ in range(ov)])
        sel_q = np.stack([rng.permutation(G)[:K] for _ in range(ov)])
    else:
        sel_r = sel_q = np.stack([rng.permutation(G)[:K]] * ov)               # same subset, same order
    take = lambda A, sel: np.stack([A[v][sel[v]] for v in range(len(sel))])   # noqa: E731
    kp_r = np.stack([full[sel_r[v]] for v in range(ov)]).astype(np.float16)
    kp_q = np.stack([full[sel_q[v]] for v in range(ov)]).astype(np.float16)
    pose = np.eye(4, dtype=np.float32)
    pose[:3, 3] = [0.25, -0.1, 0.4]
    return dict(pr=take(X, sel_r), pq=take(Xq, sel_q), kr=kp_r, kq=kp_q, pose=pose, truth=(s, R, t), G=G)


def _check(out, ref, use_filter, tight=True):
    assert int(out[29]) == ref["n_used"] and int(out[30]) == ref["n_common"]
    if use_filter:
        assert out[31] == ref["median"]                                       # exact order statistics (np.median)
    np.testing.assert_allclose(out[0], ref["s"], rtol=1e-12)
    np.testing.assert_allclose(out[1:10].reshape(3, 3), ref["R"], atol=1e-12)
    np.testing.assert_allclose(out[10:13], ref["t"], atol=1e-11)
    np.testing.assert_allclose(out[13:29].reshape(4, 4), ref["M"], atol=1e-11)
    np.testing.assert_allclose(out[32], ref["rms"], rtol=1e-9)


@pytest.mark.parametrize("K", [200, 400])
@pytest.mark.parametrize("ragged", [False, True])
@pytest.mark.parametrize("use_filter", [True, False])
@pytest.mark.parametrize("dtype", [np.float16, np.float32])
def test_sim3_at_shipping_size_vs_oracle(dev, K, ragged, use_filter, dtype):
    from oracle import post_ref
    from pi3_slam_amd import ops
    c = _overlap(K, seed=1 + K + ragged, ragged=ragged)
    pr, pq = c["pr"].astype(dtype), c["pq"].astype(dtype)
    ref = post_ref.align_chunks(pr, pq, c["kr"], c["kq"], c["pose"], use_filter)
    idx = ops.sim3_match_keypoints(torch.from_numpy(c["kr"]).to(dev), torch.from_numpy(c["kq"]).to(dev))
    assert np.array_equal(idx.cpu().numpy(), ref["idx"]), "match indices must be bit-exact"
    n_common = int((ref["idx"] >= 0).sum())
    assert n_common == ref["n_common"] > 1024                                  # several trips of every stride loop
    if ragged:                                                                 # ~K^2/G common keypoints per view
        assert 0.6 * K * K / c["G"] < n_common / OV < 1.25 * K * K / c["G"]
    else:
        assert n_common == OV * K
    out = ops.sim3_umeyama(torch.from_numpy(pr).to(dev), torch.from_numpy(pq).to(dev), idx,
                           torch.from_numpy(c["pose"]).to(dev), None, None, use_filter).cpu().numpy()
    _check(out, ref, use_filter)
    s, R, t = c["truth"]                                                       # and the truth, to the noise level
    assert abs(out[0] - s) < 2e-3 and np.abs(out[1:10].reshape(3, 3) - R).max() < 2e-3 and np.abs(out[10:13] - t).max() < 1e-2


@pytest.mark.parametrize("K", [200, 400])
@pytest.mark.parametrize("dtype", [np.float16, np.float32])
def test_sim3_validity_masks_at_shipping_size(dev, K, dtype):
    """The mask variant (alignment.estimate_sim3(use_masks=True)): a pair takes part only if both keypoints are valid."""
    from oracle import post_ref
    from pi3_slam_amd import ops
    c = _overlap(K, seed=7 + K, ragged=True)
    rng = np.random.default_rng(3)
    wr, wq = (rng.random((OV, K)) < 0.8).astype(np.uint8), (rng.random((OV, K)) < 0.7).astype(np.uint8)
    pr, pq = c["pr"].astype(dtype), c["pq"].astype(dtype)
    for use_filter in (True, False):
        ref = post_ref.align_chunks(pr, pq, c["kr"], c["kq"], c["pose"], use_filter, w_ref=wr, w_qry=wq)
        idx = ops.sim3_match_keypoints(torch.from_numpy(c["kr"]).to(dev), torch.from_numpy(c["kq"]).to(dev))
        out = ops.sim3_umeyama(torch.from_numpy(pr).to(dev), torch.from_numpy(pq).to(dev), idx,
                               torch.from_numpy(c["pose"]).to(dev), torch.from_numpy(wr).to(dev),
                               torch.from_numpy(wq).to(dev), use_filter).cpu().numpy()
        assert 1024 < ref["n_common"] < int((ref["idx"] >= 0).sum())          # the masks removed pairs
        _check(out, ref, use_filter)


@pytest.mark.parametrize("K", [200, 400])
@pytest.mark.parametrize("use_filter", [True, False])
def test_sim3_real_valued_weights_vs_oracle(dev, K, use_filter):
    """pi3_sim3_umeyama_weighted (the north star's "weighted Umeyama", SURVEY §7 step 7: w = mask * sigmoid(conf)):
    weighted means / covariance / variance / rms against the oracle's weighted closed form; zero weights drop pairs;
    all-ones weights reproduce the unweighted kernel bit for bit; heavier weights on the clean half of the pairs pull
    the solution towards the truth."""
    from oracle import post_ref
    from pi3_slam_amd import ops
    c = _overlap(K, seed=11 + K, ragged=True)
    rng = np.random.default_rng(5)
    conf_r, conf_q = rng.standard_normal((OV, K)) * 2, rng.standard_normal((OV, K)) * 2
    sig = lambda z: (1.0 / (1.0 + np.exp(-z))).astype(np.float32)             # noqa: E731
    wr = sig(conf_r) * (rng.random((OV, K)) < 0.85)
    wq = sig(conf_q) * (rng.random((OV, K)) < 0.9)
    wr, wq = wr.astype(np.float32), wq.astype(np.float32)
    pr, pq = c["pr"].astype(np.float16), c["pq"].astype(np.float16)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)            # noqa: E731
    idx = ops.sim3_match_keypoints(t(c["kr"]), t(c["kq"]))
    ref = post_ref.align_chunks(pr, pq, c["kr"], c["kq"], c["pose"], use_filter, weights_ref=wr, weights_qry=wq)
    out = ops.sim3_umeyama(t(pr), t(pq), idx, t(c["pose"]), t(wr), t(wq), use_filter).cpu().numpy()
    assert ref["n_common"] > 1024
    _check(out, ref, use_filter)
    # one-sided weights
    ref1 = post_ref.align_chunks(pr, pq, c["kr"], c["kq"], c["pose"], use_filter, weights_qry=wq)
    out1 = ops.sim3_umeyama(t(pr), t(pq), idx, t(c["pose"]), None, t(wq), use_filter).cpu().numpy()
    _check(out1, ref1, use_filter)
    # all-ones weights == the unweighted entry point, bit for bit
    ones = torch.ones(OV, K, device=dev)
    a = ops.sim3_umeyama(t(pr), t(pq), idx, t(c["pose"]), ones, ones, use_filter).cpu().numpy()
    b = ops.sim3_umeyama(t(pr), t(pq), idx, t(c["pose"]), None, None, use_filter).cpu().numpy()
    assert np.array_equal(a, b)
    # weights matter: corrupt the low-weight pairs' qry points grossly; the weighted solve stays near the truth
    bad = wq < 0.3
    pq_bad = c["pq"].copy()
    pq_bad[bad] += 0.5 * rng.standard_normal((int(bad.sum()), 3))
    wq2 = np.where(bad, np.float32(1e-4), np.float32(1.0)).astype(np.float32)
    s_true = c["truth"][0]
    ow = ops.sim3_umeyama(t(pr), t(pq_bad.astype(np.float16)), idx, t(c["pose"]), None, t(wq2), use_filter).cpu().numpy()
    ou = ops.sim3_umeyama(t(pr), t(pq_bad.astype(np.float16)), idx, t(c["pose"]), None, None, use_filter).cpu().numpy()
    assert abs(ow[0] - s_true) < 0.1 * abs(ou[0] - s_true) and abs(ow[0] - s_true) < 2e-3


def test_weighted_entry_point_argument_errors(dev):
    from pi3_slam_amd import ops
    idx = torch.zeros(2, 4, dtype=torch.int32, device=dev)
    p = torch.zeros(2, 4, 3, dtype=torch.float16, device=dev)
    pose = torch.eye(4, device=dev)
    with pytest.raises(AssertionError):                                        # mixed kinds
        ops.sim3_umeyama(p, p, idx, pose, torch.ones(2, 4, dtype=torch.uint8, device=dev), torch.ones(2, 4, device=dev))
    with pytest.raises(AssertionError):                                        # wrong size
        ops.sim3_umeyama(p, p, idx, pose, None, torch.ones(2, 3, device=dev))


def _chunk(pts, kp, masks, conf, poses):
    return {"points": torch.from_numpy(pts), "keypoints": torch.from_numpy(kp), "masks": torch.from_numpy(masks),
            "conf": torch.from_numpy(conf), "camera_poses": torch.from_numpy(poses)}


@pytest.mark.parametrize("K", [200, 400])
@pytest.mark.parametrize("n_prev", [100, 93, 30])
def test_host_entry_points_with_a_short_previous_chunk(dev, K, n_prev):
    """alignment.estimate_sim3 and dist.relative_sim3_from_boundaries (the sequential and the chunk-parallel host path)
    at cl = 100, ov = 20 with a previous chunk of 100 / 93 / 30 views: create_view_graph_matches always uses the nominal
    chunk length (offline_reconstructor.py:98), so only the view pairs that exist take part (13 of 20 at n_prev = 93,
    none at 30 -> error).  Both must equal the oracle on exactly those views, with all three weightings."""
    from oracle import post_ref
    from pi3_slam_amd.alignment import create_view_graph_matches, estimate_sim3
    from pi3_slam_amd.dist import pack_boundary, relative_sim3_from_boundaries, unpack_boundary
    cl = 100
    c = _overlap(K, seed=21 + K + n_prev, ragged=True)
    rng = np.random.default_rng(9)
    n_pairs = max(0, n_prev - (cl - OV))
    # the previous chunk's views cl-ov .. n_prev-1 are overlap views 0 .. n_pairs-1
    pts_prev = (rng.standard_normal((n_prev, K, 3)) + [0, 0, 5]).astype(np.float16)
    kp_prev = (rng.random((n_prev, K, 2)) * 300).astype(np.float16)
    conf_prev = rng.standard_normal((n_prev, K, 1)).astype(np.float16)
    m_prev = rng.random((n_prev, K, 1)) < 0.8
    conf_cur = rng.standard_normal((cl, K, 1)).astype(np.float16)
    m_cur = rng.random((cl, K, 1)) < 0.8
    pts_cur = (rng.standard_normal((cl, K, 3)) + [0, 0, 5]).astype(np.float16)
    kp_cur = (rng.random((cl, K, 2)) * 300).astype(np.float16)
    if n_pairs:
        pts_prev[cl - OV:] = c["pr"][:n_pairs].astype(np.float16)
        kp_prev[cl - OV:] = c["kr"][:n_pairs]
    pts_cur[:OV] = c["pq"].astype(np.float16)
    kp_cur[:OV] = c["kq"]
    poses_prev = np.tile(np.eye(4, dtype=np.float32), (n_prev, 1, 1))
    poses_prev[-1] = c["pose"]
    prev = _chunk(pts_prev, kp_prev, m_prev, conf_prev, poses_prev)
    cur = _chunk(pts_cur, kp_cur, m_cur, conf_cur, np.tile(np.eye(4, dtype=np.float32), (cl, 1, 1)))
    matches = create_view_graph_matches(cl, OV)
    if n_pairs == 0:
        with pytest.raises(ValueError):
            estimate_sim3(prev, cur, matches, str(dev))
        return
    rsl, qsl = slice(cl - OV, n_prev), slice(0, n_pairs)
    sig = lambda z: 1.0 / (1.0 + np.exp(-z.astype(np.float32)))                # noqa: E731
    variants = {
        "plain": (dict(), dict()),
        "masks": (dict(use_masks=True), dict(w_ref=m_prev[rsl, :, 0], w_qry=m_cur[qsl, :, 0])),
        "conf": (dict(weights="conf"),
                 dict(weights_ref=torch.sigmoid(torch.from_numpy(conf_prev[rsl, :, 0]).float()).numpy() * m_prev[rsl, :, 0],
                      weights_qry=torch.sigmoid(torch.from_numpy(conf_cur[qsl, :, 0]).float()).numpy() * m_cur[qsl, :, 0])),
    }
    for name, (kw, okw) in variants.items():
        ref = post_ref.align_chunks(pts_prev[rsl], pts_cur[qsl], kp_prev[rsl], kp_cur[qsl], c["pose"], True, **okw)
        out = estimate_sim3(prev, cur, matches, str(dev), **kw).cpu().numpy()
        _check(out, ref, True)
        assert ref["n_common"] > (1024 if n_pairs == OV else 500), name
    # chunk-parallel path: boundary blocks (tail of prev, head of cur), the same numbers
    ref = post_ref.align_chunks(pts_prev[rsl], pts_cur[qsl], kp_prev[rsl], kp_cur[qsl], c["pose"], True)
    bp = unpack_boundary(pack_boundary(prev, OV, K, device=str(dev)), OV, K, n_frames=n_prev)
    bc = unpack_boundary(pack_boundary(cur, OV, K, device=str(dev)), OV, K, n_frames=cl)
    out = relative_sim3_from_boundaries(bp, bc, OV, str(dev), chunk_length=cl).cpu().numpy()
    _check(out, ref, True)
