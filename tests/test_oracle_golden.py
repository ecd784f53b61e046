"""CPU: the oracle restatements against the vectors produced by the REAL reference (oracle/gen_golden*.py).
This is what pins the oracle; the GPU tests then compare the HIP path with the oracle / the same vectors."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from oracle import pi3_ref, post_ref
from oracle.gen_golden import CASES, golden_images
from oracle.gen_golden_post import CASES as POST_CASES, synthetic_chunk
from pi3_slam_amd.weights import Pi3Config, recipe_state_dict_cpu


@pytest.fixture(scope="module")
def full_sd():
    return recipe_state_dict_cpu(Pi3Config())   # 958.7 M parameters, ~1 min of numpy


@pytest.mark.parametrize("name", ["pi3_tiny_a", "pi3_tiny_b"])
def test_pi3_oracle_matches_reference_vectors(full_sd, name):
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    B, N, H, W = CASES[name]
    out = pi3_ref.pi3_forward(full_sd, golden_images(name, B, N, H, W), Pi3Config(), return_intermediates=True)
    for k in ("points", "local_points", "conf", "camera_poses"):
        np.testing.assert_allclose(out[k].numpy(), g[k], rtol=1e-4, atol=2e-5, err_msg=k)
    for k in g.files:
        if k.startswith("i_"):
            np.testing.assert_allclose(out["_intermediates"][k[2:]].numpy(), g[k], rtol=1e-4, atol=5e-5, err_msg=k)


@pytest.mark.parametrize("name", list(POST_CASES))
def test_post_oracle_matches_reference_vectors(name):
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    N, H, W = POST_CASES[name]
    d = synthetic_chunk(name, N, H, W)
    masks = post_ref.compute_masks(d["conf"], d["local_points"])
    assert np.array_equal(masks.numpy(), g["masks"])
    s = post_ref.scale_factor(d["moge_depth"], d["local_points"][0][..., 2], masks[0])
    assert np.float32(s.item()) == g["scale"]
    for tag, max_kp, seed in (("full", 4096, None), ("sub", 12, 1234)):
        gen = torch.Generator().manual_seed(seed) if seed is not None else None
        kp = post_ref.grid_keypoints(N, H, W, max_kp, gen)
        assert np.array_equal(kp.numpy(), g[f"kp_{tag}"]), "keypoints must be bit-exact"
        it = post_ref.interpolate_at_keypoints(d["points"], d["local_points"], d["conf"], masks, kp, H, W)
        assert np.array_equal(it["points"].numpy(), g[f"ipoints_{tag}"])
        assert np.array_equal(it["local_points"].numpy(), g[f"ilocal_{tag}"])
        assert np.array_equal(it["conf"].numpy(), g[f"iconf_{tag}"])
        assert np.array_equal(it["masks"].numpy(), g[f"imasks_{tag}"])
        assert np.array_equal(post_ref.keypoint_colors(d["images"], kp).numpy(), g[f"colors_{tag}"])
    conf_masks = torch.sigmoid(d["conf"][..., 0]) > 0.1
    focal, shift = post_ref.recover_focal_shift(d["local_points"], conf_masks)
    np.testing.assert_allclose(focal.numpy(), g["focal"], rtol=1e-6)
    np.testing.assert_allclose(shift.numpy(), g["shift"], rtol=1e-6, atol=1e-7)


def test_layout_oracle_matches_reference_vectors():
    g = np.load(os.path.join(GOLDEN, "post_layout.npz"))
    for k in g.files:
        parts = k.split("_")
        if parts[0] == "chunks":
            n, cl, ov = map(int, parts[1:])
            assert np.array_equal(np.array(post_ref.chunk_indices(n, cl, ov)).reshape(-1, 2), g[k]), k
        else:
            cl, ov = map(int, parts[1:])
            assert np.array_equal(np.array(post_ref.create_view_graph_matches(cl, ov)).reshape(-1, 2), g[k]), k


def test_umeyama_known_answers():
    """Sim(3) closed form: exact recovery of a known similarity, reflection handling, noise robustness.
    (No reference vector exists for this step — pytheia is absent — so these are known-answer tests.)"""
    rng = np.random.default_rng(1)
    x = rng.standard_normal((500, 3)) * 2.0
    ang = 0.7
    R = np.array([[np.cos(ang), -np.sin(ang), 0], [np.sin(ang), np.cos(ang), 0], [0, 0, 1.0]])
    R = R @ np.array([[1, 0, 0], [0, np.cos(0.3), -np.sin(0.3)], [0, np.sin(0.3), np.cos(0.3)]])
    s, t = 1.7, np.array([0.3, -2.0, 5.0])
    y = s * (R @ x.T).T + t
    s2, R2, t2, M = post_ref.umeyama(x, y)
    assert abs(s2 - s) < 1e-12 and np.abs(R2 - R).max() < 1e-12 and np.abs(t2 - t).max() < 1e-11
    y_noisy = y + 1e-3 * rng.standard_normal(y.shape)
    s3, R3, t3, _ = post_ref.umeyama(x, y_noisy)
    assert abs(s3 - s) < 1e-3 and np.abs(R3 - R).max() < 1e-3
    # planar, mirrored data must still return a proper rotation
    xp = x.copy(); xp[:, 2] = 0
    yp = xp.copy(); yp[:, 0] *= -1
    _, Rp, _, _ = post_ref.umeyama(xp, yp)
    assert abs(np.linalg.det(Rp) - 1.0) < 1e-9


def test_umeyama_rotation_equals_scipys_kabsch():
    """A third implementation nobody here wrote: scipy's `Rotation.align_vectors` (Kabsch with weights and the reflection
    fix) on the centred point sets must return the oracle's rotation, weighted and unweighted, on noisy, planar and
    mirrored data; scale and translation then follow from the closed form's own definitions (s = <R xc, yc> / |xc|^2,
    t = my - s R mx), checked here as identities."""
    from scipy.spatial.transform import Rotation
    rng = np.random.default_rng(11)
    for case in range(12):
        n = int(rng.integers(4, 200))
        x = rng.standard_normal((n, 3)) * rng.uniform(0.2, 3.0, 3) + rng.uniform(-2, 2, 3)
        if case % 4 == 1:
            x[:, 2] = x[:, 0] * 0.3 - 1.0                               # planar
        R0 = Rotation.from_rotvec(rng.standard_normal(3)).as_matrix()
        y = rng.uniform(0.3, 3.0) * x @ R0.T + rng.uniform(-3, 3, 3) + 0.02 * rng.standard_normal(x.shape)
        if case % 4 == 2:
            y[:, 0] *= -1.0                                             # mirrored: the best PROPER rotation is wanted
        w = rng.random(n) + 0.05 if case % 2 else None
        s, R, t, _ = post_ref.umeyama(x, y, w)
        ww = np.ones(n) if w is None else w
        mx, my = (ww[:, None] * x).sum(0) / ww.sum(), (ww[:, None] * y).sum(0) / ww.sum()
        xc, yc = x - mx, y - my
        Rs = Rotation.align_vectors(yc, xc, weights=ww)[0].as_matrix()   # yc ~ Rs xc
        np.testing.assert_allclose(R, Rs, atol=1e-9)
        s_def = (ww * ((xc @ R.T) * yc).sum(1)).sum() / (ww * (xc ** 2).sum(1)).sum()
        assert abs(s - s_def) <= 1e-12 * abs(s_def)
        np.testing.assert_allclose(t, my - s * R @ mx, atol=1e-12)


def test_weighted_umeyama_known_answers():
    """The weighted closed form: integer weights equal repeated points; it is the minimiser of the
    weighted cost (scipy on the 7 similarity parameters cannot do better); uniform weights change nothing; and in
    align_chunks pairs of weight 0 drop out while the near-half filter stays the unweighted median."""
    from scipy.optimize import least_squares
    from scipy.spatial.transform import Rotation
    rng = np.random.default_rng(3)
    x = rng.standard_normal((40, 3)) * [2.0, 1.0, 0.5] + [0, 0, 4.0]
    R = Rotation.from_rotvec([0.2, -0.4, 0.1]).as_matrix()
    y = 1.4 * x @ R.T + [0.3, 0.1, -0.7] + 0.05 * rng.standard_normal(x.shape)
    w = rng.integers(1, 5, 40).astype(np.float64)
    s1, R1, t1, M1 = post_ref.umeyama(x, y, w)
    rep = np.repeat(np.arange(40), w.astype(int))
    s2, R2, t2, M2 = post_ref.umeyama(x[rep], y[rep])
    np.testing.assert_allclose(M1, M2, atol=1e-12)
    np.testing.assert_allclose(post_ref.umeyama(x, y, np.full(40, 0.37))[3], post_ref.umeyama(x, y)[3], atol=1e-12)
    wr = rng.random(40) + 0.05

    def res(p):
        Rm = Rotation.from_rotvec(p[:3]).as_matrix()
        return (np.sqrt(wr)[:, None] * (np.exp(p[3]) * x @ Rm.T + p[4:7] - y)).ravel()
    s3, R3, t3, _ = post_ref.umeyama(x, y, wr)
    p0 = np.concatenate([Rotation.from_matrix(R3).as_rotvec(), [np.log(s3)], t3])
    sol = least_squares(res, p0, xtol=1e-14, ftol=1e-14, gtol=1e-14)
    assert np.abs(sol.x - p0).max() < 1e-6 and (res(p0) ** 2).sum() <= (res(sol.x) ** 2).sum() * (1 + 1e-9)
    # align_chunks: zero weights drop pairs, the filter is the unweighted strict median over what is left
    ov, K = 3, 30
    kp = (rng.random((ov, K, 2)) * 300).astype(np.float16)
    pr = (rng.standard_normal((ov, K, 3)) + [0, 0, 5]).astype(np.float16)
    pq = ((pr.astype(np.float64) - [0.1, 0.2, 0.3]) / 1.1).astype(np.float16)
    wq = rng.random((ov, K)).astype(np.float32)
    wq[:, ::3] = 0.0
    pose = np.eye(4, dtype=np.float32)
    a = post_ref.align_chunks(pr, pq, kp, kp, pose, True, weights_qry=wq)
    assert a["n_common"] == ov * K - ov * len(range(0, K, 3))
    valid = (wq > 0)
    b = post_ref.align_chunks(pr, pq, kp, kp, pose, True, w_qry=valid)
    assert a["median"] == b["median"] and a["n_used"] == b["n_used"]
    assert np.abs(a["M"] - b["M"]).max() > 1e-9                   # the weights do enter the solve
    few = post_ref.align_chunks(pr[:1, :2], pq[:1, :2], kp[:1, :2], kp[:1, :2], pose, False)
    assert few["n_used"] == 2 and np.array_equal(few["M"], np.eye(4))


@pytest.mark.parametrize("name", ["moge_small", "moge_chunk", "moge_pinhole_small", "moge_pinhole_chunk",
                                  "moge_var_pixelshuffle", "moge_var_interp", "moge_var_elu", "moge_vitl",
                                  "moge_vitb_reg"])
def test_moge_oracle_matches_reference_vectors(name):
    """MoGe-2 restatement vs the real MoGeModel class (synthetic model_config + recipe weights; the 'pinhole' cases
    edit a few 1x1 convolutions so the predicted map is camera-consistent: focal > 0, well-conditioned shift)."""
    from oracle import moge_ref
    from oracle.gen_golden_moge import CASES as MCASES, case_config, case_state_dict, moge_image
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    H, W, level = MCASES[name]
    out = moge_ref.moge_infer(case_state_dict(name), case_config(name), moge_image(name, H, W), level)
    if "pinhole" in name:
        assert g["focal_shift"][0] > 0.5 and abs(float(out["focal"]) - g["focal_shift"][0]) < 1e-4
        assert abs(float(out["shift"]) - g["focal_shift"][1]) < 1e-4
    mask = np.unpackbits(g["mask"])[: H * W].reshape(H, W).astype(bool)
    assert np.array_equal(out["mask"].numpy(), mask)
    np.testing.assert_allclose(out["points_affine"][..., 2].numpy(), g["points_affine_z"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(out["metric_scale"].numpy(), g["metric_scale"][0], rtol=1e-5)
    assert np.all(np.isinf(out["depth"].numpy()[~mask]))
    if "_var_" in name:       # config-space variants pin the NETWORK (their random maps make the focal solve degenerate)
        return
    np.testing.assert_allclose(out["depth"].numpy()[mask], g["depth"][mask], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(out["intrinsics"].numpy(), g["intrinsics"], rtol=1e-4)


def test_projection_oracle_matches_reference():
    """Observation projection restatement vs the reference's own ChunkPTRecon methods (gen_golden_post.py)."""
    from oracle import post_ref
    g = np.load(os.path.join(GOLDEN, "post_proj.npz"))
    N, K, W, H, max_after = [int(v) for v in g["shape"]]
    uv, valid = post_ref.project_observations(g["points"], g["poses"], g["intrinsics"], W, H, max_after)
    assert np.array_equal(valid, g["valid"])
    np.testing.assert_allclose(uv, g["uv"], rtol=0, atol=1e-9)
    assert valid.any() and not valid.all()


@pytest.mark.parametrize("name", ["ingest_down", "ingest_up", "ingest_mixed"])
def test_ingest_oracle_and_host_tables_match_pillow_vectors(name):
    """§8f rank 2: the resample restatement and the product's vectorised tap tables against Pillow-generated vectors."""
    from oracle import ingest_ref
    from pi3_slam_amd.image_io import resample_coeffs
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    H1, W1 = [int(v) for v in g["target"]]
    got = np.stack([ingest_ref.resize_bilinear_u8(f, (H1, W1)) for f in g["frames"]])
    assert np.array_equal(got, g["resized"])
    assert np.array_equal(ingest_ref.ingest_frames(g["frames"], (H1, W1)), g["tensor"])
    for n_in, n_out in [(g["frames"].shape[2], W1), (g["frames"].shape[1], H1), (752, 406), (480, 308), (7, 15)]:
        b0, k0 = ingest_ref.resample_coeffs(n_in, n_out)
        b1, k1 = resample_coeffs(n_in, n_out)
        assert np.array_equal(b0, b1) and np.array_equal(k0, k1)


# ------------------------------------------------------------------------------------------------ Sim(3): two closed forms
def _sim3_cloud(rng, n, kind):
    x = rng.standard_normal((n, 3)) * np.array([2.0, 1.0, 0.5]) + np.array([0.3, -0.2, 4.0])
    if kind == "planar":
        x[:, 2] = 4.0 + 0.3 * x[:, 0] - 0.1 * x[:, 1]
    ang = 0.7
    R = np.array([[np.cos(ang), -np.sin(ang), 0], [np.sin(ang), np.cos(ang), 0], [0, 0, 1.0]]) @ \
        np.array([[1, 0, 0], [0, np.cos(0.3), -np.sin(0.3)], [0, np.sin(0.3), np.cos(0.3)]])
    y = 1.7 * x @ R.T + np.array([0.5, 2.0, -1.0])
    if kind == "mirrored":           # the data is a reflection: the best PROPER rotation is wanted (det fix / quaternion)
        y = y * np.array([1.0, 1.0, -1.0])
    if kind in ("noisy", "mirrored", "planar"):
        y = y + 1e-2 * rng.standard_normal(y.shape)
    return x, y, (1.7, R)


@pytest.mark.parametrize("kind", ["exact", "noisy", "mirrored", "planar"])
@pytest.mark.parametrize("n", [3, 4, 50, 4000])
def test_sim3_umeyama_equals_horn_quaternion(kind, n):
    """Sim(3) parity is UNPINNED (pytheia absent, the reference holds no vectors): the SVD form (Umeyama 1991) and the
    quaternion form (Horn 1987) are derived independently and must agree to 1e-10 wherever the optimum is unique."""
    rng = np.random.default_rng(n * 7 + len(kind))
    x, y, (s0, R0) = _sim3_cloud(rng, n, kind)
    s1, R1, t1, M1 = post_ref.umeyama(x, y)
    s2, R2, t2, M2 = post_ref.horn_sim3(x, y)
    for R in (R1, R2):
        np.testing.assert_allclose(R @ R.T, np.eye(3), atol=1e-12)
        assert abs(np.linalg.det(R) - 1.0) < 1e-12
    np.testing.assert_allclose(M1, M2, atol=1e-10, rtol=1e-10)
    assert abs(s1 - s2) < 1e-10
    if kind == "exact":
        assert abs(s1 - s0) < 1e-10 and np.abs(R1 - R0).max() < 1e-10


def test_sim3_closed_forms_on_collinear_points():
    """Collinear points leave the rotation about the line free: both forms must still return a proper rotation and
    reach the same (minimal) residual, scale and mapped points."""
    rng = np.random.default_rng(0)
    lam = rng.standard_normal(40)
    x = np.outer(lam, [1.0, 2.0, -0.5]) + np.array([0.1, 0.2, 3.0])
    y = 0.8 * np.outer(lam, [0.0, 1.0, 1.0]) / np.sqrt(2) * np.linalg.norm([1.0, 2.0, -0.5]) + np.array([1.0, 0.0, 0.0])
    outs = [post_ref.umeyama(x, y), post_ref.horn_sim3(x, y)]
    for s, R, t, M in outs:
        assert abs(np.linalg.det(R) - 1.0) < 1e-9 and np.abs(R @ R.T - np.eye(3)).max() < 1e-9
        assert abs(s - 0.8) < 1e-9
        np.testing.assert_allclose(s * x @ R.T + t, y, atol=1e-9)


@pytest.mark.parametrize("tag", ["d64", "d32"])
def test_rope_2d_cpu_restatement_matches_the_reference_rope2d_class(tag):
    """`oracle.pi3_ref.rope_2d_cpu` restates the CPU branch of the reference's native FFI entry (curope.cpp:11-47);
    tests/golden/rope2d.npz holds what the reference's OWN torch `RoPE2D` (pos_embed.py:112-159, the class it runs when
    the extension is absent) returns on the same tokens (oracle/gen_golden_rope.py).  SURVEY §8 a5: the two agree to
    1e-5; so does the oracle's table form `rope2d`, and fwd = -1 undoes fwd = +1 (curope2d.py:24-29)."""
    g = np.load(os.path.join(GOLDEN, "rope2d.npz"))
    tok, pos, want = g[tag + "_tokens"], g[tag + "_positions"], g[tag + "_out"]            # tokens (B, heads, N, D)
    got = np.ascontiguousarray(tok.transpose(0, 2, 1, 3)).copy()                              # the FFI layout (B, N, H, D)
    pi3_ref.rope_2d_cpu(got, pos, 100.0, 1.0)
    np.testing.assert_allclose(got.transpose(0, 2, 1, 3), want, rtol=0, atol=1e-5)
    np.testing.assert_allclose(pi3_ref.rope2d(torch.from_numpy(tok), torch.from_numpy(pos)).numpy(), want, rtol=0, atol=1e-5)
    pi3_ref.rope_2d_cpu(got, pos, 100.0, -1.0)
    np.testing.assert_allclose(got.transpose(0, 2, 1, 3), tok, rtol=0, atol=1e-5)
    # position 0 (pi3's special tokens) is the identity
    assert np.array_equal(want[0, :, 0], tok[0, :, 0])
