"""GPU: the bundle-adjustment kernels (csrc/ba.hip) against the numpy oracle of the same restated
algorithm (oracle/ba_ref.py) on the same inputs.  Parity with pytheia / Ceres is UNPINNED (stated in both files); the
kernel-vs-oracle tolerance is fp64 round-off: costs to 1e-9 relative, parameters to 1e-7."""
import numpy as np
import pytest
import torch

from ba_problem import make_problem

pytestmark = pytest.mark.gpu


def _to_dev(pb, dev):
    N = len(pb["R"])
    rc = torch.from_numpy(np.concatenate([pb["R"].reshape(N, 9), pb["C"]], 1)).to(dev).contiguous()
    return (torch.from_numpy(pb["X"]).to(dev).contiguous(), rc, torch.from_numpy(pb["intr"]).to(dev).contiguous(),
            torch.from_numpy(pb["uv"]).to(dev).contiguous(), torch.from_numpy(pb["valid"]).to(dev).contiguous())


@pytest.fixture(scope="module")
def dev(built_lib):
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.mark.parametrize("case", ["clean_perturbed", "noise_huber", "priors"])
def test_bundle_adjust_matches_oracle(dev, case):
    from oracle import ba_ref
    from pi3_slam_amd import ops
    kw = dict(clean_perturbed=dict(N=6, K=9, seed=1, perturb=1.0),
              noise_huber=dict(N=7, K=10, seed=4, noise_px=0.8, outlier_frac=0.06, perturb=0.6),
              priors=dict(N=6, K=8, seed=9, noise_px=0.3, perturb=0.4))[case]
    pb = make_problem(**kw)
    N = len(pb["R"])
    # 6 iterations: the well-conditioned phase, where the Schur-complement solve on the device and the oracle's dense
    # solve of the full normal equations follow the same path step for step (later, with the trust region wide open, the
    # 7 gauge directions make the reduced system numerically singular and a failed factorisation - a rejected step in
    # both - can fall on different iterations; the long run below compares the optimum instead)
    huber, iters = (3.0, 6) if case == "priors" else (2.0, 6)
    prior = None
    pr = pc = pf = None
    if case == "priors":
        flag = np.zeros(N, np.uint8); flag[:3] = 1
        prior = dict(R=pb["R_gt"], C=pb["C_gt"] + 0.05, flag=flag, sqrt_info_rot=0.5 ** 0.5, sqrt_info_pos=0.2)
        pr = torch.from_numpy(pb["R_gt"].reshape(N, 9)).to(dev)
        pc = torch.from_numpy(prior["C"]).to(dev)
        pf = torch.from_numpy(flag).to(dev)
    R, C, X, s = ba_ref.bundle_adjust(pb["R"], pb["C"], pb["intr"], pb["X"], pb["uv"], pb["valid"], huber, iters, prior)
    pts, rc, intr, uv, valid = _to_dev(pb, dev)
    out = ops.bundle_adjust(pts, rc, intr, uv, valid, huber, iters, pr, pc, pf, 0.5 ** 0.5, 0.2).cpu().numpy()
    torch.cuda.synchronize()
    assert abs(out[8] - s["initial_cost"]) <= 1e-9 * s["initial_cost"]
    if case == "clean_perturbed":      # converges to the float32 rounding floor of the pixels: the step count there is noise
        assert out[0] < 1e-9 * out[8] and s["final_cost"] < 1e-9 * s["initial_cost"]
    else:
        assert int(out[5]) == s["iterations"] and int(out[6]) == s["accepted_steps"] and out[9] == 0, (out, s)
        assert abs(out[0] - s["final_cost"]) <= 1e-7 * max(s["final_cost"], 1e-12) + 1e-12, (out[0], s["final_cost"])
    assert out[0] < out[8]
    rc = rc.cpu().numpy()
    np.testing.assert_allclose(rc[:, :9].reshape(N, 3, 3), R, atol=1e-7)
    np.testing.assert_allclose(rc[:, 9:], C, atol=1e-7)
    np.testing.assert_allclose(pts.cpu().numpy(), X, atol=1e-6)
    Rm = rc[:, :9].reshape(N, 3, 3)
    assert np.abs(Rm @ Rm.transpose(0, 2, 1) - np.eye(3)).max() < 1e-12
    # outlier tracks on the refined values
    est = ops.ba_outlier_tracks(pts, torch.from_numpy(rc).to(dev), intr, uv, valid, 2.0, 0.25).cpu().numpy().reshape(-1)
    ref = ba_ref.outlier_tracks(Rm, rc[:, 9:], pb["intr"], pts.cpu().numpy(), pb["uv"], pb["valid"], 2.0, 0.25)
    assert np.array_equal(est, ref)
    # long run: same optimum
    _, _, _, s_long = ba_ref.bundle_adjust(pb["R"], pb["C"], pb["intr"], pb["X"], pb["uv"], pb["valid"], huber, 40, prior)
    pts, rc, intr, uv, valid = _to_dev(pb, dev)
    out = ops.bundle_adjust(pts, rc, intr, uv, valid, huber, 40, pr, pc, pf, 0.5 ** 0.5, 0.2).cpu().numpy()
    assert abs(out[0] - s_long["final_cost"]) <= 1e-4 * s_long["final_cost"] + 1e-9, (out[0], s_long["final_cost"])


def _compare_with_schur_oracle(dev, pb, huber, iters, prior=None, cost_rtol=1e-9, atol_pose=1e-7, atol_pts=1e-6,
                               homogeneous=False):
    """pi3_bundle_adjust (homogeneous: pi3_bundle_adjust_homogeneous) against oracle/ba_ref.bundle_adjust_schur on the same
    problem: same number of iterations and accepted steps, costs to cost_rtol, parameters to atol.  Returns (device
    summary, oracle summary)."""
    from oracle import ba_ref
    from pi3_slam_amd import ops
    N = len(pb["R"])
    R, C, X, s = ba_ref.bundle_adjust_schur(pb["R"], pb["C"], pb["intr"], pb["X"], pb["uv"], pb["valid"], huber, iters, prior,
                                            homogeneous=homogeneous)
    pts, rc, intr, uv, valid = _to_dev(pb, dev)
    pr = pc = pf = None
    if prior is not None:
        pr = torch.from_numpy(prior["R"].reshape(N, 9)).to(dev)
        pc = torch.from_numpy(prior["C"]).to(dev)
        pf = torch.from_numpy(prior["flag"]).to(dev)
    out = ops.bundle_adjust(pts, rc, intr, uv, valid, huber, iters, pr, pc, pf,
                            prior["sqrt_info_rot"] if prior else 0.0, prior["sqrt_info_pos"] if prior else 0.0,
                            homogeneous=homogeneous).cpu().numpy()
    torch.cuda.synchronize()
    assert abs(out[8] - s["initial_cost"]) <= cost_rtol * s["initial_cost"], (out[8], s["initial_cost"])
    assert (int(out[5]), int(out[6])) == (s["iterations"], s["accepted_steps"]), (out, s)
    assert abs(out[0] - s["final_cost"]) <= cost_rtol * s["final_cost"], (out[0], s["final_cost"])
    rc = rc.cpu().numpy()
    np.testing.assert_allclose(rc[:, :9].reshape(N, 3, 3), R, atol=atol_pose)
    np.testing.assert_allclose(rc[:, 9:], C, atol=atol_pose)
    np.testing.assert_allclose(pts.cpu().numpy(), X, atol=atol_pts)
    return out, s


@pytest.mark.parametrize("N,K", [(20, 12), (24, 12), (27, 10)])
def test_multi_panel_cholesky_and_schur_slices_match_the_oracle(dev, N, K):
    """6N = 120 / 144 / 162 unknowns = 3 / 3 / 4 panels of the blocked Cholesky (BA_NB = 48 columns; partial last panel
    of 24 / none / 18), i.e. ba_chol_panel, ba_chol_update and the multi-panel loops of ba_chol_solve all execute
    (with N <= 8 cameras the factorisation is one diagonal block), and the 32 source-frame slices of ba_schur_rows hold
    0 or 1 frame each.  One LM iteration (the step itself) and six (the trust-region sequence), step for step against
    the Schur-complement oracle; reference call: utils/chunk_reconstruction.py:188-219."""
    pb = make_problem(N=N, K=K, seed=100 + N, noise_px=0.5, outlier_frac=0.03, perturb=0.5)
    for iters in (1, 6):
        out, s = _compare_with_schur_oracle(dev, pb, 2.0, iters)
        assert out[9] == 0 and s["chol_failures"] == 0 and out[0] < out[8]


def test_multi_panel_with_pose_priors_matches_the_oracle(dev):
    """The prior-constrained form (utils/reconstruction_alignment.py:107-171: Huber 3.0, covariance 2 I / 25 I on the
    overlap views) at 3 panels."""
    N, K = 22, 10
    pb = make_problem(N=N, K=K, seed=31, noise_px=0.4, perturb=0.4)
    flag = np.zeros(N, np.uint8); flag[:5] = 1
    prior = dict(R=pb["R_gt"].copy(), C=pb["C_gt"] + 0.03, flag=flag, sqrt_info_rot=0.5 ** 0.5, sqrt_info_pos=0.2)
    _compare_with_schur_oracle(dev, pb, 3.0, 8, prior)


@pytest.mark.parametrize("iters,homogeneous", [(1, False), (10, False), (2, True)])
def test_whole_chunk_matches_the_oracle(dev, iters, homogeneous):
    """The shipped size: 100 cameras x 200 keypoints (600 unknowns = 12.5 panels, 20 000 tracks, ~1 M observations in
    the reference's pattern: every earlier frame + the next two) on bench.synthetic_ba_problem, one LM iteration and
    the per-chunk stage's ten, against the Schur-complement oracle.  Cost 1e-9; parameters 1e-7 after one step and
    1e-6 after ten (round-off of 1 M-term sums taken in different orders, amplified along the weakly constrained
    directions by ten solves)."""
    from bench import synthetic_ba_problem
    pb = synthetic_ba_problem(100, 200, seed=3, noise_px=0.5, perturb=1.0)
    tol = dict(atol_pose=1e-7, atol_pts=1e-6) if iters == 1 else dict(atol_pose=1e-6, atol_pts=1e-5)
    out, s = _compare_with_schur_oracle(dev, pb, 2.0, iters, homogeneous=homogeneous, **tol)   # (2, True): the product's default form
    assert out[9] == 0 and s["chol_failures"] == 0 and out[0] < 0.2 * out[8], (out, s)


def test_failed_factorisation_rejects_the_step_and_returns_the_inputs(dev):
    """Forced Cholesky failure: one NaN pixel poisons camera 0's block, the diagonal pivot is not > 0, chol_fail is
    raised, the step is rejected, the trust region shrinks (1e4 / 2 / 4 / 8 ...) until the loop gives up after 15
    iterations - exactly as the oracle - and points / poses come back bit for bit (also from the multi-panel path)."""
    from oracle import ba_ref
    from pi3_slam_amd import ops
    for N, K in ((5, 6), (20, 8)):
        pb = make_problem(N=N, K=K, seed=3, noise_px=0.3, perturb=0.3)
        pb["uv"] = pb["uv"].copy()
        assert pb["valid"][1, 0, 2]
        pb["uv"][1, 0, 2, 0] = np.nan
        _, _, _, s = ba_ref.bundle_adjust_schur(pb["R"], pb["C"], pb["intr"], pb["X"], pb["uv"], pb["valid"], 2.0, 30)
        pts, rc, intr, uv, valid = _to_dev(pb, dev)
        p0, r0 = pts.clone(), rc.clone()
        out = ops.bundle_adjust(pts, rc, intr, uv, valid, 2.0, 30).cpu().numpy()
        assert out[9] == 1 and int(out[6]) == 0 == s["accepted_steps"] and int(out[5]) == s["iterations"] == 15
        assert s["chol_failures"] == 15 and torch.equal(pts, p0) and torch.equal(rc, r0)


def test_sanity_gate_keeps_the_input_when_the_geometry_is_inconsistent(dev):
    """bundle_adjust_chunk must not take over a result that threw every track out or walked a camera away by more than
    the scene extent (LM 'succeeds' on contradictory observations: round 2 saw poses move by 10^3 units on recipe-weight
    chunks with success=True): the chunk stays as it was and the info says why."""
    from pi3_slam_amd.bundle_adjust import bundle_adjust_chunk
    N, K, W, H = 6, 30, 406, 308
    g = torch.Generator().manual_seed(0)
    poses = torch.eye(4).repeat(N, 1, 1)
    poses[:, :3, 3] = torch.randn(N, 3, generator=g) * 0.1
    chunk = dict(points=(torch.randn(N, K, 3, generator=g) + torch.tensor([0.0, 0.0, 4.0])).half(), camera_poses=poses,
                 keypoints=(torch.rand(N, K, 2, generator=g) * torch.tensor([W - 1.0, H - 1.0])).half(),   # unrelated pixels
                 masks=torch.ones(N, K, 1, dtype=torch.bool))
    before = {k: v.clone() for k, v in chunk.items()}
    info = bundle_adjust_chunk(chunk, W, H, 5, str(dev))
    assert not info["success"] and "rejected" in info, info
    for k in before:
        assert torch.equal(chunk[k], before[k]), k
    assert "track_estimated" not in chunk


def test_bundle_adjust_is_deterministic_and_zero_iterations_is_identity(dev):
    from pi3_slam_amd import ops
    pb = make_problem(N=8, K=12, seed=6, noise_px=0.5, perturb=0.5)
    outs = []
    for _ in range(3):
        pts, rc, intr, uv, valid = _to_dev(pb, dev)
        ops.bundle_adjust(pts, rc, intr, uv, valid, 2.0, 10)
        outs.append((pts.cpu().clone(), rc.cpu().clone()))
    assert all(torch.equal(outs[0][0], o[0]) and torch.equal(outs[0][1], o[1]) for o in outs[1:])   # fixed-order sums
    pts, rc, intr, uv, valid = _to_dev(pb, dev)
    p0, r0 = pts.clone(), rc.clone()
    s = ops.bundle_adjust(pts, rc, intr, uv, valid, 2.0, 0).cpu()
    assert torch.equal(pts, p0) and torch.equal(rc, r0) and s[5] == 0 and s[0] == s[8] > 0


def test_chunk_bundle_adjust_and_reconstructor_flag(dev, tmp_path):
    """bundle_adjust_chunk on a chunk dict whose observations are consistent by construction (they are projections of
    the chunk's own points through its own poses, as in the reference): a fixed point up to fp16 / float32 rounding;
    then perturbed poses are pulled back.  Finally the reconstructor with bundle_adjust=True runs both BA stages."""
    import json
    import os
    from pi3_slam_amd.bundle_adjust import bundle_adjust_chunk
    from pi3_slam_amd.reconstructor import OfflineReconstructor
    rng = np.random.default_rng(0)
    N, K, W, H = 6, 40, 406, 308
    pb = make_problem(N=N, K=K, seed=2)
    poses = np.tile(np.eye(4, dtype=np.float32), (N, 1, 1))
    poses[:, :3, :3] = pb["R_gt"].transpose(0, 2, 1)
    poses[:, :3, 3] = pb["C_gt"]
    K3 = np.zeros((N, 3, 3), np.float32)
    K3[:, 0, 0], K3[:, 1, 1], K3[:, 0, 2], K3[:, 1, 2], K3[:, 2, 2] = pb["intr"][:, 0], pb["intr"][:, 1], pb["intr"][:, 2], pb["intr"][:, 3], 1
    idx = np.arange(N)
    chunk = dict(points=torch.from_numpy(pb["X_gt"].reshape(N, K, 3)).half(), camera_poses=torch.from_numpy(poses),
                 intrinsics=torch.from_numpy(K3), keypoints=torch.from_numpy(pb["uv"][idx, idx]).half(),
                 masks=torch.ones(N, K, 1, dtype=torch.bool), colors=torch.full((N, K, 3), 99.0).half())
    a = {k: v.clone() for k, v in chunk.items()}
    info = bundle_adjust_chunk(a, W, H, 5, str(dev))
    assert info["success"] and info["final_cost"] <= info["initial_cost"] and info["initial_cost"] < 0.5 * N * K
    assert (a["camera_poses"] - chunk["camera_poses"]).abs().max() < 5e-3 and a["points"].dtype == torch.float32
    assert a["track_estimated"].shape == (N, K)
    b = {k: v.clone() for k, v in chunk.items()}
    b["camera_poses"][:, :3, 3] += torch.from_numpy(0.02 * rng.standard_normal((N, 3))).float()
    info_b = bundle_adjust_chunk(b, W, H, 5, str(dev), settings=dict(max_iters=30, huber_width=2.0,
                                                                     max_reprojection_px=2.0, min_triangulation_angle_deg=0.25))
    # (the projected observations are generated from the perturbed poses themselves, as in the reference: only the
    # keypoint observations disagree, so the optimum is a compromise, not zero)
    assert info_b["success"] and info_b["final_cost"] < 0.9 * info_b["initial_cost"] and info_b["accepted_steps"] >= 1
    # reconstructor: two overlapping chunks cut from the same scene
    os.makedirs(tmp_path / "chunks")
    for c, sl in enumerate((slice(0, 4), slice(2, 6))):
        d = {k: v[sl].clone() for k, v in chunk.items()}
        d.update(image_paths=[[f"img_{i:03d}.png"] for i in range(sl.start, sl.stop)], original_width=W, original_height=H,
                 chunk_index=c)
        torch.save(d, tmp_path / "chunks" / f"chunk_{c:06d}.pt")
    json.dump({"chunk_length": 4, "overlap": 2, "target_size": [H, W]}, open(tmp_path / "chunk_metadata.json", "w"))
    traj = {}
    for flag in (False, True):
        rec = OfflineReconstructor(str(tmp_path), str(tmp_path / f"out{int(flag)}"), bundle_adjust=flag)
        rec.run()
        traj[flag] = np.loadtxt(tmp_path / f"out{int(flag)}" / "trajectory_tum.txt")
        if flag:
            assert all(i is None or i.get("bundle_adjustment", {}).get("success") for i in rec.alignment_infos)
            assert rec.ba_infos and all(i["success"] for i in rec.ba_infos)
    assert traj[True].shape == traj[False].shape == (6, 8)
    assert np.abs(traj[True][:, 1:4] - traj[False][:, 1:4]).max() < 2e-2     # consistent data: BA only polishes
    assert np.abs(traj[True][:, 1:4] - pb["C_gt"]).max() < 2e-2
    # the reference's --use-inverse-depth: both stages run with one inverse depth per track
    rec = OfflineReconstructor(str(tmp_path), str(tmp_path / "out_id"), bundle_adjust=True, use_inverse_depth=True)
    rec.run()
    tid = np.loadtxt(tmp_path / "out_id" / "trajectory_tum.txt")
    assert rec.ba_infos and all(i["success"] for i in rec.ba_infos)
    assert all(i is None or i.get("bundle_adjustment", {}).get("success") for i in rec.alignment_infos)
    assert rec.refinement_summary["per_chunk_bundle_adjust"]["applied"] == 2
    assert tid.shape == (6, 8) and np.abs(tid[:, 1:4] - pb["C_gt"]).max() < 2e-2


@pytest.mark.parametrize("N,K,iters,with_prior", [(6, 10, 6, False), (9, 14, 4, True), (22, 12, 3, True), (27, 10, 2, False)])
def test_homogeneous_point_parametrization_matches_the_oracle(dev, N, K, iters, with_prior):
    """pi3_bundle_adjust_homogeneous (the point parametrization the reference's calls configure:
    tracks step in the tangent space of their 4-vector, ceres::HomogeneousVectorParameterization) against the oracle's
    Schur form with the same parametrization: step for step in the well-conditioned phase, small and multi-panel
    camera systems, with and without pose priors.  The product's bundle_adjust_chunk uses this form by default."""
    pb = make_problem(N=N, K=K, seed=31 + N, noise_px=0.5, outlier_frac=0.04, perturb=0.7)
    prior = None
    if with_prior:
        flag = np.zeros(N, np.uint8)
        flag[: max(2, N // 4)] = 1
        prior = dict(R=pb["R_gt"], C=pb["C_gt"] + 0.03, flag=flag, sqrt_info_rot=0.5 ** 0.5, sqrt_info_pos=0.2)
    out_h, s_h = _compare_with_schur_oracle(dev, pb, 3.0 if with_prior else 2.0, iters, prior, cost_rtol=1e-9, atol_pose=1e-7,
                                            atol_pts=1e-6, homogeneous=True)
    out_e, s_e = _compare_with_schur_oracle(dev, pb, 3.0 if with_prior else 2.0, iters, prior, homogeneous=False)
    assert out_h[0] < out_h[8] and out_e[0] < out_e[8]
    assert out_h[0] != out_e[0]                       # a different path (the same optimum: the 50-iteration test below)


def test_homogeneous_and_euclidean_settings_of_the_chunk_adjuster(dev):
    """bundle_adjust_chunk: homogeneous_points=True (default, = the reference's configuration) and False reach the same
    cost to 1e-4 on a consistent chunk after the after-alignment setting's 50 iterations; zero iterations is the identity
    in both."""
    from pi3_slam_amd import ops
    pb = make_problem(N=8, K=12, seed=5, noise_px=0.4, perturb=0.6)
    N = 8
    costs = {}
    for hom in (True, False):
        pts, rc, intr, uv, valid = _to_dev(pb, dev)
        out = ops.bundle_adjust(pts, rc, intr, uv, valid, 3.0, 50, homogeneous=hom).cpu().numpy()
        costs[hom] = out[0]
        p0, r0, _, _, _ = _to_dev(pb, dev)
        p1, r1 = p0.clone(), r0.clone()
        ops.bundle_adjust(p1, r1, intr, uv, valid, 3.0, 0, homogeneous=hom)
        assert torch.equal(p0, p1) and torch.equal(r0, r1)
    assert abs(costs[True] - costs[False]) <= 1e-4 * costs[False], costs


@pytest.mark.parametrize("N,K,iters,with_prior,noise", [(5, 8, 4, False, 0.5), (7, 10, 6, True, 0.3), (12, 9, 3, False, 1.0),
                                                        (26, 8, 2, True, 0.5)])
def test_inverse_depth_parametrization_matches_the_oracle(dev, N, K, iters, with_prior, noise):
    """pi3_bundle_adjust_inverse_depth (the reference's --use-inverse-depth: InitializeInverseDepth +
    use_inverse_depth_parametrization, utils/chunk_reconstruction.py:187-204) against oracle/ba_ref.
    bundle_adjust_inverse_depth (dense normal equations over 6 N + N K unknowns; Jacobians checked by finite differences
    in tests/test_ba_oracle.py): the same iterations and accepted steps, cost to 1e-9, cameras to 1e-7, the returned
    Euclidean points (on the rays of their reference keypoints) to 1e-6 - small and multi-panel camera systems, with and
    without pose priors, Huber active."""
    from oracle import ba_ref
    from pi3_slam_amd import ops
    pb = make_problem(N=N, K=K, seed=77 + N, noise_px=noise, outlier_frac=0.04, perturb=0.6)
    huber = 3.0 if with_prior else 2.0
    prior = None
    pr = pc = pf = None
    if with_prior:
        flag = np.zeros(N, np.uint8)
        flag[: max(2, N // 4)] = 1
        prior = dict(R=pb["R_gt"], C=pb["C_gt"] + 0.03, flag=flag, sqrt_info_rot=0.5 ** 0.5, sqrt_info_pos=0.2)
        pr, pc, pf = (torch.from_numpy(pb["R_gt"].reshape(N, 9)).to(dev), torch.from_numpy(prior["C"]).to(dev),
                      torch.from_numpy(flag).to(dev))
    R, C, X, s = ba_ref.bundle_adjust_inverse_depth(pb["R"], pb["C"], pb["intr"], pb["X"], pb["uv"], pb["valid"], huber, iters,
                                                    prior)
    pts, rc, intr, uv, valid = _to_dev(pb, dev)
    out = ops.bundle_adjust(pts, rc, intr, uv, valid, huber, iters, pr, pc, pf, 0.5 ** 0.5 if prior else 0.0,
                            0.2 if prior else 0.0, inverse_depth=True).cpu().numpy()
    torch.cuda.synchronize()
    assert abs(out[8] - s["initial_cost"]) <= 1e-9 * s["initial_cost"], (out[8], s["initial_cost"])
    assert (int(out[5]), int(out[6])) == (s["iterations"], s["accepted_steps"]) and out[9] == 0, (out, s)
    assert abs(out[0] - s["final_cost"]) <= 1e-8 * s["final_cost"] + 1e-12, (out[0], s["final_cost"])
    assert out[0] < out[8]
    rcn = rc.cpu().numpy()
    np.testing.assert_allclose(rcn[:, :9].reshape(N, 3, 3), R, atol=1e-7)
    np.testing.assert_allclose(rcn[:, 9:], C, atol=1e-7)
    np.testing.assert_allclose(pts.cpu().numpy(), X, atol=1e-6)
    # the returned points sit on the rays of their reference keypoints: reprojection into the reference view = the keypoint
    Xn = pts.cpu().numpy().reshape(N, K, 3)
    for s_ in range(N):
        p = (Xn[s_] - rcn[s_, 9:]) @ rcn[s_, :9].reshape(3, 3).T
        u = pb["intr"][s_, 0] * p[:, 0] / p[:, 2] + pb["intr"][s_, 2]
        v = pb["intr"][s_, 1] * p[:, 1] / p[:, 2] + pb["intr"][s_, 3]
        assert np.abs(u - pb["uv"][s_, s_, :, 0]).max() < 1e-6 and np.abs(v - pb["uv"][s_, s_, :, 1]).max() < 1e-6
    # zero iterations: only the snap (InitializeInverseDepth), poses untouched
    pts0, rc0, _, _, _ = _to_dev(pb, dev)
    ops.bundle_adjust(pts0, rc0, intr, uv, valid, huber, 0, inverse_depth=True)
    b, rho, anchor = ba_ref.inverse_depth_state(pb["R"], pb["C"], pb["intr"], pb["X"], pb["uv"], pb["valid"])
    np.testing.assert_allclose(pts0.cpu().numpy(), ba_ref.inverse_depth_points(pb["R"], pb["C"], b, rho, anchor), atol=1e-12)
    assert torch.equal(rc0, _to_dev(pb, dev)[1])
