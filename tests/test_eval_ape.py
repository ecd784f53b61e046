"""tools/eval_ape.py = `evo_ape tum REF EST -as` (scripts/eval_7scenes.sh:175): timestamp association, Sim(3) Umeyama
alignment, translation RMSE.  Checked on the reference's own ground-truth file for 7-Scenes heads seq-01
(tests/golden/gt_7scenes_heads.txt = /root/reference/scripts/groundtruths/7scenes/heads.txt, a data fixture) and on synthetic
trajectories with known answers; the Umeyama step also against the oracle's independent restatement."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import eval_ape  # noqa: E402

GT = os.path.join(ROOT, "tests", "golden", "gt_7scenes_heads.txt")


def _write_tum(path, stamps, pos, quat=None, fmt="{:d}"):
    quat = np.tile([0, 0, 0, 1.0], (len(pos), 1)) if quat is None else quat
    with open(path, "w") as f:
        f.write("# timestamp tx ty tz qx qy qz qw\n")
        for s, p, q in zip(stamps, pos, quat):
            f.write(" ".join([fmt.format(s)] + [f"{v:.6f}" for v in list(p) + list(q)]) + "\n")


def _sim(rng):
    from scipy.spatial.transform import Rotation
    R = Rotation.from_rotvec(rng.standard_normal(3)).as_matrix()
    return R, rng.standard_normal(3) * 2, float(np.exp(rng.standard_normal() * 0.5))


def test_groundtruth_file_reads_as_the_reference_wrote_it():
    s, p, q = eval_ape.read_tum(GT)
    assert len(s) == 1000 and np.array_equal(s, np.arange(1000)) and p.shape == (1000, 3) and q.shape == (1000, 4)
    assert np.allclose(np.linalg.norm(q, axis=1), 1.0, atol=1e-6)


def test_a_similarity_transformed_copy_has_zero_error_and_noise_gives_its_rms(tmp_path):
    rng = np.random.default_rng(0)
    s, p, q = eval_ape.read_tum(GT)
    R, t, c = _sim(rng)
    est = (p - t) @ R / c                                    # ref = c R est + t
    _write_tum(tmp_path / "est.txt", s.astype(int), est, q)
    r = eval_ape.ape(GT, str(tmp_path / "est.txt"))
    assert r["pairs"] == 1000 and r["rmse"] < 2e-6 * c + 2e-6          # 6-decimal text
    assert abs(r["scale"] - c) < 1e-4 * c and np.allclose(r["rotation"], R, atol=1e-4)
    # isotropic noise sigma on the estimate -> rmse ~ c sigma sqrt(3) (7 fitted parameters of 3000 residuals)
    sigma = 0.01
    _write_tum(tmp_path / "noisy.txt", s.astype(int), est + sigma * rng.standard_normal(est.shape), q)
    r2 = eval_ape.ape(GT, str(tmp_path / "noisy.txt"))
    assert abs(r2["rmse"] / (c * sigma * np.sqrt(3)) - 1) < 0.05
    assert r2["min"] <= r2["median"] <= r2["max"] and abs(r2["sse"] - r2["rmse"] ** 2 * r2["pairs"]) < 1e-9
    # without scale correction a scaled copy is not explained; without alignment nothing is
    r3 = eval_ape.ape(GT, str(tmp_path / "est.txt"), correct_scale=False)
    r4 = eval_ape.ape(GT, str(tmp_path / "est.txt"), align=False)
    assert r3["rmse"] > 100 * r["rmse"] and r4["rmse"] > r3["rmse"] and r3["scale"] == 1.0


def test_umeyama_equals_the_oracles_independent_restatement():
    from oracle import post_ref
    rng = np.random.default_rng(1)
    x = rng.standard_normal((200, 3)) * [3, 1, 0.3]
    R, t, c = _sim(rng)
    y = c * x @ R.T + t + 0.05 * rng.standard_normal(x.shape)
    r1, t1, c1 = eval_ape.umeyama(x, y)
    s2, R2, t2, _ = post_ref.umeyama(x, y)
    assert abs(c1 - s2) < 1e-12 and np.allclose(r1, R2, atol=1e-12) and np.allclose(t1, t2, atol=1e-12)
    ym = y * [1, 1, -1]                                        # a mirrored target still gets a proper rotation
    r3, _, _ = eval_ape.umeyama(x, ym)
    assert abs(np.linalg.det(r3) - 1) < 1e-9
    with pytest.raises(ValueError):
        eval_ape.umeyama(np.zeros((5, 3)), np.ones((5, 3)))   # degenerate


def test_association_follows_evo(tmp_path):
    """Shorter trajectory drives; closest stamp within max_diff; unmatched poses are dropped; no overlap raises."""
    ref = np.arange(0, 10, 1.0)
    est = np.array([0.004, 1.02, 2.0, 4.996, 7.5, 30.0])
    i_ref, i_est = eval_ape.associate(ref, est, 0.01)
    assert list(i_ref) == [0, 2, 5] and list(i_est) == [0, 2, 3]
    i_ref, i_est = eval_ape.associate(est, ref, 0.01)          # roles swapped: the same pairs
    assert list(i_ref) == [0, 2, 3] and list(i_est) == [0, 2, 5]
    with pytest.raises(ValueError):
        eval_ape.associate(ref, ref + 100.0)
    # a trajectory that skips frames (the creator's --skip-start) and is written with float stamps still associates
    s, p, q = eval_ape.read_tum(GT)
    _write_tum(tmp_path / "part.txt", s[7:300:3], p[7:300:3], q[7:300:3], fmt="{:.9f}")
    r = eval_ape.ape(GT, str(tmp_path / "part.txt"))
    assert r["pairs"] == len(s[7:300:3]) and r["rmse"] < 1e-5


def test_command_line_and_json(tmp_path):
    s, p, q = eval_ape.read_tum(GT)
    _write_tum(tmp_path / "est.txt", s.astype(int), p * 2.0 + 1.0, q)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "eval_ape.py"), GT, str(tmp_path / "est.txt"), "--json"],
                         capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    rec = json.loads(out.stdout)
    assert rec["pairs"] == 1000 and rec["rmse"] < 1e-5 and abs(rec["scale"] - 0.5) < 1e-6
    txt = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "eval_ape.py"), GT, str(tmp_path / "est.txt")],
                         capture_output=True, text=True, timeout=120).stdout
    assert "rmse" in txt and "Sim(3)" in txt
    bad = tmp_path / "bad.txt"
    bad.write_text("0 1 2 3\n")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "eval_ape.py"), GT, str(bad)], capture_output=True, text=True)
    assert r.returncode != 0 and "8 entries" in r.stderr


def test_eval_sequences_plans_the_reference_scripts_runs(capsys):
    """tools/eval_sequences.py --dry-run: the command lines of scripts/eval_7scenes.sh (offline: K = 400 grid keypoints,
    metric depth, intrinsics, 10 observations per track; online: integer time stamps) and scripts/eval_euroc.sh (cam0
    folder, calibration, per-sequence start frame, 7 observations, inverse depth) for this build's CLI."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("eval_sequences", os.path.join(ROOT, "tools", "eval_sequences.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.main(["7scenes", "--dataset-path", "/d", "--groundtruth-dir", "/gt", "--dry-run"])
    out = capsys.readouterr().out.strip().split("\n")
    assert len(out) == 7 * 3
    assert out[0] == ("python -m pi3_slam_amd.cli create --images /d/chess/seq-01/color --model-path recipe --output "
                      "logs/7scenes/chess --chunk-length 50 --overlap 5 --device cuda --metric-depth --keypoints grid "
                      "--max-kp 400 --estimate-intrinsics --num-workers 2")
    assert out[1].endswith("--output logs/7scenes/chess/reconstruction --max-observations-per-track 10")
    assert out[2] == "python tools/eval_ape.py /gt/chess.txt logs/7scenes/chess/reconstruction/trajectory_tum.txt"
    mod.main(["euroc", "--dataset-path", "/d", "--groundtruth-dir", "/gt", "--calib-file", "c.json", "--mode", "online",
              "--chunk-length", "100", "--overlap", "20", "--dry-run"])
    out = capsys.readouterr().out.strip().split("\n")
    assert len(out) == 5 * 2 and "--skip_start 885" in out[0] and "--use_inverse_depth" in out[0]
    assert "--max_observations_per_track 7" in out[0] and "--cam_dist_path c.json" in out[0]
    assert "--tum_integer_timestamp" not in out[0] and "--chunk_length 100 --overlap 20" in out[0]
    assert out[1] == "python tools/eval_ape.py /gt/MH_01_easy.txt logs/euroc/MH_01_easy/online/trajectory.tum"
    # the planned argv parse with the CLI's own parser
    from pi3_slam_amd import cli
    for seq, images, steps, traj, gt in mod.plan("euroc", type("A", (), dict(
            dataset_path="/d", output_dir="o", moge_model_path="m.pt", mode="offline", model_path="ckpt", chunk_length=50,
            overlap=5, calib_file="c.json", groundtruth_dir="/gt"))()):
        for st in steps:
            cli.build_parser().parse_args(st)

