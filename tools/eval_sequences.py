"""The reference's two evaluation scripts (scripts/eval_7scenes.sh, scripts/eval_euroc.sh) for this build: per
sequence create -> reconstruct (or the online stream), then the APE of the trajectory against the ground-truth TUM
file (`evo_ape tum GT EST -as`, restated in tools/eval_ape.py), with the flags, per-dataset settings and output layout
of those scripts.

    python tools/eval_sequences.py 7scenes --dataset-path /data/7scenes/ --groundtruth-dir GT/7scenes \\
           --model-path /ckpt/pi3 [--moge-model-path model.pt] [--mode offline|online] [--chunk-length 50] [--overlap 5]
    python tools/eval_sequences.py euroc --dataset-path /data/euroc/ --groundtruth-dir GT/euroc --calib-file calib.json ...

The ground-truth files are the reference's scripts/groundtruths/{7scenes,euroc}/*.txt (data; two of them are committed as
test fixtures under tests/golden/).  Needs the released checkpoints and the datasets: nothing of this runs offline,
which is why bench.py's `second_metric` only reports a value when PI3_WEIGHTS / PI3_SEVEN_SCENES are set.
"""
from __future__ import annotations

import argparse
import json
import os
import shutil
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

# eval_7scenes.sh:39-47 / eval_euroc.sh:42-61: sequences, image folders, per-sequence start frames, per-dataset settings
SUITES = {
    "7scenes": dict(
        sequences=["chess", "fire", "heads", "office", "pumpkin", "redkitchen", "stairs"],
        images=lambda root, s: os.path.join(root, s, "seq-01", "color"),
        start=lambda s: 0, max_obs=10, inverse_depth=False, integer_stamps=True, calib=False),
    "euroc": dict(
        sequences=["MH_01_easy", "MH_02_easy", "MH_03_medium", "MH_04_difficult", "MH_05_difficult"],
        images=lambda root, s: os.path.join(root, s, "mav0", "cam0", "data"),
        start=lambda s: {"MH_01_easy": 885, "MH_02_easy": 922, "MH_03_medium": 388, "MH_04_difficult": 415,
                         "MH_05_difficult": 425}.get(s, 0),
        max_obs=7, inverse_depth=True, integer_stamps=False, calib=True),
}


def plan(suite: str, a: argparse.Namespace):
    """[(sequence, argv of the create / reconstruct or the online run, trajectory file, ground-truth file)]."""
    cfg = SUITES[suite]
    runs = []
    for seq in cfg["sequences"]:
        images = cfg["images"](a.dataset_path, seq)
        out = os.path.join(a.output_dir, seq)
        common_moge = ["--moge-model-path", a.moge_model_path] if a.moge_model_path else []
        if a.mode == "offline":
            create = ["create", "--images", images, "--model-path", a.model_path, "--output", out, "--chunk-length",
                      str(a.chunk_length), "--overlap", str(a.overlap), "--device", "cuda", "--metric-depth", "--keypoints",
                      "grid", "--max-kp", "400", "--estimate-intrinsics", "--num-workers", "2"] + common_moge
            if cfg["calib"]:
                create += ["--cam-dist-path", a.calib_file, "--skip-start", str(cfg["start"](seq))]
            recon = ["reconstruct", "--chunks", out, "--output", os.path.join(out, "reconstruction"),
                     "--max-observations-per-track", str(cfg["max_obs"])] + (["--use-inverse-depth"] if cfg["inverse_depth"] else [])
            steps, traj = [create, recon], os.path.join(out, "reconstruction", "trajectory_tum.txt")
        else:
            online = ["online", "--image_dir", images, "--model_path", a.model_path, "--device", "cuda", "--chunk_length",
                      str(a.chunk_length), "--overlap", str(a.overlap), "--keypoint_type", "grid", "--max_num_keypoints", "400",
                      "--max_observations_per_track", str(cfg["max_obs"]), "--do_metric_depth", "--output_path",
                      os.path.join(out, "online"), "--save_tum", "--no_visualization"]
            online += ["--moge_model_path", a.moge_model_path] if a.moge_model_path else []
            if cfg["integer_stamps"]:
                online.append("--tum_integer_timestamp")
            if cfg["inverse_depth"]:
                online.append("--use_inverse_depth")
            if cfg["calib"]:
                online += ["--cam_dist_path", a.calib_file, "--skip_start", str(cfg["start"](seq))]
            steps, traj = [online], os.path.join(out, "online", "trajectory.tum")
        runs.append((seq, images, steps, traj, os.path.join(a.groundtruth_dir, seq + ".txt")))
    return runs


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("suite", choices=sorted(SUITES))
    ap.add_argument("--dataset-path", required=True)
    ap.add_argument("--groundtruth-dir", required=True)
    ap.add_argument("--output-dir", default=None, help="default logs/<suite> as in the reference scripts")
    ap.add_argument("--overlap", type=int, default=5)
    ap.add_argument("--chunk-length", type=int, default=50)
    ap.add_argument("--mode", choices=("offline", "online"), default="offline")
    ap.add_argument("--calib-file", default=None, help="camera calibration JSON (euroc: example/euroc_cam0_calib.json)")
    ap.add_argument("--model-path", default="recipe")
    ap.add_argument("--moge-model-path", default=None)
    ap.add_argument("--dry-run", action="store_true", help="print the command lines, run nothing")
    a = ap.parse_args(argv)
    a.output_dir = a.output_dir or os.path.join("logs", a.suite)
    if SUITES[a.suite]["calib"] and not a.calib_file:
        ap.error("--calib-file is required for this suite")
    results = {}
    for seq, images, steps, traj, gt in plan(a.suite, a):
        if a.dry_run:
            for st in steps:
                print("python -m pi3_slam_amd.cli " + " ".join(st))
            print(f"python tools/eval_ape.py {gt} {traj}")
            continue
        if not os.path.isdir(images):
            print(f"⚠️  Dataset directory not found: {images}")
            continue
        from pi3_slam_amd import cli
        for st in steps:
            if st[0] == "reconstruct":
                shutil.rmtree(st[st.index("--output") + 1], ignore_errors=True)      # "Removing old reconstruction folder"
            cli.main(st)
        if not os.path.isfile(gt):
            print(f"⚠️  Groundtruth file not found: {gt}")
            continue
        import eval_ape
        stats = eval_ape.ape(gt, traj)
        stats = {k: (float(v) if k != "pairs" else int(v)) for k, v in stats.items()
                 if k in ("rmse", "mean", "median", "std", "min", "max", "pairs", "scale")}
        results[seq] = stats
        print(f"📈 {seq}: APE rmse {stats['rmse']:.4f} m over {stats['pairs']} poses (scale {stats['scale']:.4f})")
    if results:
        mean = sum(r["rmse"] for r in results.values()) / len(results)
        print(json.dumps({"suite": a.suite, "mode": a.mode, "mean_rmse_m": mean, "sequences": results}))
    return 0


if __name__ == "__main__":
    sys.exit(main())
