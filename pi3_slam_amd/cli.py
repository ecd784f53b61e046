"""Command line of the MI355X path.  Two sub-commands that take the flags of the reference's two offline scripts
(create_offline_chunks.py:44-62, reconstruct_offline.py:21-29), declared as tables below, plus this build's own:

  python -m pi3_slam_amd.cli create --images /data/seq --output /data/seq_chunks --chunk-length 100 --overlap 20 \\
         --model-path /ckpt/pi3 [--moge-model-path model.pt] [--cam-dist-path calib.json] [--device-resize] [--hip-graph]
  python -m pi3_slam_amd.cli reconstruct --chunks /data/seq_chunks --output /data/seq_chunks/reconstruction

Under `python -m torch.distributed.run --nproc-per-node G -m pi3_slam_amd.cli ...` both stages shard over the G GPUs.
"""
from __future__ import annotations

import argparse
import glob
import os
from typing import Dict, List, Sequence, Tuple

# (flag, kwargs) - names, types and defaults follow the reference scripts; REQ marks what has a personal default path there
REQ = dict(required=True)
CREATE_FLAGS: Sequence[Tuple[str, Dict]] = (
    ("--images", dict(REQ, help="folder of images, glob pattern, or text file with one path per line")),
    ("--output", dict(REQ, help="output directory (chunks/, chunks_manifest.json, chunk_metadata.json)")),
    ("--model-path", dict(default="recipe", help="local Pi3 checkpoint; 'recipe' = synthetic weights (no network)")),
    ("--chunk-length", dict(type=int, default=50)),
    ("--overlap", dict(type=int, default=5)),
    ("--device", dict(default="cuda")),
    ("--cam-dist-path", dict(default=None, help="camera calibration JSON: undistort the frames first")),
    ("--keypoints", dict(default="grid", choices=("aliked", "grid", "none"))),
    ("--max-kp", dict(type=int, default=200)),
    ("--kp-threshold", dict(type=float, default=0.005)),
    ("--num-workers", dict(type=int, default=4)),
    ("--skip-start", dict(type=int, default=0, help="frames to drop at the beginning")),
    ("--skip-end", dict(type=int, default=0, help="frames to drop at the end")),
    # this build's additions
    ("--moge-model-path", dict(default=None, help="local MoGe-2 model.pt; 'recipe' = synthetic weights")),
    ("--keypoint-seed", dict(type=int, default=0, help="seed of the grid subsampling, -1 = unseeded")),
)
CREATE_SWITCHES = (("--device-resize", "Resize + ToTensor on the GPU (loader workers decode only)"),
                   ("--hip-graph", "replay the per-chunk forward as one captured hipGraph"),
                   ("--reuse-overlap-encoder", "overlap frames take their encoder output from the previous chunk "
                                               "(bit-identical; single-GPU streams)"))
RECON_FLAGS: Sequence[Tuple[str, Dict]] = (
    ("--chunks", dict(REQ, help="directory holding chunks/chunk_*.pt and chunk_metadata.json")),
    ("--output", dict(REQ, help="directory for trajectory_tum.txt and the ply files")),
    ("--chunk-length", dict(type=int, default=None)),
    ("--overlap", dict(type=int, default=None)),
    ("--max-observations-per-track", dict(type=int, default=5)),
    ("--device", dict(default="cuda")),
)
RECON_SWITCHES = (("--save-per-chunk", "per-chunk ply files as well"),
                  ("--use-inverse-depth", "one inverse depth per track along its reference keypoint's ray in both bundle "
                                         "adjustments (utils/chunk_reconstruction.py:187-204; pi3_bundle_adjust_inverse_depth)"),
                  ("--save-observations", "also write the projected track observations"),
                  ("--no-bundle-adjust", "closed-form Sim(3) chain only: skip the per-chunk and the prior-constrained "
                                         "bundle adjustment (utils/chunk_reconstruction.py:188-219, "
                                         "utils/reconstruction_alignment.py:107-171)"),
                  ("--no-ba-sanity-gate", "apply every bundle adjustment that ends without a numerical failure, as the "
                                          "reference does (default: keep the input when fewer than 3 tracks survive or a "
                                          "camera moved by more than the scene extent)"))


def list_images(root: str) -> List[str]:
    """Folder -> its png, jpg, jpeg, bmp files (each extension group sorted, groups in that order); file -> its
    non-empty lines; anything else is a glob pattern."""
    if os.path.isdir(root):
        return [f for ext in ("png", "jpg", "jpeg", "bmp") for f in sorted(glob.glob(os.path.join(root, "*." + ext)))]
    if os.path.isfile(root):
        return [ln for ln in (raw.strip() for raw in open(root)) if ln]
    return sorted(glob.glob(root))


def build_parser() -> argparse.ArgumentParser:
    top = argparse.ArgumentParser(prog="pi3_slam_amd.cli", description=__doc__.split("\n")[0])
    sub = top.add_subparsers(dest="command", required=True)
    for name, flags, switches in (("create", CREATE_FLAGS, CREATE_SWITCHES), ("reconstruct", RECON_FLAGS, RECON_SWITCHES)):
        p = sub.add_parser(name)
        for flag, kw in flags:
            p.add_argument(flag, **kw)
        for flag, text in switches:
            p.add_argument(flag, action="store_true", help=text)
        if name == "create":   # the two tri-state switches of the reference script
            p.add_argument("--metric-depth", dest="metric_depth", action="store_true", default=True)
            p.add_argument("--no-metric-depth", dest="metric_depth", action="store_false")
            p.add_argument("--estimate-intrinsics", action="store_true", default=True)
    return top


def run_create(a: argparse.Namespace) -> None:
    from .chunk_creator import OfflineChunkCreator, OfflineCreatorConfig
    paths = list_images(a.images)
    if not paths:
        raise SystemExit(f"No images found for: {a.images}")
    lo, hi = max(0, a.skip_start), len(paths) - max(0, a.skip_end)
    if lo >= len(paths) or hi <= lo:
        raise SystemExit(f"Invalid frame range after --skip-start {a.skip_start} / --skip-end {a.skip_end}: "
                         f"{len(paths)} images")
    cfg = OfflineCreatorConfig(
        model_path=a.model_path, output_dir=a.output, chunk_length=a.chunk_length, overlap=a.overlap, device=a.device,
        do_metric_depth=a.metric_depth, keypoint_type=a.keypoints, max_num_keypoints=a.max_kp,
        keypoint_detection_threshold=a.kp_threshold, estimate_camera_params=a.estimate_intrinsics,
        num_loader_workers=a.num_workers, cam_dist_path=a.cam_dist_path, moge_model_path=a.moge_model_path,
        keypoint_seed=None if a.keypoint_seed < 0 else a.keypoint_seed, device_resize=a.device_resize,
        hip_graph=a.hip_graph, reuse_overlap_encoder=a.reuse_overlap_encoder)
    OfflineChunkCreator(cfg).process_and_save(paths[lo:hi])


def run_reconstruct(a: argparse.Namespace) -> None:
    from .reconstructor import OfflineReconstructor
    os.makedirs(a.output, exist_ok=True)
    OfflineReconstructor(chunk_dir=a.chunks, output_dir=a.output, chunk_length=a.chunk_length, overlap=a.overlap,
                         max_observations_per_track=a.max_observations_per_track, save_per_chunk=a.save_per_chunk,
                         use_inverse_depth=a.use_inverse_depth, device=a.device,
                         save_observations=a.save_observations, bundle_adjust=not a.no_bundle_adjust,
                         ba_sanity_gate=not a.no_ba_sanity_gate).run()


def main(argv=None) -> None:
    a = build_parser().parse_args(argv)
    (run_create if a.command == "create" else run_reconstruct)(a)


if __name__ == "__main__":
    main()
