"""Command line of the MI355X path.  Sub-commands that take the flags of the reference's two offline scripts
(create_offline_chunks.py:44-62, reconstruct_offline.py:21-29) and of its online script
(pi3_slam_online_modular.py:117-183, without the viser window), declared as tables below, plus this build's own:

  python -m pi3_slam_amd.cli create --images /data/seq --output /data/seq_chunks --chunk-length 100 --overlap 20 \\
         --model-path /ckpt/pi3 [--moge-model-path model.pt] [--cam-dist-path calib.json] [--device-resize] [--hip-graph]
  python -m pi3_slam_amd.cli reconstruct --chunks /data/seq_chunks --output /data/seq_chunks/reconstruction

  python -m pi3_slam_amd.cli online --image_dir /data/seq --output_path /data/seq_result --chunk_length 100 --overlap 20 \
         --model_path /ckpt/pi3 --save_tum [--cam_dist_path calib.json] [--use_inverse_depth]

Under `python -m torch.distributed.run --nproc-per-node G -m pi3_slam_amd.cli ...` every stage shards over the G GPUs.
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


# pi3_slam_online_modular.py:117-183: same names (underscores), types and defaults; the reference's personal default paths
# become required arguments, the three viser flags are accepted and ignored (no window in this build)
ONLINE_FLAGS: Sequence[Tuple[str, Dict]] = (
    ("--image_dir", dict(default=None, help="directory containing images")),
    ("--video_path", dict(default=None, help="video file (not supported here: no decoder in this image)")),
    ("--start_frame", dict(type=int, default=0)),
    ("--end_frame", dict(type=int, default=None)),
    ("--skip_start", dict(type=int, default=0, help="frames to drop at the beginning")),
    ("--skip_end", dict(type=int, default=0, help="frames to drop at the end")),
    ("--model_path", dict(default="recipe", help="local Pi3 checkpoint; 'recipe' = synthetic weights (no network)")),
    ("--device", dict(default="cuda")),
    ("--chunk_length", dict(type=int, default=30)),
    ("--overlap", dict(type=int, default=5)),
    ("--conf_threshold", dict(type=float, default=0.5)),
    ("--cam_scale", dict(type=float, default=1.0)),
    ("--cam_dist_path", dict(default=None, help="camera calibration JSON: undistort the frames first")),
    ("--keypoint_type", dict(default="grid")),
    ("--max_num_keypoints", dict(type=int, default=200)),
    ("--keypoint_detection_threshold", dict(type=float, default=0.005)),
    ("--max_observations_per_track", dict(type=int, default=6)),
    ("--viz_port", dict(type=int, default=8080, help="ignored: no visualisation window in this build")),
    ("--output_path", dict(REQ, help="directory (or .ply file) for trajectory.ply / trajectory.tum")),
    ("--max_points", dict(type=int, default=1000000)),
    # this build's additions
    ("--moge_model_path", dict(default=None, help="local MoGe-2 model.pt; 'recipe' = synthetic weights")),
    ("--num_workers", dict(type=int, default=4, help="decode threads")),
)
ONLINE_SWITCHES = (("--save_chunk_reconstructions", "save each chunk reconstruction to disk"),
                   ("--save_transformed_reconstructions", "save transformed reconstructions as PLY files"),
                   ("--save_debug_reconstructions", "accepted for compatibility"),
                   ("--save_debug_projections", "accepted for compatibility"),
                   ("--use_inverse_depth", "inverse-depth parametrization in both bundle adjustments"),
                   ("--no_visualization", "accepted for compatibility (there is never a window)"),
                   ("--keep_viz_open", "accepted for compatibility"),
                   ("--save_tum", "save the trajectory in TUM format"),
                   ("--tum_integer_timestamp", "integer timestamps in the TUM file (7-Scenes)"),
                   ("--no_bundle_adjust", "closed-form Sim(3) chain only"),
                   ("--no_hip_graph", "plain kernel launches instead of the captured per-chunk graph"),
                   ("--reuse_overlap_encoder", "overlap frames take their encoder output from the previous chunk"))


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
    for name, flags, switches in (("create", CREATE_FLAGS, CREATE_SWITCHES), ("reconstruct", RECON_FLAGS, RECON_SWITCHES),
                                  ("online", ONLINE_FLAGS, ONLINE_SWITCHES)):
        p = sub.add_parser(name)
        for flag, kw in flags:
            p.add_argument(flag, **kw)
        for flag, text in switches:
            p.add_argument(flag, action="store_true", help=text)
        if name == "create":   # the two tri-state switches of the reference script
            p.add_argument("--metric-depth", dest="metric_depth", action="store_true", default=True)
            p.add_argument("--no-metric-depth", dest="metric_depth", action="store_false")
            p.add_argument("--estimate-intrinsics", action="store_true", default=True)
        if name == "online":   # store_true with default True in the reference: always on there, switchable here
            p.add_argument("--estimate_camera_params", action="store_true", default=True)
            p.add_argument("--do_metric_depth", dest="do_metric_depth", action="store_true", default=True)
            p.add_argument("--no_metric_depth", dest="do_metric_depth", action="store_false")
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


def online_image_paths(a: argparse.Namespace) -> List[str]:
    """load_image_paths of the reference's online script (pi3_slam_online_modular.py:66-113) for image directories:
    skip_start / skip_end trim the sorted list; a video needs a decoder this image does not have."""
    if bool(a.image_dir) == bool(a.video_path):
        raise SystemExit("Must specify either --image_dir or --video_path" if not a.image_dir
                         else "Cannot specify both --image_dir and --video_path")
    if a.video_path:
        raise SystemExit("--video_path: no video decoder (torchcodec / cv2) in this build; extract the frames and use --image_dir")
    if a.skip_start < 0 or a.skip_end < 0:
        raise SystemExit("--skip_start / --skip_end must be non-negative")
    paths = list_images(a.image_dir)
    lo, hi = a.skip_start, len(paths) - a.skip_end
    if lo >= hi:
        raise SystemExit(f"No images found after applying frame skipping ({len(paths)} images)")
    return paths[lo:hi]


def run_online(a: argparse.Namespace) -> None:
    """main() of pi3_slam_online_modular.py:186-372 without the viser window: stream the images through Pi3SLAMOnline,
    then trajectory.ply (+ trajectory.tum with --save_tum) under --output_path."""
    from .online import Pi3SLAMOnline
    from .undistortion import create_undistortion_maps
    paths = online_image_paths(a)
    print(f"✅ Total images/frames to process: {len(paths):,}")
    maps = create_undistortion_maps(a.cam_dist_path, device=a.device) if a.cam_dist_path else None
    out_is_dir = os.path.isdir(a.output_path) or a.output_path.endswith("/") or not a.output_path.endswith(".ply")
    out_dir = a.output_path if out_is_dir else os.path.dirname(a.output_path)
    os.makedirs(out_dir or ".", exist_ok=True)
    slam = Pi3SLAMOnline(
        chunk_length=a.chunk_length, overlap=a.overlap, device=a.device, conf_threshold=a.conf_threshold,
        undistortion_maps=maps, cam_scale=a.cam_scale, estimate_camera_params=a.estimate_camera_params,
        keypoint_type=a.keypoint_type, max_num_keypoints=a.max_num_keypoints,
        keypoint_detection_threshold=a.keypoint_detection_threshold,
        save_chunk_reconstructions=a.save_chunk_reconstructions, max_observations_per_track=a.max_observations_per_track,
        do_metric_depth=a.do_metric_depth, model_path=a.model_path,
        use_inverse_depth=a.use_inverse_depth, moge_model_path=a.moge_model_path, hip_graph=not a.no_hip_graph,
        output_dir=out_dir, num_loader_workers=a.num_workers, bundle_adjust=not a.no_bundle_adjust,
        reuse_overlap_encoder=a.reuse_overlap_encoder)
    slam.save_transformed_reconstructions = a.save_transformed_reconstructions
    slam.save_debug_reconstructions = a.save_debug_reconstructions
    slam.process_chunks(paths)
    stats = slam.get_statistics()
    print(f"✅ Total chunks processed: {stats['num_chunks']}\n✅ Total frames processed: {stats['num_frames']}")
    if slam.rank != 0:
        return
    ply = os.path.join(out_dir, "trajectory.ply") if out_is_dir else a.output_path
    tum = os.path.join(out_dir, "trajectory.tum") if out_is_dir else a.output_path.replace(".ply", ".tum")
    slam.save_final_result(ply, max_points=a.max_points)
    if a.save_tum:
        slam.save_trajectory_tum(tum, integer_timestamp=a.tum_integer_timestamp)
    print(f"💾 Saved {ply}" + (f" and {tum}" if a.save_tum else ""))


def main(argv=None) -> None:
    a = build_parser().parse_args(argv)
    {"create": run_create, "reconstruct": run_reconstruct, "online": run_online}[a.command](a)


if __name__ == "__main__":
    main()
