"""CLI: create pi3 chunks on the MI355X path and save them to disk.  Same flags as the reference's
create_offline_chunks.py (:44-62) plus three of this build's own; under torch.distributed.run the chunks are sharded
over the GPUs of the node.

  python -m pi3_slam_amd.create_offline_chunks --images /data/seq --model-path /ckpt/pi3 --output /data/seq_chunks \\
      --chunk-length 100 --overlap 20 --keypoints grid --max-kp 200 [--cam-dist-path example/euroc_cam0_calib.json]
"""
from __future__ import annotations

import argparse
import glob
import os
from typing import List

from .chunk_creator import OfflineChunkCreator, OfflineCreatorConfig

IMAGE_PATTERNS = ("*.png", "*.jpg", "*.jpeg", "*.bmp")


def list_images(root: str) -> List[str]:
    """A folder (per extension, each group sorted, in the order png, jpg, jpeg, bmp - create_offline_chunks.py:27-40),
    a text file with one path per line, or a glob pattern."""
    if os.path.isdir(root):
        return [p for pat in IMAGE_PATTERNS for p in sorted(glob.glob(os.path.join(root, pat)))]
    if os.path.isfile(root):
        with open(root) as f:
            return [ln.strip() for ln in f if ln.strip()]
    return sorted(glob.glob(root))


def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(description="Create offline PI3 chunks and save to disk (MI355X path)")
    p.add_argument("--images", required=True, help="Folder with images, a glob pattern, or a text file listing paths")
    p.add_argument("--model-path", default="recipe",
                   help="local Pi3 checkpoint (directory or file); 'recipe' = synthetic weights (no network here)")
    p.add_argument("--output", required=True, help="Output directory")
    p.add_argument("--chunk-length", type=int, default=50)
    p.add_argument("--overlap", type=int, default=5)
    p.add_argument("--device", default="cuda")
    p.add_argument("--cam-dist-path", type=str, default=None, help="camera calibration JSON for undistortion")
    p.add_argument("--metric-depth", action="store_true", help="Enable MoGe metric scaling")
    p.add_argument("--no-metric-depth", dest="metric_depth", action="store_false")
    p.set_defaults(metric_depth=True)
    p.add_argument("--keypoints", default="grid", choices=["aliked", "grid", "none"])
    p.add_argument("--max-kp", type=int, default=200)
    p.add_argument("--kp-threshold", type=float, default=0.005)
    p.add_argument("--estimate-intrinsics", action="store_true", default=True)
    p.add_argument("--num-workers", type=int, default=4)
    p.add_argument("--skip-start", type=int, default=0, help="Number of frames to skip from the beginning")
    p.add_argument("--skip-end", type=int, default=0, help="Number of frames to skip from the end")
    # additions of this build
    p.add_argument("--moge-model-path", default=None, help="local MoGe-2 model.pt ('recipe' = synthetic weights)")
    p.add_argument("--device-resize", action="store_true", help="Resize + ToTensor on the GPU (workers decode only)")
    p.add_argument("--keypoint-seed", type=int, default=0, help="seed of the grid subsampling (-1: unseeded)")
    p.add_argument("--hip-graph", action="store_true", help="replay the per-chunk forward as a captured hipGraph")
    return p


def main(argv=None) -> None:
    args = build_parser().parse_args(argv)
    paths = list_images(args.images)
    if not paths:
        raise SystemExit(f"No images found for: {args.images}")
    total = len(paths)
    start, end = max(0, int(args.skip_start)), total - max(0, int(args.skip_end))
    if start >= total:
        raise SystemExit(f"Invalid --skip-start {args.skip_start}: exceeds total images {total}")
    if end <= start:
        raise SystemExit(f"Invalid frame range after skipping: start {start}, end {end}")
    cfg = OfflineCreatorConfig(
        model_path=args.model_path, output_dir=args.output, chunk_length=args.chunk_length, overlap=args.overlap,
        device=args.device, do_metric_depth=args.metric_depth, keypoint_type=args.keypoints,
        max_num_keypoints=args.max_kp, keypoint_detection_threshold=args.kp_threshold,
        estimate_camera_params=args.estimate_intrinsics, num_loader_workers=args.num_workers,
        cam_dist_path=args.cam_dist_path, moge_model_path=args.moge_model_path,
        keypoint_seed=None if args.keypoint_seed < 0 else args.keypoint_seed, device_resize=args.device_resize,
        hip_graph=args.hip_graph)
    OfflineChunkCreator(cfg).process_and_save(paths[start:end])


if __name__ == "__main__":
    main()
