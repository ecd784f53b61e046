"""CLI: align saved chunks into one trajectory / point cloud (flags of the reference's reconstruct_offline.py:21-29).

  python -m pi3_slam_amd.reconstruct_offline --chunks /data/seq_chunks --output /data/seq_chunks/reconstruction
"""
from __future__ import annotations

import argparse
import os

from .reconstructor import OfflineReconstructor


def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(description="Reconstruct from saved PI3 chunks (MI355X path)")
    p.add_argument("--chunks", required=True, help="Directory that holds chunks/chunk_*.pt and chunk_metadata.json")
    p.add_argument("--output", required=True, help="Directory to write the trajectory / ply files")
    p.add_argument("--chunk-length", type=int, default=None)
    p.add_argument("--overlap", type=int, default=None)
    p.add_argument("--max-observations-per-track", type=int, default=5)
    p.add_argument("--save-per-chunk", action="store_true", help="Save per-chunk .ply files as well")
    p.add_argument("--use-inverse-depth", action="store_true", help="accepted for compatibility (bundle adjustment is "
                   "not part of this build)")
    p.add_argument("--device", default="cuda:0")
    p.add_argument("--save-observations", action="store_true", help="also write the projected track observations")
    return p


def main(argv=None) -> None:
    args = build_parser().parse_args(argv)
    os.makedirs(args.output, exist_ok=True)
    OfflineReconstructor(chunk_dir=args.chunks, output_dir=args.output, chunk_length=args.chunk_length,
                         overlap=args.overlap, max_observations_per_track=args.max_observations_per_track,
                         save_per_chunk=args.save_per_chunk, use_inverse_depth=args.use_inverse_depth,
                         device=args.device, save_observations=args.save_observations).run()


if __name__ == "__main__":
    main()
