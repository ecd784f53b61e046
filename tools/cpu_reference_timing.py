"""CPU baseline of BASELINE.md §3.2: the REAL reference (imported from /root/reference, build container only) timed
on this box's host cores, with ONE change: the `sdpa_kernel([MATH, EFFICIENT])` context of
pi3/models/layers/attention.py:339-341 (and :105-107) is not entered, so torch picks its fused CPU SDPA.  As written
the reference cannot run N=100 on a CPU: the MATH backend materialises 16 x 64 300^2 x 4 B = 265 GB of scores.

    python tools/cpu_reference_timing.py --frames 32          # BASELINE configs[0] frame count
    python tools/cpu_reference_timing.py --frames 100         # the headline chunk, ~20-40 min on 8 cores

What is timed (one pass, no warm-up: a pass is minutes long): `OfflineChunkCreator._process_single_chunk` of the
reference (slam/offline_chunk_creator.py:161-256) = Pi3 forward (fp32, eager; the reference's CPU autocast is
fp32 = off) + masks + intrinsics LM (scipy) + grid keypoints + gather + fp16 pack, with grid K=200 keypoints and
without MoGe (quirk 6 of SURVEY §8: MoGe is hard-wired to 'cuda' and silently disabled on a CPU box).  The object is
built with object.__new__ (its __init__ fetches checkpoints by name); weights are the module's own random init
(values do not change the time).  utils3d is absent here, so the reference's intrinsics estimation raises inside its own
try/except and is skipped exactly as it would be for a user without utils3d.
Third-party modules absent here get empty placeholders (as oracle/gen_golden_post.py does).
Prints one JSON line; record it in BASELINE.md §2 and cite it in bench.py's cpu_baseline.sample."""
from __future__ import annotations

import argparse
import contextlib
import json
import os
import sys
import time
import types

import torch

REF = "/root/reference"


class _Placeholder(types.ModuleType):
    __path__ = []

    def __getattr__(self, item):
        if item.startswith("__"):
            raise AttributeError(item)
        return _Placeholder(f"{self.__name__}.{item}")


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=32)
    ap.add_argument("--height", type=int, default=308)
    ap.add_argument("--width", type=int, default=406)
    ap.add_argument("--threads", type=int, default=os.cpu_count() or 8)
    ap.add_argument("--keypoints", type=int, default=200)
    args = ap.parse_args()
    torch.set_num_threads(args.threads)
    os.environ.setdefault("TORCHDYNAMO_DISABLE", "1")

    for name in ("cv2", "natsort", "plyfile", "torchvision", "torchvision.transforms", "torchcodec",
                 "torchcodec.decoders", "pytheia"):
        if name not in sys.modules:
            sys.modules[name] = _Placeholder(name)
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    sys.path.insert(0, REF)
    os.chdir(REF)
    from pi3.models.pi3 import Pi3
    from slam.offline_chunk_creator import OfflineChunkCreator, OfflineCreatorConfig
    from utils.keypoint_extraction import create_keypoint_extractor

    # the one change: do not pin the SDPA backend (attention.py reaches it as nn.attention.sdpa_kernel)
    torch.nn.attention.sdpa_kernel = lambda *a, **k: contextlib.nullcontext()

    t0 = time.time()
    model = Pi3().eval()
    build_s = time.time() - t0
    cr = object.__new__(OfflineChunkCreator)
    cr.config = OfflineCreatorConfig(model_path="unused", output_dir="/tmp/pi3_cpu_ref", chunk_length=args.frames,
                                     overlap=20, device="cpu", do_metric_depth=False, keypoint_type="grid",
                                     max_num_keypoints=args.keypoints, estimate_camera_params=True)
    cr.model = model
    cr.moge_model = None
    cr.keypoint_extractor = create_keypoint_extractor(keypoint_type="grid", max_num_keypoints=args.keypoints,
                                                      detection_threshold=0.005, device="cpu")
    cr.target_size = (args.height, args.width)
    cr.undistortion_maps = None
    g = torch.Generator().manual_seed(0)
    imgs = torch.rand(1, args.frames, 3, args.height, args.width, generator=g)
    paths = [[f"f{i}.png"] for i in range(args.frames)]
    print(f"reference Pi3 built in {build_s:.1f} s; timing _process_single_chunk on {args.frames} frames "
          f"{args.height}x{args.width}, {args.threads} threads ...", flush=True)
    t0 = time.time()
    res = cr._process_single_chunk(imgs, paths)
    wall = time.time() - t0
    line = {"what": "reference _process_single_chunk, sdpa_kernel context not entered", "frames": args.frames,
            "size": [args.height, args.width], "threads": args.threads, "nproc": os.cpu_count(),
            "torch": torch.__version__, "wall_s": round(wall, 2), "forward_s": round(res["_metrics"]["infer_s"], 2),
            "frames_per_s": round(args.frames / wall, 5), "keypoints": int(res["keypoints"].shape[1]),
            "has_intrinsics": "intrinsics" in res and res["intrinsics"] is not None,
            "peak_rss_gb": round(__import__("resource").getrusage(__import__("resource").RUSAGE_SELF).ru_maxrss / 2 ** 20, 2)}
    print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
