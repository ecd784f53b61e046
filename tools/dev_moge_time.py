"""MoGe forward alone on the GPU at the chunk's frame size (308 x 406): eager and hipGraph replay times; run under
rocprofv3 --kernel-trace --stats for the per-kernel breakdown (tools/gpu_moge_profile.sh)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pi3_slam_amd.moge import MoGeEngine  # noqa: E402

dev = torch.device("cuda:0")
eng = MoGeEngine.from_pretrained("recipe", str(dev))
H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (308, 406)
img = torch.rand(3, H, W, device=dev)


def timed(fn, n=10):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


print(f"eager  {timed(lambda: eng.infer(img)['depth']):.3f} ms per frame {H}x{W}")
if hasattr(eng, "infer_graphed"):
    print(f"graph  {timed(lambda: eng.infer_graphed(img)['depth']):.3f} ms per frame")
