"""HBM-bound staging kernels at the headline shape (100 frames 308x406): time and algorithmic TB/s."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pi3_slam_amd import ops
from pi3_slam_amd.weights import IMAGE_MEAN, IMAGE_STD
dev = torch.device("cuda:0")
F, H, W, T, P = 100, 308, 406, 643, 638


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


pfeat = torch.randn(F * T, 640, device=dev) * 0.1
cfeat = torch.randn(F * T, 256, device=dev)
poses = torch.eye(4, device=dev).repeat(F, 1, 1).contiguous()
lp, pts, conf = (torch.empty(F, H, W, c, device=dev) for c in (3, 3, 1))
ms = timeit(lambda: ops.unpatchify_points(pfeat, cfeat, poses, F, H, W, T, 5, lp, pts, conf))
nbytes = F * P * (588 + 196) * 4 + F * H * W * 7 * 4
print(f"unpatchify_points   {ms * 1e3:7.1f} us   {nbytes / ms / 1e9:6.2f} TB/s of algorithmic bytes ({nbytes / 1e6:.0f} MB)")
imgs = torch.rand(F, 3, H, W, device=dev)
patches = torch.empty(F * P, 640, device=dev, dtype=torch.bfloat16)
ms = timeit(lambda: ops.patch_gather(imgs, patches, IMAGE_MEAN, IMAGE_STD))
nbytes = F * 3 * H * W * 4 + F * P * 640 * 2
print(f"patch_gather        {ms * 1e3:7.1f} us   {nbytes / ms / 1e9:6.2f} TB/s of algorithmic bytes ({nbytes / 1e6:.0f} MB)")
