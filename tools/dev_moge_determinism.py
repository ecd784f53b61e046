"""Stress: repeated eager MoGe infer on the same input must be bit-identical (finds racy kernels)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pi3_slam_amd.moge import MoGeEngine
eng = MoGeEngine.from_pretrained("recipe", "cuda:0")
g = torch.Generator(device="cuda:0").manual_seed(5)
imgs = [torch.rand(3, 84, 112, device="cuda:0", generator=g) for _ in range(3)]
keys = ("points_affine", "mask", "shift", "focal", "depth")
for ii, img in enumerate(imgs):
    base = {k: v.clone() for k, v in eng.infer(img, resolution_level=0).items() if k in keys}
    nbad = {k: 0 for k in keys}
    for rep in range(40):
        if rep % 3 == 0:
            junk = [torch.randn(1 << (10 + (rep % 12)), device="cuda:0") for _ in range(4)]
        out = eng.infer(img, resolution_level=0) if rep % 2 == 0 else eng.infer_graphed(img, resolution_level=0)
        for k in keys:
            p, q = torch.nan_to_num(out[k].float(), posinf=1e30), torch.nan_to_num(base[k].float(), posinf=1e30)
            if not torch.equal(p, q):
                nbad[k] += 1
                if nbad[k] <= 2:
                    print(f"img {ii} rep {rep} {'eager' if rep % 2 == 0 else 'graph'} {k}: maxdiff {(p - q).abs().max().item():.3e} n={int((p != q).sum())}")
        junk = None
    print("img", ii, "mismatching repetitions per key:", nbad)
