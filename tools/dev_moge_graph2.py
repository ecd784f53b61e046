import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pi3_slam_amd.moge import MoGeEngine
engine = MoGeEngine.from_pretrained("recipe", "cuda:0")
g = torch.Generator(device="cuda:0").manual_seed(5)
a, b = (torch.rand(3, 84, 112, device="cuda:0", generator=g) for _ in range(2))
c = torch.rand(3, 70, 98, device="cuda:0", generator=g)
keys = ("depth", "mask", "intrinsics", "points_affine")
for it, img in enumerate((a, b, c, a)):
    ref = {k: engine.infer(img, resolution_level=0)[k].clone() for k in keys}
    full = {k: v.clone() for k, v in engine.infer(img, resolution_level=0).items() if torch.is_tensor(v)}
    out = engine.infer_graphed(img, resolution_level=0)
    torch.cuda.synchronize()
    for k in keys:
        x, y, z = out[k].float(), ref[k].float(), full[k].float()
        x, y, z = (torch.nan_to_num(t, posinf=1e30) for t in (x, y, z))
        print(it, k, "graph==ref", torch.equal(x, y), "graph==full", torch.equal(x, z), "ref==full", torch.equal(y, z),
              (x - y).abs().max().item())
    print(it, "shift graph", out["shift"].item(), "full", full["shift"].item())
