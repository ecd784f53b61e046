"""Debug: eager vs eager vs graphed MoGe infer, per output key (bit equality / max diff)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pi3_slam_amd.moge import MoGeEngine
eng = MoGeEngine.from_pretrained("recipe", "cuda:0")
g = torch.Generator(device="cuda:0").manual_seed(5)
a, b = (torch.rand(3, 84, 112, device="cuda:0", generator=g) for _ in range(2))
c = torch.rand(3, 70, 98, device="cuda:0", generator=g)
keys = ("points_affine", "mask", "shift", "focal", "depth")
def cmp(x, y, tag):
    for k in keys:
        p, q = x[k].float(), y[k].float()
        p, q = torch.nan_to_num(p, posinf=1e30), torch.nan_to_num(q, posinf=1e30)
        print(f"  {tag} {k}: equal={torch.equal(p, q)} maxdiff={(p - q).abs().max().item():.3e}")
for name, img in (("a", a), ("b", b), ("c", c), ("a2", a)):
    e1 = {k: v.clone() for k, v in eng.infer(img, resolution_level=0).items() if k in keys}
    junk = torch.full((1 << 22,), float("nan"), device="cuda:0"); del junk
    e2 = {k: v.clone() for k, v in eng.infer(img, resolution_level=0).items() if k in keys}
    gr = {k: v.clone() for k, v in eng.infer_graphed(img, resolution_level=0).items() if k in keys}
    gr2 = {k: v.clone() for k, v in eng.infer_graphed(img, resolution_level=0).items() if k in keys}
    torch.cuda.synchronize()
    print(name); cmp(e1, e2, "eager/eager"); cmp(e1, gr, "eager/graph"); cmp(gr, gr2, "graph/graph")
