"""Debug: find reads of uninitialised device memory.  torch.empty is patched to poison every new tensor (NaN for
floating point, 0x7f bytes otherwise); outputs must equal the un-poisoned run bit for bit."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pi3_slam_amd.moge import MoGeEngine
from pi3_slam_amd.engine import Pi3Engine
from pi3_slam_amd.weights import Pi3Config

_empty = torch.empty
mode = {"v": None}
def poisoned(*a, **k):
    t = _empty(*a, **k)
    if mode["v"] is not None and t.is_cuda:
        if t.dtype.is_floating_point:
            t.fill_(mode["v"])
        else:
            t.view(torch.uint8).fill_(0x7f)
    return t
torch.empty = poisoned

def run(fn, keys):
    outs = []
    for v in (None, float("nan"), 1e30, 0.0):
        mode["v"] = v
        o = fn()
        outs.append({k: o[k].clone() for k in keys})
    mode["v"] = None
    for i, tag in ((1, "nan"), (2, "1e30"), (3, "zero")):
        for k in keys:
            p, q = torch.nan_to_num(outs[0][k].float(), nan=-7.0, posinf=1e30), torch.nan_to_num(outs[i][k].float(), nan=-7.0, posinf=1e30)
            if not torch.equal(p, q):
                print(f"   MISMATCH poison={tag} {k}: n={int((p != q).sum())} maxdiff={(p - q).abs().max().item():.3e}")
    print("   done")

g = torch.Generator(device="cuda:0").manual_seed(5)
eng = MoGeEngine.from_pretrained("recipe", "cuda:0")
img = torch.rand(3, 84, 112, device="cuda:0", generator=g)
print("moge 84x112 level 0"); run(lambda: eng.infer(img, resolution_level=0), ("points_affine", "mask", "shift", "focal", "depth"))
img2 = torch.rand(3, 308, 406, device="cuda:0", generator=g)
print("moge 308x406 level 9"); run(lambda: eng.infer(img2, resolution_level=9), ("points_affine", "mask", "shift", "focal", "depth"))
small = Pi3Engine(Pi3Config(dim=128, enc_depth=2, dec_depth=4, head_depth=1, cam_dim=128, pos_grid=5), "cuda:0")
x = torch.rand(1, 4, 3, 56, 70, device="cuda:0", generator=g)
def f():
    small._buf.clear()
    return small.forward(x)
print("pi3 small"); run(f, ("points", "local_points", "conf", "camera_poses"))
full = Pi3Engine(Pi3Config(), "cuda:0")
x2 = torch.rand(1, 3, 3, 56, 70, device="cuda:0", generator=g)
def f2():
    full._buf.clear()
    return full.forward(x2)
print("pi3 full model, tiny input"); run(f2, ("points", "local_points", "conf", "camera_poses"))
