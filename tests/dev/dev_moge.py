"""Developer check of the MoGe engine on a GPU box against the golden vectors + stage-by-stage vs the CPU oracle."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import moge_ref
from oracle.gen_golden_moge import CASES, moge_image
from pi3_slam_amd.moge import MoGeEngine, SYNTHETIC_CONFIG, recipe_state_dict_cpu

eng = MoGeEngine.from_pretrained("recipe", "cuda:0")
sd = recipe_state_dict_cpu(SYNTHETIC_CONFIG)
for name, (H, W, level) in CASES.items():
    g = np.load(os.path.join("tests/golden", name + ".npz"))
    img = moge_image(name, H, W)
    t0 = time.time(); out = eng.infer(img, resolution_level=level); torch.cuda.synchronize(); dt = time.time() - t0
    z = out["points_affine"][..., 2].cpu().numpy()
    d = np.abs(z - g["points_affine_z"])
    mask_ref = np.unpackbits(g["mask"])[: H * W].reshape(H, W).astype(bool)
    mask = out["mask"].cpu().numpy()
    print(name, f"{dt*1e3:.1f} ms  z err mean {d.mean():.3e} max {d.max():.3e} (anchor {g['bf16err_z']})  mask flips {(mask != mask_ref).mean():.4f}")
    both = mask & mask_ref
    depth = out["depth"].cpu().numpy()
    rel = np.abs(depth[both] - g["depth"][both]) / g["depth"][both]
    print("   depth rel median", np.median(rel), "max", rel.max(), " focal", out["focal"].item(), "shift", out["shift"].item(), "ref K", g["intrinsics"][0, 0], out["intrinsics"][0, 0].item())
    t0 = time.time(); out = eng.infer(img, resolution_level=level); torch.cuda.synchronize(); print("   second call ms", (time.time() - t0) * 1e3)
