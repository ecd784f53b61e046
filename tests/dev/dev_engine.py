"""Developer check of the whole pi3 forward on a GPU box: small config vs the CPU oracle, full config vs golden,
and a timed north-star forward.  Usage: python tools/dev_engine.py [--full] [--time]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from oracle import pi3_ref
from oracle.gen_golden import CASES, golden_images
from pi3_slam_amd.engine import Pi3Engine
from pi3_slam_amd.weights import Pi3Config, recipe_state_dict_cpu


def report(name, a, b):
    a, b = a.float().cpu(), b.float().cpu()
    d = (a - b).abs()
    print(f"  {name:16s} max|d| {d.max().item():.3e}  mean|d| {d.mean().item():.3e}  ref mean|x| {b.abs().mean().item():.3e}"
          f"  rel-mean {d.mean().item() / (b.abs().mean().item() + 1e-12):.3e}")


def rot_err_deg(Ra, Rb):
    R = Ra[..., :3, :3].double() @ Rb[..., :3, :3].double().transpose(-1, -2)
    tr = (R.diagonal(dim1=-2, dim2=-1).sum(-1) - 1) / 2
    return torch.rad2deg(torch.acos(tr.clamp(-1, 1))).max().item()


def small():
    cfg = Pi3Config(dim=128, enc_depth=2, dec_depth=4, head_depth=1, cam_dim=128, pos_grid=5)
    sd = recipe_state_dict_cpu(cfg)
    eng = Pi3Engine(cfg, "cuda:0")
    for (B, N, H, W) in [(1, 3, 28, 42), (2, 2, 70, 70), (1, 4, 56, 84)]:
        imgs = golden_images("dev", B, N, H, W)
        ref = pi3_ref.pi3_forward(sd, imgs, cfg, return_intermediates=True)
        out = eng.forward(imgs, return_intermediates=True)
        torch.cuda.synchronize()
        print("small config", (B, N, H, W))
        for k in ref["_intermediates"]:
            report("i_" + k, out["_intermediates"][k], ref["_intermediates"][k])
        for k in ("points", "local_points", "conf", "camera_poses"):
            report(k, out[k], ref[k])
        print("  rot err deg", rot_err_deg(out["camera_poses"].cpu(), ref["camera_poses"]))


def full():
    cfg = Pi3Config()
    t0 = time.time()
    eng = Pi3Engine(cfg, "cuda:0")
    torch.cuda.synchronize()
    print(f"engine init (recipe weights on device): {time.time() - t0:.1f}s")
    gdir = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests", "golden")
    for name, (B, N, H, W) in CASES.items():
        g = np.load(os.path.join(gdir, name + ".npz"))
        imgs = golden_images(name, B, N, H, W)
        out = eng.forward(imgs, return_intermediates=True)
        torch.cuda.synchronize()
        print("full config", name)
        for k in g.files:
            if k.startswith("i_"):
                report(k, out["_intermediates"][k[2:]], torch.from_numpy(g[k]))
        for k in ("points", "local_points", "conf", "camera_poses"):
            report(k, out[k], torch.from_numpy(g[k]))
        print("  rot err deg", rot_err_deg(out["camera_poses"].cpu(), torch.from_numpy(g["camera_poses"])))
    return eng


def timed(eng):
    imgs = torch.rand(1, 100, 3, 308, 406, device="cuda:0")
    for it in range(3):
        torch.cuda.synchronize()
        t0 = time.time()
        out = eng.forward(imgs)
        torch.cuda.synchronize()
        dt = time.time() - t0
        fl = eng.flops(1, 100, 308, 406)["total"]
        print(f"forward N=100 308x406: {dt * 1e3:.1f} ms  {100 / dt:.1f} frames/s  {fl / dt / 1e12:.1f} TF/s"
              f"  finite={bool(torch.isfinite(out['points']).all())}")
    print(f"peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")


if __name__ == "__main__":
    small()
    if "--full" in sys.argv:
        eng = full()
        if "--time" in sys.argv:
            timed(eng)
