"""Which attention calls of one pi3 forward (recipe weights, 100 frames 308x406) have waves whose result came from the
online-max loop, by call order: [call index, B, S, bounded waves, online-max waves].  Development diagnostic."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pi3_slam_amd import lib, ops
from pi3_slam_amd.engine import Pi3Engine
from pi3_slam_amd.weights import Pi3Config

dev = torch.device("cuda:0")
engine = Pi3Engine(Pi3Config(), str(dev))
imgs = torch.rand(1, 100, 3, 308, 406, device=dev)
cnt = torch.zeros(2, 2, 32, device=dev, dtype=torch.int32)
calls = []
orig = ops.attention


def spy(qkv, out, B, S, H, k2max=None):
    torch.cuda.synchronize()
    cnt.zero_()
    r = orig(qkv, out, B, S, H, k2max=k2max)
    torch.cuda.synchronize()
    w = cnt.sum(-1).cpu()
    kind = 0 if S >= 4096 else 1
    q = qkv.float().view(B, S, 3, H, 64)
    calls.append((B, S, int(w[kind, 0]), int(w[kind, 1]), float((q[:, :, 0].norm(dim=-1).amax())), float(q[:, :, 1].norm(dim=-1).amax())))
    return r


for knob in (1, 2):
    lib.set_knob("attn_nomax", knob)
    calls.clear()
    ops.attention = spy
    import pi3_slam_amd.vit as vit
    vit.ops.attention = spy
    ops.attention_path_counters(cnt)
    with torch.no_grad():
        engine.forward(imgs)
    torch.cuda.synchronize()
    ops.attention_path_counters(None)
    print(f"knob attn_nomax = {knob}: {len(calls)} attention calls")
    for i, c in enumerate(calls):
        if c[3]:
            print(f"  call {i:3d}  B={c[0]} S={c[1]}  bounded {c[2]}  online-max {c[3]}   max|q| {c[4]:.1f} (exp2 domain)  max|k| {c[5]:.1f}")
