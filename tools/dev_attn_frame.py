"""Frame-wise attention (100 x 643 and 100 x 645 tokens, 16 heads) timed in both forms the engine uses: max |k|^2 supplied
(decoder / head blocks: bounded-score loop) and not supplied (encoder blocks: online max), plus the global shape;
knobs via PI3_ATTN_* (e.g. PI3_ATTN_PRIO=1).  Interleaved rounds in ONE process, median and min reported."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pi3_slam_amd import lib, ops
dev = torch.device("cuda:0")
torch.manual_seed(0)

def attn_ref(qkv, B, S, H):
    q, k, v = qkv.float().view(B, S, 3, H, 64).permute(2, 0, 3, 1, 4)
    p = torch.softmax((q @ k.transpose(-1, -2)) * math.log(2.0), dim=-1)
    return (p @ v).permute(0, 2, 1, 3).reshape(B * S, H * 64)

def ev_time(fn, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

cases = {}
for (B, S, H) in [(100, 643, 16), (100, 645, 16)]:
    qkv = torch.randn(B * S, 3 * H * 64, device=dev); qkv[:, :H * 64] *= ops.QSCALE * 2; qkv = qkv.bfloat16()
    out = torch.empty(B * S, H * 64, device=dev, dtype=torch.bfloat16)
    k = qkv.view(B, S, 3, H, 64)[:, :, 1].float()
    k2 = (k * k).sum(-1).amax(1).reshape(-1).contiguous()
    ops.attention(qkv, out, B, S, H, k2max=k2)
    r = attn_ref(qkv[:3 * S], 3, S, H)
    print("check", (B, S, H), ((out[:3 * S].float() - r).abs().max() / r.abs().max()).item())
    cases[f"frame S={S} k2max"] = (lambda qkv=qkv, out=out, B=B, S=S, H=H, k2=k2: ops.attention(qkv, out, B, S, H, k2max=k2), 4.0 * B * H * S * S * 64)
    cases[f"frame S={S} online"] = (lambda qkv=qkv, out=out, B=B, S=S, H=H: ops.attention(qkv, out, B, S, H), 4.0 * B * H * S * S * 64)

    def nw2(qkv=qkv, out=out, B=B, S=S, H=H, k2=k2):        # two-wave workgroups (knob attn_frame_nw), same launch otherwise
        lib.set_knob("attn_frame_nw", 2)
        ops.attention(qkv, out, B, S, H, k2max=k2)
        lib.set_knob("attn_frame_nw", 4)
    nw2()
    print("check nw2", (B, S, H), ((out[:3 * S].float() - r).abs().max() / r.abs().max()).item())
    cases[f"frame S={S} k2max nw2"] = (nw2, 4.0 * B * H * S * S * 64)
B, S, H = 1, 64300, 16
qkvg = torch.randn(S, 3 * H * 64, device=dev); qkvg[:, :H * 64] *= ops.QSCALE * 2; qkvg = qkvg.bfloat16()
outg = torch.empty(S, H * 64, device=dev, dtype=torch.bfloat16)
kg = qkvg.view(S, 3, H, 64)[:, 1].float()
k2g = (kg * kg).sum(-1).amax(0).contiguous()
cases["global S=64300 k2max"] = (lambda: ops.attention(qkvg, outg, 1, S, H, k2max=k2g), 4.0 * H * S * S * 64)
for fn, _ in cases.values():
    fn(); fn()
torch.cuda.synchronize()
res = {k: [] for k in cases}
for rnd in range(5):
    for k, (fn, fl) in cases.items():
        res[k].append(ev_time(fn, 10 if "frame" in k else 3))
for k, (fn, fl) in cases.items():
    v = sorted(res[k])
    print(f"PRIO={os.environ.get('PI3_ATTN_PRIO','0')} TAILOPT={os.environ.get('PI3_ATTN_TAILOPT','1')} CARRY={os.environ.get('PI3_ATTN_CARRY','1')} {k:24s}: median {v[len(v)//2]:.4f} ms  min {v[0]:.4f} ms  {fl / v[len(v)//2] / 1e9:.0f} TF/s")
