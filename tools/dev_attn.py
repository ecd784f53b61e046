"""Time the attention kernel at the north-star shapes (variant via PI3_ATTN_WAVES) + correctness spot check."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pi3_slam_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)

def attn_ref(qkv, B, S, H):
    q, k, v = qkv.float().view(B, S, 3, H, 64).permute(2, 0, 3, 1, 4)
    p = torch.softmax((q @ k.transpose(-1, -2)) * math.log(2.0), dim=-1)
    return (p @ v).permute(0, 2, 1, 3).reshape(B * S, H * 64)

for (B, S, H) in [(2, 643, 2), (1, 3000, 2)]:
    qkv = torch.randn(B * S, 3 * H * 64, device=dev); qkv[:, :H*64] *= ops.QSCALE * 2; qkv = qkv.bfloat16()
    out = torch.empty(B * S, H * 64, device=dev, dtype=torch.bfloat16)
    ops.attention(qkv, out, B, S, H)
    r = attn_ref(qkv, B, S, H)
    print("check", (B, S, H), ((out.float() - r).abs().max() / r.abs().max()).item())
for (B, S, H) in [(100, 643, 16), (1, 64300, 16)]:
    qkv = (torch.randn(B * S, 3 * H * 64, device=dev) * 0.5).bfloat16()
    out = torch.empty(B * S, H * 64, device=dev, dtype=torch.bfloat16)
    for _ in range(2): ops.attention(qkv, out, B, S, H)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 5
    e0.record()
    for _ in range(n): ops.attention(qkv, out, B, S, H)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print(f"PI3_ATTN_WAVES={os.environ.get('PI3_ATTN_WAVES','-')} attn B={B} S={S}: {ms:.3f} ms  {4.0*B*H*S*S*64/ms/1e9:.1f} TF/s")
