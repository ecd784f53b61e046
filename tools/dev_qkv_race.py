"""Race screen for the fused qkv epilogue (RoPE tables staged in LDS at kernel start, permlane row sums, per-wave
atomicMax of max |k|^2): repeated launches on the same operands must agree bit for bit (packed qkv AND k2max), for the
full epilogue, the RoPE-only form (head blocks) and the max|k|^2-only form (encoder blocks), global and frame-wise splits."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pi3_slam_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(2)
H, T, K, F = 16, 643, 1024, 100
M = F * T
a = torch.randn(M, K, device=dev).bfloat16()
w = (torch.randn(3 * H * 64, K, device=dev) / math.sqrt(K)).bfloat16()
bias = torch.randn(3 * H * 64, device=dev) * 0.1
pos = torch.zeros(T, 2, dtype=torch.int32)
pos[5:, 0] = (torch.arange(T - 5) // 29 + 1).int(); pos[5:, 1] = (torch.arange(T - 5) % 29 + 1).int()
pos = pos.to(dev)
inv = 1.0 / (100.0 ** (torch.arange(0, 32, 2).float() / 32))
ang = torch.arange(30).float()[:, None] * inv[None]
cs = torch.stack([ang.cos(), ang.sin()], -1).contiguous().to(dev)
qw, qb, kw, kb = [(torch.randn(64) * 0.2 + (1 if i % 2 == 0 else 0)).to(dev) for i in range(4)]
bad = 0
for name, kwargs in (("norm+rope", dict(pos=pos, cs=cs, qw=qw, qb=qb, kw=kw, kb=kb)), ("rope", dict(pos=pos, cs=cs)), ("k2 only", {})):
    for attn_B, attn_S in ((1, M), (F, T)):
        first = None
        for r in range(25):
            qkv = torch.empty(M, 3 * H * 64, device=dev, dtype=torch.bfloat16)
            k2 = torch.full((attn_B * H,), -1.0, device=dev)
            ops.gemm_qkv(a, w, qkv, M=M, H=H, bias=bias, T=T, k2max=k2, attn_B=attn_B, attn_S=attn_S, **kwargs)
            if first is None:
                first = (qkv, k2)
            elif not (torch.equal(qkv, first[0]) and torch.equal(k2, first[1])):
                bad += 1
                print("MISMATCH", name, (attn_B, attn_S), "rep", r)
                break
        print("case", name, (attn_B, attn_S), "done", flush=True)
print("QKV EPILOGUE RACE SCREEN", "FAILED" if bad else "clean", bad)
