"""Correctness + timing of the GEMM kernels at the north-star shapes (PI3_GEMM_IMPL=1 forces the 128x128 kernel)."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pi3_slam_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)

def rel(a, b):
    a, b = a.float(), b.float()
    return ((a - b).abs().max() / (b.abs().max() + 1e-12)).item()

for (M, N, K) in [(1500, 512, 128), (2049, 1024, 1024), (5000, 256, 4096)]:
    a = torch.randn(M, K, device=dev).bfloat16()
    w = (torch.randn(N, K, device=dev) / math.sqrt(K)).bfloat16()
    bias, gamma, resid = torch.randn(N, device=dev), torch.rand(N, device=dev) + 0.5, torch.randn(M, N, device=dev)
    ref = a.float() @ w.float().T + bias
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    ops.gemm(a, w, out, bias=bias)
    e1 = rel(out, ref)
    ops.gemm(a, w, out, bias=bias, act=ops.ACT_GELU)
    e2 = rel(out, torch.nn.functional.gelu(ref))
    o32 = resid.clone()
    ops.gemm(a, w, o32, bias=bias, gamma=gamma, resid=o32)
    e3 = rel(o32, resid + gamma * ref)
    print("check", (M, N, K), e1, e2, e3)

def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

M = 64300
tot = 0.0
for (N, K, kind) in [(3072, 1024, "qkv"), (1024, 1024, "proj"), (4096, 1024, "fc1"), (1024, 4096, "fc2")]:
    a = torch.randn(M, K, device=dev).bfloat16()
    w = (torch.randn(N, K, device=dev) / math.sqrt(K)).bfloat16()
    bias = torch.randn(N, device=dev); gamma = torch.rand(N, device=dev)
    if kind in ("proj", "fc2"):
        out = torch.randn(M, N, device=dev)
        fn = lambda: ops.gemm(a, w, out, bias=bias, gamma=gamma, resid=out)
    elif kind == "fc1":
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        fn = lambda: ops.gemm(a, w, out, bias=bias, act=ops.ACT_GELU)
    else:
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        fn = lambda: ops.gemm(a, w, out, bias=bias)
    ms = timeit(fn)
    tot += ms
    print(f"IMPL={os.environ.get('PI3_GEMM_IMPL','0')} {kind:5s} M={M} N={N} K={K}: {ms:.3f} ms  {2.0*M*N*K/ms/1e9:.1f} TF/s")
print(f"IMPL={os.environ.get('PI3_GEMM_IMPL','0')} block total {tot:.3f} ms")

# fused q/k epilogue (decoder block: qk-norm + RoPE + scale + max|k|^2) against the two-pass form it replaces
H, T = 16, 643
a = torch.randn(M, 1024, device=dev).bfloat16()
w = (torch.randn(3072, 1024, device=dev) / 32).bfloat16()
bias = torch.randn(3072, device=dev) * 0.1
pos = torch.zeros(T, 2, dtype=torch.int32)
pos[5:, 0] = (torch.arange(T - 5) // 29 + 1).int(); pos[5:, 1] = (torch.arange(T - 5) % 29 + 1).int()
pos = pos.to(dev)
inv = 1.0 / (100.0 ** (torch.arange(0, 32, 2).float() / 32))
ang = torch.arange(30).float()[:, None] * inv[None]
cs = torch.stack([ang.cos(), ang.sin()], -1).contiguous().to(dev)
qw, qb, kw, kb = [(torch.randn(64) * 0.2 + (1 if i % 2 == 0 else 0)).to(dev) for i in range(4)]
qkv = torch.empty(M, 3072, device=dev, dtype=torch.bfloat16)
k2 = torch.empty(H, device=dev)
fused = lambda: ops.gemm_qkv(a, w, qkv, M=M, H=H, bias=bias, T=T, pos=pos, cs=cs, qw=qw, qb=qb, kw=kw, kb=kb, k2max=k2, attn_B=1, attn_S=M)
def two_pass():
    ops.gemm(a, w, qkv, M=M, bias=bias)
    ops.qknorm_rope(qkv, M, H, T, pos, cs, qw, qb, kw, kb, eps=1e-5)
print(f"qkv fused epilogue: {timeit(fused):.3f} ms   two-pass (gemm + qknorm_rope, no key-norm pre-pass): {timeit(two_pass):.3f} ms")
