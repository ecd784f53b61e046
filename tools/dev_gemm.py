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
