"""Is the block GEMM power-limited?  The same launches on random and on all-zero operands (identical instruction stream;
zeros toggle no multiplier inputs): a faster all-zero run means the part trades power for clock in this kernel too."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pi3_slam_amd import ops
dev = torch.device("cuda:0")
M = 64300
for (N, K, name) in ((3072, 1024, "qkv"), (1024, 1024, "proj"), (4096, 1024, "fc1"), (1024, 4096, "fc2")):
    res = {}
    for kind in ("random", "zeros", "random"):
        a = (torch.randn(M, K, device=dev) if kind == "random" else torch.zeros(M, K, device=dev)).bfloat16()
        w = ((torch.randn(N, K, device=dev) / K ** 0.5) if kind == "random" else torch.zeros(N, K, device=dev)).bfloat16()
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        bias = torch.zeros(N, device=dev)
        for _ in range(5):
            ops.gemm(a, w, out, bias=bias)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30):
            ops.gemm(a, w, out, bias=bias)
        e1.record()
        torch.cuda.synchronize()
        res.setdefault(kind, []).append(e0.elapsed_time(e1) / 30)
    r, z = min(res["random"]), res["zeros"][0]
    lib = {}
    for kind in ("random", "zeros"):
        a = (torch.randn(M, K, device=dev) if kind == "random" else torch.zeros(M, K, device=dev)).bfloat16()
        w = ((torch.randn(N, K, device=dev) / K ** 0.5) if kind == "random" else torch.zeros(N, K, device=dev)).bfloat16()
        for _ in range(5):
            torch.matmul(a, w.t())
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30):
            torch.matmul(a, w.t())
        e1.record()
        torch.cuda.synchronize()
        lib[kind] = e0.elapsed_time(e1) / 30
    print(f"{name:5s} M={M} N={N} K={K}: random operands {r:.4f} ms ({2.0 * M * N * K / r / 1e9:.0f} TF/s), all-zero operands {z:.4f} ms "
          f"({2.0 * M * N * K / z / 1e9:.0f} TF/s): {100 * (r / z - 1):+.1f} %;  torch.matmul (hipBLASLt, no bias/epilogue) "
          f"random {lib['random']:.4f} ms ({2.0 * M * N * K / lib['random'] / 1e9:.0f} TF/s), zeros {lib['zeros']:.4f} ms "
          f"({2.0 * M * N * K / lib['zeros'] / 1e9:.0f} TF/s)")
