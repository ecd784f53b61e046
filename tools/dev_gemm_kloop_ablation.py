"""Where a K tile of gemm256's main loop spends its time: timing-only ablations of the DEVELOPMENT library
(make -C pi3_slam_amd/csrc dev; results of the ablated variants are wrong by construction).

    PI3_LIB_PATH=pi3_slam_amd/libpi3slam_hip_dev.so python tools/dev_gemm_kloop_ablation.py [rounds] [launches]

knob gemm_abl (bits): 1 = no epilogue, 2 = no LDS-DMA issues in the K loop, 4 = no fragment reads from LDS, 8 = no counted
vmcnt waits in the loop (the LDS-DMA is issued from inline asm, so hipcc adds no waits of its own).
"""
import math
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from pi3_slam_amd import lib, ops

assert "dev" in os.path.basename(lib.LIB_PATH), "run with PI3_LIB_PATH=.../libpi3slam_hip_dev.so"
dev = torch.device("cuda:0")
torch.manual_seed(0)
R = int(sys.argv[1]) if len(sys.argv) > 1 else 5
NL = int(sys.argv[2]) if len(sys.argv) > 2 else 20
M = int(os.environ.get("AB_M", "64300"))
VARIANTS = [("full", 0), ("no epilogue", 1), ("no epi, no DMA", 3), ("no epi, no reads", 5), ("no epi, MFMA + barriers only", 7),
            ("no epi, no counted waits", 9), ("no epi, no reads, no counted waits", 13)]
shapes = [(3072, 1024, "qkv"), (1024, 4096, "fc2 shape, bf16 out")]
fns = {}
for (N, K, kind) in shapes:
    a = torch.randn(M, K, device=dev).bfloat16()
    w = (torch.randn(N, K, device=dev) / math.sqrt(K)).bfloat16()
    bias = torch.randn(N, device=dev)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    fns[kind] = (lambda a=a, w=w, out=out, bias=bias: ops.gemm(a, w, out, bias=bias)), N, K


def timed(fn, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


res = {}
for kind, (fn, N, K) in fns.items():
    for name, bits in VARIANTS:
        lib.set_knob("gemm_abl", bits)
        fn()
torch.cuda.synchronize()
for r in range(R):
    for kind, (fn, N, K) in fns.items():
        for name, bits in VARIANTS:
            lib.set_knob("gemm_abl", bits)
            res.setdefault((kind, name), []).append(timed(fn, NL))
lib.set_knob("gemm_abl", 0)
for kind, (fn, N, K) in fns.items():
    tiles = math.ceil(M / 256) * (N // 256)
    per_wg = math.ceil(tiles / 256)
    print(f"{kind}: M={M} N={N} K={K}: {tiles} tiles, {per_wg} per workgroup, {K // 64} K tiles each")
    for name, bits in VARIANTS:
        t = statistics.median(res[(kind, name)])
        line = f"  {name:32s} {t:.4f} ms"
        if bits & 1:
            line += f"   {1e3 * t / (per_wg * (K // 64)):.3f} us per K tile and workgroup"
        print(line)
