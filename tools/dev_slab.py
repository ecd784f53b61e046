"""Does running the row-local part of a transformer block (proj -> LN -> fc1+GELU -> fc2 -> LN -> qkv) slab by slab keep
its working set in the 256 MB memory-side cache and pay?  Full M = 64 300 per kernel (breadth first, as the engine
does) against 2 / 4 / 8 row slabs run depth first, same kernels, same total work; captured as hipGraphs so that the
extra launches cost what they cost in the product path."""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pi3_slam_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
M, D = 64300, 1024
bf = torch.bfloat16
x = torch.randn(M, D, device=dev)
attn_out = torch.randn(M, D, device=dev).to(bf)
xn = torch.empty(M, D, device=dev, dtype=bf)
h = torch.empty(M, 4 * D, device=dev, dtype=bf)
qkv = torch.empty(M, 3 * D, device=dev, dtype=bf)
w = {k: (torch.randn(n, kk, device=dev) / math.sqrt(kk)).to(bf) for k, (n, kk) in
     dict(proj=(D, D), fc1=(4 * D, D), fc2=(D, 4 * D), qkv=(3 * D, D)).items()}
b = {k: torch.randn(v.shape[0], device=dev) * 0.1 for k, v in w.items()}
g1, g2 = torch.rand(D, device=dev) * 0.1, torch.rand(D, device=dev) * 0.1
lw, lb = torch.rand(D, device=dev) + 0.5, torch.randn(D, device=dev) * 0.1


def chain(r0, r1):
    xs, a, n, hh, q = x[r0:r1], attn_out[r0:r1], xn[r0:r1], h[r0:r1], qkv[r0:r1]
    R = r1 - r0
    ops.gemm(a, w["proj"], xs, M=R, bias=b["proj"], gamma=g1, resid=xs)
    ops.layernorm(xs, lw, lb, n, 1e-6, rows=R)
    ops.gemm(n, w["fc1"], hh, M=R, bias=b["fc1"], act=ops.ACT_GELU)
    ops.gemm(hh, w["fc2"], xs, M=R, bias=b["fc2"], gamma=g2, resid=xs)
    ops.layernorm(xs, lw, lb, n, 1e-6, rows=R)
    ops.gemm(n, w["qkv"], q, M=R, bias=b["qkv"])


def run(nslab):
    # slab boundaries on multiples of 256 rows (whole GEMM row panels)
    panels = (M + 255) // 256
    cuts = [min(M, ((panels * i) // nslab) * 256) for i in range(nslab)] + [M]
    for i in range(nslab):
        chain(cuts[i], cuts[i + 1])


def timed(nslab, reps=30):
    run(nslab)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        run(nslab)
    graph.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        graph.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for rnd in range(4):
    for nslab in (1, 2, 4):
        print(f"slabs {nslab}: {timed(nslab):.3f} ms for proj+LN+fc1+fc2+LN+qkv over M = {M}", flush=True)
