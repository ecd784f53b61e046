"""Two chunk forwards at once on two streams (two engines, own workspaces) against the same two forwards back to back:
does the HBM-bound part of one chunk (LayerNorm, epilogues) hide under the matrix-bound part of the other?"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pi3_slam_amd.engine import Pi3Engine  # noqa: E402
from pi3_slam_amd.weights import Pi3Config  # noqa: E402

dev = torch.device("cuda:0")
N, H, W = 100, 308, 406
e1, e2 = Pi3Engine(Pi3Config(), str(dev)), Pi3Engine(Pi3Config(), str(dev))
x1 = torch.rand(1, N, 3, H, W, device=dev)
x2 = torch.rand(1, N, 3, H, W, device=dev)
s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)


def seq(n):
    for _ in range(n):
        e1.forward(x1)
        e2.forward(x2)


def par(n, offset_s=0.0):
    for _ in range(n):
        with torch.cuda.stream(s1):
            e1.forward(x1)
        with torch.cuda.stream(s2):
            e2.forward(x2)


def timed(fn, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn(n)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (2 * n) * 1e3


seq(1); par(1)
for rnd in range(2):
    print(f"back to back: {timed(seq, 3):.1f} ms per chunk   two streams: {timed(par, 3):.1f} ms per chunk", flush=True)
