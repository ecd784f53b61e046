"""Same inputs through the block GEMMs of whichever library PI3_LIB_PATH names: prints a digest of every output (to
compare two builds bit for bit across processes) and interleaved timings.

    PI3_LIB_PATH=pi3_slam_amd/libpi3slam_hip_X.so python tools/dev_gemm_libcmp.py [rounds] [launches]
"""
import hashlib
import math
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from pi3_slam_amd import lib, ops

dev = torch.device("cuda:0")
R = int(sys.argv[1]) if len(sys.argv) > 1 else 5
NL = int(sys.argv[2]) if len(sys.argv) > 2 else 20
M = int(os.environ.get("AB_M", "64300"))
print("library:", os.path.basename(lib.LIB_PATH), "M =", M)
g = torch.Generator(device=dev).manual_seed(1234)
rn = lambda *s: torch.randn(*s, device=dev, generator=g)   # noqa: E731
shapes = [(3072, 1024, "qkv"), (3072, 1024, "qkv_fused"), (1024, 1024, "proj"), (4096, 1024, "fc1"), (1024, 4096, "fc2")]
T = 643
ph, pw = 22, 29
pos = torch.zeros(T, 2, dtype=torch.int32)
yy, xx = torch.meshgrid(torch.arange(ph), torch.arange(pw), indexing="ij")
pos[5:, 0] = (yy.reshape(-1) + 1).to(torch.int32)
pos[5:, 1] = (xx.reshape(-1) + 1).to(torch.int32)
inv_freq = 1.0 / (100.0 ** (torch.arange(0, 32, 2).float() / 32))
ang = torch.einsum("i,j->ij", torch.arange(max(ph, pw) + 1).float(), inv_freq)
cs = torch.stack([ang.cos(), ang.sin()], dim=-1).contiguous().to(dev)
pos = pos.to(dev)
runs = {}
for (N, K, kind) in shapes:
    a = rn(M, K).bfloat16()
    w = (rn(N, K) / math.sqrt(K)).bfloat16()
    bias = rn(N)
    gamma = torch.rand(N, device=dev, generator=g)
    if kind in ("proj", "fc2"):
        x0 = rn(M, N)
        out = x0.clone()
        fn = (lambda a=a, w=w, out=out, bias=bias, gamma=gamma: ops.gemm(a, w, out, bias=bias, gamma=gamma, resid=out))
        reset = (lambda out=out, x0=x0: out.copy_(x0))
    elif kind == "fc1":
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        fn = (lambda a=a, w=w, out=out, bias=bias: ops.gemm(a, w, out, bias=bias, act=ops.ACT_GELU))
        reset = lambda: None   # noqa: E731
    elif kind == "qkv_fused":
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        qw, qb, kw, kb = (rn(64) for _ in range(4))
        k2 = torch.zeros((M // T) * 16, device=dev)
        fn = (lambda a=a, w=w, out=out, bias=bias, qw=qw, qb=qb, kw=kw, kb=kb, k2=k2:
              ops.gemm_qkv(a, w, out, M=M // T * T, H=16, bias=bias, T=T, pos=pos, cs=cs, qw=qw, qb=qb, kw=kw, kb=kb,
                           k2max=k2, attn_B=M // T, attn_S=T))
        reset = lambda: None   # noqa: E731
    else:
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        fn = (lambda a=a, w=w, out=out, bias=bias: ops.gemm(a, w, out, bias=bias))
        reset = lambda: None   # noqa: E731
    runs[kind] = (fn, reset, out)

for kind, (fn, reset, out) in runs.items():
    reset()
    fn()
    torch.cuda.synchronize()
    h = hashlib.sha1(out.view(torch.uint8).cpu().numpy().tobytes()).hexdigest()[:16]
    print(f"digest {kind:10s} {h}")


def timed(fn, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


res = {k: [] for k in runs}
for r in range(R):
    for kind, (fn, reset, out) in runs.items():
        res[kind].append(timed(fn, NL))
tot = 0.0
for kind in runs:
    t = statistics.median(res[kind])
    print(f"time   {kind:10s} {t:.4f} ms (min {min(res[kind]):.4f})")
    if kind != "qkv":
        tot += t
print(f"block (fused qkv + proj + fc1 + fc2) {tot:.4f} ms")
