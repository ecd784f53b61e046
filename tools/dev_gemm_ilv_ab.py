"""A/B of gemm256's interleaved K loop (knob gemm_ilv = 1) against the shipped ping-pong loop in ONE process: bit-identity of
every block GEMM (fused qkv included) at M = 64 300, then interleaved timing rounds.

    python tools/dev_gemm_ilv_ab.py [rounds] [launches]
"""
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
g = torch.Generator(device=dev).manual_seed(1234)
rn = lambda *s: torch.randn(*s, device=dev, generator=g)   # noqa: E731
T, ph, pw = 643, 22, 29
pos = torch.zeros(T, 2, dtype=torch.int32)
yy, xx = torch.meshgrid(torch.arange(ph), torch.arange(pw), indexing="ij")
pos[5:, 0] = (yy.reshape(-1) + 1).to(torch.int32)
pos[5:, 1] = (xx.reshape(-1) + 1).to(torch.int32)
inv_freq = 1.0 / (100.0 ** (torch.arange(0, 32, 2).float() / 32))
ang = torch.einsum("i,j->ij", torch.arange(max(ph, pw) + 1).float(), inv_freq)
cs = torch.stack([ang.cos(), ang.sin()], dim=-1).contiguous().to(dev)
pos = pos.to(dev)
runs = {}
for (N, K, kind) in [(3072, 1024, "qkv"), (3072, 1024, "qkv_fused"), (1024, 1024, "proj"), (4096, 1024, "fc1"), (1024, 4096, "fc2")]:
    a = rn(M, K).bfloat16()
    w = (rn(N, K) / math.sqrt(K)).bfloat16()
    bias = rn(N)
    gamma = torch.rand(N, device=dev, generator=g)
    reset = lambda: None   # noqa: E731
    if kind in ("proj", "fc2"):
        x0 = rn(M, N)
        out = x0.clone()
        fn = (lambda a=a, w=w, out=out, bias=bias, gamma=gamma: ops.gemm(a, w, out, bias=bias, gamma=gamma, resid=out))
        reset = (lambda out=out, x0=x0: out.copy_(x0))
    elif kind == "fc1":
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        fn = (lambda a=a, w=w, out=out, bias=bias: ops.gemm(a, w, out, bias=bias, act=ops.ACT_GELU))
    elif kind == "qkv_fused":
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        qw, qb, kw, kb = (rn(64) for _ in range(4))
        k2 = torch.zeros((M // T) * 16, device=dev)
        fn = (lambda a=a, w=w, out=out, bias=bias, qw=qw, qb=qb, kw=kw, kb=kb, k2=k2:
              ops.gemm_qkv(a, w, out, M=M // T * T, H=16, bias=bias, T=T, pos=pos, cs=cs, qw=qw, qb=qb, kw=kw, kb=kb,
                           k2max=k2, attn_B=M // T, attn_S=T))
    else:
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        fn = (lambda a=a, w=w, out=out, bias=bias: ops.gemm(a, w, out, bias=bias))
    runs[kind] = (fn, reset, out)

VARS = (0, 1, 2, 3)     # 0 shipped ping-pong; 1 interleaved; 2 interleaved + waves 4-7 at priority 1; 3 + alternating priority
NAMES = {0: "ping-pong", 1: "ilv", 2: "ilv prio-static", 3: "ilv prio-alt"}
bad = 0
for kind, (fn, reset, out) in runs.items():
    got = []
    for v in VARS:
        lib.set_knob("gemm_ilv", v)
        reset()
        fn()
        torch.cuda.synchronize()
        got.append(out.clone())
    it = torch.int16 if out.dtype == torch.bfloat16 else torch.int32
    for v, g_ in zip(VARS[1:], got[1:]):
        if not torch.equal(got[0].view(it), g_.view(it)):
            bad += 1
            d = (got[0].float() - g_.float()).abs()
            print(f"!! {kind}: gemm_ilv={v} differs: max {d.max().item():.3e}, {(d > 0).float().mean().item():.3e} of the elements")
print("BIT-IDENTITY", "ok" if bad == 0 else f"FAILED on {bad} shapes")


def timed(fn, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


res = {(k, v): [] for k in runs for v in VARS}
for r in range(R):
    for kind, (fn, reset, out) in runs.items():
        for v in VARS:
            lib.set_knob("gemm_ilv", v)
            res[(kind, v)].append(timed(fn, NL))
lib.set_knob("gemm_ilv", 0)
tot = {v: 0.0 for v in VARS}
for kind in runs:
    ts = {v: statistics.median(res[(kind, v)]) for v in VARS}
    print(f"{kind:10s} " + "   ".join(f"{NAMES[v]} {ts[v]:.4f}" for v in VARS))
    if kind != "qkv":
        for v in VARS:
            tot[v] += ts[v]
print("block (fused qkv + proj + fc1 + fc2): " + "   ".join(f"{NAMES[v]} {tot[v]:.4f} ms ({100 * (tot[v] / tot[0] - 1):+.1f} %)" for v in VARS))
