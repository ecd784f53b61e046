"""Interleaved A/B of gemm256 variants in ONE process (run-time knobs through pi3_set_knob): per shape of a transformer
block at M = 64 300, R rounds of N launches per variant, median and min of the per-round means.

    python tools/dev_gemm_ab.py [rounds] [launches]

Variants (every one computes the same result; the check at the start compares each against variant 0):
  base               shipped defaults
  gelu_as            gelu_form = 1: the round 1-3 Abramowitz-Stegun GELU (fc1 only)
  4w                 gemm_4w = 1: four waves per workgroup, 128 x 128 block per wave, accumulators in AGPRs
  4w-ilv             gemm_4w = 2: the same with fragment reads and LDS-DMA issues interleaved between the MFMAs (all asm)
  stag<N>            gemm_stagger_ns = N: workgroup b starts b * N ns late                      (AB_ALL=1)
  rpref              gemm_rpref = 1: residual lines touched during the last K tiles (proj, fc2) (AB_ALL=1)
"""
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from pi3_slam_amd import lib, ops

dev = torch.device("cuda:0")
torch.manual_seed(0)
R = int(sys.argv[1]) if len(sys.argv) > 1 else 5
NL = int(sys.argv[2]) if len(sys.argv) > 2 else 20
M = int(os.environ.get("AB_M", "64300"))
KNOBS = ("gelu_form", "gemm_stagger_ns", "gemm_rpref", "gemm_4w")
VARIANTS = [("base", {}), ("gelu_as", {"gelu_form": 1}), ("4w", {"gemm_4w": 1}), ("4w-ilv", {"gemm_4w": 2})]
if os.environ.get("AB_ALL"):      # the round-4 experiments that found nothing (profiles/EXPERIMENTS.md)
    VARIANTS += [("stag30", {"gemm_stagger_ns": 30}), ("stag120", {"gemm_stagger_ns": 120}), ("rpref", {"gemm_rpref": 1})]


def set_variant(kn):
    for k in KNOBS:
        lib.set_knob(k, kn.get(k, 0))


shapes = [(3072, 1024, "qkv"), (1024, 1024, "proj"), (4096, 1024, "fc1"), (1024, 4096, "fc2")]
ops_ = {}
for (N, K, kind) in shapes:
    a = torch.randn(M, K, device=dev).bfloat16()
    w = (torch.randn(N, K, device=dev) / math.sqrt(K)).bfloat16()
    bias = torch.randn(N, device=dev)
    gamma = torch.rand(N, device=dev)
    if kind in ("proj", "fc2"):
        x0 = torch.randn(M, N, device=dev)
        out = x0.clone()
        fn = (lambda a=a, w=w, out=out, bias=bias, gamma=gamma: ops.gemm(a, w, out, bias=bias, gamma=gamma, resid=out))
        reset = (lambda out=out, x0=x0: out.copy_(x0))
    elif kind == "fc1":
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        fn = (lambda a=a, w=w, out=out, bias=bias: ops.gemm(a, w, out, bias=bias, act=ops.ACT_GELU))
        reset = lambda: None
    else:
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        fn = (lambda a=a, w=w, out=out, bias=bias: ops.gemm(a, w, out, bias=bias))
        reset = lambda: None
    ops_[kind] = (fn, reset, out)

# ---- every variant gives the same result (gelu_as: the other GELU form, compared at bf16 resolution)
for kind, (fn, reset, out) in ops_.items():
    ref = None
    for name, kn in VARIANTS:
        set_variant(kn)
        reset()
        fn()
        torch.cuda.synchronize()
        got = out.float().clone()
        if ref is None:
            ref = got
        else:
            d = (got - ref).abs().max().item()
            same = torch.equal(got, ref)
            frac = (got != ref).float().mean().item()
            tag = "identical" if same else f"max |diff| {d:.3e}, {frac:.2e} of the elements differ"
            if not same and not (name == "gelu_as" and kind == "fc1" and frac < 2e-3):
                print(f"!! {kind} {name}: {tag}")
            elif not same:
                print(f"   {kind} {name}: {tag} (the two GELU forms round differently on a few elements)")
print("variant check done")


def time_variant(fn, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


res = {(k, v[0]): [] for k in ops_ for v in VARIANTS}
for kind, (fn, reset, out) in ops_.items():
    for name, kn in VARIANTS:      # warm every instance
        set_variant(kn)
        fn()
    torch.cuda.synchronize()
for r in range(R):
    for kind, (fn, reset, out) in ops_.items():
        for name, kn in VARIANTS:
            if name == "gelu_as" and kind != "fc1":
                continue
            if name.startswith("rpref") and kind not in ("proj", "fc2"):
                continue
            set_variant(kn)
            res[(kind, name)].append(time_variant(fn, NL))
set_variant({})
import statistics

print(f"{'variant':14s}" + "".join(f"{k:>18s}" for k in ops_))
tot = {}
for name, _ in VARIANTS:
    row = f"{name:14s}"
    for kind in ops_:
        v = res[(kind, name)]
        if v:
            row += f"   {statistics.median(v):.4f} ({min(v):.4f})"
            tot.setdefault(name, 0.0)
            tot[name] += statistics.median(v)
        else:
            row += " " * 18
            tot.setdefault(name, 0.0)
            tot[name] += statistics.median(res[(kind, "base")])
    print(row + f"   block {tot[name]:.4f} ms")
