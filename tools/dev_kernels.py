"""Developer check of the core kernels on a GPU box (not part of the pytest suite): correctness vs torch fp32
references + raw timings.  Usage: python tools/dev_kernels.py [--big]"""
import math
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from pi3_slam_amd import ops
from pi3_slam_amd.recipe import fnv1a64, recipe_tensor

dev = torch.device("cuda:0")
torch.manual_seed(0)


def rel_err(a, b):
    a, b = a.float(), b.float()
    return ((a - b).abs().max() / (b.abs().max() + 1e-12)).item(), ((a - b).abs().mean() / (b.abs().mean() + 1e-12)).item()


def timeit(fn, n=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def test_gemm():
    for (M, N, K) in [(300, 256, 128), (1000, 1024, 1024), (643 * 3, 384, 640)]:
        a = torch.randn(M, K, device=dev).bfloat16()
        w = (torch.randn(N, K, device=dev) / math.sqrt(K)).bfloat16()
        bias = torch.randn(N, device=dev)
        gamma = torch.rand(N, device=dev) + 0.5
        resid = torch.randn(M, N, device=dev)
        ref = a.float() @ w.float().T + bias
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        ops.gemm(a, w, out, bias=bias)
        print("gemm bf16->bf16", (M, N, K), rel_err(out, ref))
        ops.gemm(a, w, out, bias=bias, act=ops.ACT_GELU)
        print("gemm gelu", rel_err(out, torch.nn.functional.gelu(ref)))
        o32 = resid.clone()
        ops.gemm(a, w, o32, bias=bias, gamma=gamma, resid=o32)
        print("gemm resid f32", rel_err(o32, resid + gamma * ref))
        ops.gemm(a, w, out, bias=bias, qscale=0.5, qcols=128)
        ref2 = ref.clone()
        ref2[:, :128] *= 0.5
        print("gemm qscale", rel_err(out, ref2))
        # f32 operands
        af, wf = a.float() + 0.001 * torch.randn(M, K, device=dev), w.float()
        reff = (af.double() @ wf.double().T + bias.double()).float()
        o32 = torch.empty(M, N, device=dev)
        ops.gemm(af, wf, o32, bias=bias)
        print("gemm f32", rel_err(o32, reff))
        ops.gemm(af, wf, o32, bias=bias, act=ops.ACT_RELU, resid=resid)
        print("gemm f32 relu+resid", rel_err(o32, resid + torch.relu(reff)))
    # row remap + table
    F, P, T, N, K = 3, 6, 11, 128, 640
    a = torch.randn(F * P, K, device=dev).bfloat16()
    w = (torch.randn(N, K, device=dev) / math.sqrt(K)).bfloat16()
    bias = torch.randn(N, device=dev)
    tab = torch.randn(P, N, device=dev)
    out = torch.zeros(F * T, N, device=dev)
    ops.gemm(a, w, out, bias=bias, rpg=P, gstride=T, goff=5, addtab=tab)
    ref = (a.float() @ w.float().T + bias).view(F, P, N) + tab
    print("gemm remap", rel_err(out.view(F, T, N)[:, 5:], ref), out.view(F, T, N)[:, :5].abs().max().item())


def attn_ref(qkv, B, S, H):
    q, k, v = qkv.float().view(B, S, 3, H, 64).permute(2, 0, 3, 1, 4)
    s = (q @ k.transpose(-1, -2)) * math.log(2.0)
    p = torch.softmax(s, dim=-1)
    return (p @ v).permute(0, 2, 1, 3).reshape(B * S, H * 64)


def test_attn():
    for (B, S, H) in [(3, 643, 2), (1, 1500, 3), (2, 64, 1), (1, 7, 2), (1, 129, 1)]:
        qkv = torch.randn(B * S, 3 * H * 64, device=dev)
        qkv[:, :H * 64] *= ops.QSCALE * 2.0  # a bit peaky
        qkv = qkv.bfloat16()
        out = torch.empty(B * S, H * 64, device=dev, dtype=torch.bfloat16)
        ops.attention(qkv, out, B, S, H)
        ref = attn_ref(qkv, B, S, H)
        print("attn", (B, S, H), rel_err(out, ref))
    # forced-rescale case: one key spikes late in the sequence
    B, S, H = 1, 1000, 1
    qkv = torch.randn(B * S, 3 * 64, device=dev) * 0.3
    qkv[900, 64:128] = qkv[17, 0:64] * 40.0
    qkv = qkv.bfloat16()
    out = torch.empty(B * S, 64, device=dev, dtype=torch.bfloat16)
    ops.attention(qkv, out, B, S, H)
    print("attn spike", rel_err(out, attn_ref(qkv, B, S, H)))


def test_ln_rope():
    rows, D = 1001, 1024
    x = torch.randn(rows, D, device=dev) * 3 + 1
    w, b = torch.rand(D, device=dev) + 0.5, torch.randn(D, device=dev)
    out = torch.empty(rows, D, device=dev, dtype=torch.bfloat16)
    ops.layernorm(x, w, b, out)
    print("layernorm", rel_err(out, torch.nn.functional.layer_norm(x, (D,), w, b, 1e-6)))
    # qk-norm + rope
    H, T, F = 2, 11, 3
    rows = F * T
    qkv0 = torch.randn(rows, 3 * H * 64, device=dev).bfloat16()
    pos = torch.zeros(T, 2, dtype=torch.int32)
    for t in range(5, T):
        pos[t, 0] = (t - 5) // 3 + 1
        pos[t, 1] = (t - 5) % 3 + 1
    npos = 8
    inv = 1.0 / (100.0 ** (torch.arange(0, 32, 2).float() / 32))
    ang = torch.arange(npos).float()[:, None] * inv[None]
    cs = torch.stack([ang.cos(), ang.sin()], -1).contiguous()
    qw, qb, kw, kb = [torch.randn(64, device=dev) * 0.2 + (1 if i % 2 == 0 else 0) for i in range(4)]
    qkv = qkv0.clone()
    ops.qknorm_rope(qkv, rows, H, T, pos.to(dev), cs.to(dev), qw, qb, kw, kb)
    x = qkv0.float().view(rows, 3, H, 64)
    q = torch.nn.functional.layer_norm(x[:, 0], (64,), qw, qb, 1e-6)
    k = torch.nn.functional.layer_norm(x[:, 1], (64,), kw, kb, 1e-6)

    def rope(tk):
        tpos = pos.to(dev)[torch.arange(rows, device=dev) % T].long()
        out = torch.empty_like(tk)
        for half, col in ((0, 0), (1, 1)):
            seg = tk[..., 32 * half:32 * half + 32]
            c = ang.cos().to(dev)[tpos[:, col]][:, None, :]
            s = ang.sin().to(dev)[tpos[:, col]][:, None, :]
            c, s = torch.cat([c, c], -1), torch.cat([s, s], -1)
            rot = torch.cat([-seg[..., 16:], seg[..., :16]], -1)
            out[..., 32 * half:32 * half + 32] = seg * c + rot * s
        return out

    ref = x.clone()
    ref[:, 0] = rope(q) * ops.QSCALE
    ref[:, 1] = rope(k)
    print("qknorm_rope", rel_err(qkv.view(rows, 3, H, 64)[:, :2], ref[:, :2]),
          "v untouched:", torch.equal(qkv.view(rows, 3, H, 64)[:, 2], qkv0.view(rows, 3, H, 64)[:, 2]))


def test_recipe():
    for dt in (torch.float32, torch.bfloat16):
        out = torch.empty(100003, device=dev, dtype=dt)
        ops.recipe_fill(out, fnv1a64("decoder.3.attn.qkv.weight"), 0.01, 0.3)
        ref = torch.from_numpy(recipe_tensor("decoder.3.attn.qkv.weight", (100003,), 0.01, 0.3)).to(dt)
        print("recipe", dt, torch.equal(out.cpu(), ref))


def bench():
    M = 64300
    for (N, K) in [(3072, 1024), (1024, 1024), (4096, 1024), (1024, 4096)]:
        a = torch.randn(M, K, device=dev).bfloat16()
        w = (torch.randn(N, K, device=dev) / math.sqrt(K)).bfloat16()
        bias = torch.randn(N, device=dev)
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        ms = timeit(lambda: ops.gemm(a, w, out, bias=bias))
        print(f"gemm M={M} N={N} K={K}: {ms:.3f} ms  {2.0 * M * N * K / ms / 1e9:.1f} TF/s")
    for (B, S, H) in [(100, 643, 16), (1, 16075, 16), (1, 64300, 16)]:
        qkv = (torch.randn(B * S, 3 * H * 64, device=dev) * 0.5).bfloat16()
        out = torch.empty(B * S, H * 64, device=dev, dtype=torch.bfloat16)
        ms = timeit(lambda: ops.attention(qkv, out, B, S, H), n=3)
        print(f"attn B={B} S={S} H={H}: {ms:.3f} ms  {4.0 * B * H * S * S * 64 / ms / 1e9:.1f} TF/s")
    x = torch.randn(M, 1024, device=dev)
    w, b = torch.rand(1024, device=dev), torch.rand(1024, device=dev)
    out = torch.empty(M, 1024, device=dev, dtype=torch.bfloat16)
    ms = timeit(lambda: ops.layernorm(x, w, b, out))
    print(f"layernorm {M}x1024: {ms:.3f} ms  {M * 1024 * 6 / ms / 1e6:.1f} GB/s")


if __name__ == "__main__":
    test_recipe()
    test_gemm()
    test_attn()
    test_ln_rope()
    if "--big" in sys.argv:
        bench()
