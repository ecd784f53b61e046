"""Randomised shape sweep of the GEMM / attention / LayerNorm entry points against fp32 torch references (run on the
GPU box: python tools/fuzz_kernels.py [seconds]).  Looks for edge-shape bugs the fixed test shapes do not reach:
ragged M against the 128 / 256 tiles, short and ragged sequences on both attention kernels, both softmax paths."""
import math, os, random, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pi3_slam_amd import ops
dev = torch.device("cuda:0")
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = random.Random(7)
torch.manual_seed(7)
bad = 0
n = {"gemm": 0, "attn": 0, "ln": 0}

def rel(a, b):
    return ((a.float() - b.float()).abs().max() / (b.float().abs().max() + 1e-12)).item()

def attn_ref(qkv, B, S, H):
    q, k, v = qkv.float().view(B, S, 3, H, 64).permute(2, 0, 3, 1, 4)
    p = torch.softmax((q @ k.transpose(-1, -2)) * math.log(2.0), dim=-1)
    return (p @ v).permute(0, 2, 1, 3).reshape(B * S, H * 64)

t_end = time.time() + budget
while time.time() < t_end:
    kind = rng.choice(["gemm", "gemm", "attn", "attn", "ln"])
    n[kind] += 1
    if kind == "gemm":
        M = rng.choice([rng.randint(1, 300), rng.randint(1000, 1300), rng.randint(1, 9000), 1024, 2048, 2049, 255, 257])
        N = rng.choice([128, 256, 384, 512, 640, 768, 1024, 3072])
        K = 64 * rng.randint(1, 20)
        a = torch.randn(M, K, device=dev).bfloat16()
        w = (torch.randn(N, K, device=dev) / math.sqrt(K)).bfloat16()
        bias, gamma = torch.randn(N, device=dev), torch.rand(N, device=dev) + 0.5
        ref = a.float() @ w.float().T + bias
        form = rng.randint(0, 2)
        if form == 0:
            out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
            ops.gemm(a, w, out, bias=bias)
            e, tol = rel(out, ref), 6e-3
        elif form == 1:
            out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
            ops.gemm(a, w, out, bias=bias, act=ops.ACT_GELU)
            e, tol = rel(out, torch.nn.functional.gelu(ref)), 6e-3
        else:
            x0 = torch.randn(M, N, device=dev)
            out = x0.clone()
            ops.gemm(a, w, out, bias=bias, gamma=gamma, resid=out)
            e, tol = rel(out, x0 + gamma * ref), 3e-5
        if not (e < tol):
            bad += 1
            print("GEMM FAIL", (M, N, K), "form", form, e, flush=True)
    elif kind == "attn":
        H = rng.choice([1, 2, 3, 16])
        if rng.random() < 0.5:
            B, S = rng.randint(1, 4), rng.randint(1, 1500)
        else:
            B, S = 1, rng.choice([4096, 4097, 4160, 5000, 6143, 6144, 8191, rng.randint(4096, 9000)])
        if B * S * H > 40000:
            H = 1 if S > 4096 else H
        scale = rng.choice([0.3, 0.5, 1.0, 2.5])       # 2.5 pushes |q||k| past the bounded-score threshold
        qkv = torch.randn(B * S, 3 * H * 64, device=dev)
        qkv[:, :2 * H * 64] *= scale
        if rng.random() < 0.3:
            qkv[rng.randrange(B * S), H * 64: H * 64 + 64] *= 6.0      # one heavy key: online-max rescale / bound miss
        qkv = qkv.bfloat16()
        out = torch.empty(B * S, H * 64, device=dev, dtype=torch.bfloat16)
        k2 = None
        if S >= 256 and rng.random() < 0.5:      # max |k|^2 supplied, as the qkv epilogue does: bounded-score loop on the
            kk = qkv.view(B, S, 3, H, 64)[:, :, 1].float()      # four-wave kernel too (missing query blocks / key halves)
            k2 = (kk * kk).sum(-1).amax(1).reshape(-1).contiguous()
        ops.attention(qkv, out, B, S, H, k2max=k2)
        ref = attn_ref(qkv, B, S, H)
        e = rel(out, ref)
        if not (e < 1e-2) or not torch.isfinite(out.float()).all():
            bad += 1
            print("ATTN FAIL", (B, S, H), "scale", scale, e, flush=True)
    else:
        rows, D = rng.randint(1, 3000), rng.choice([128, 256, 384, 512, 768, 1024, 2048])
        x = torch.randn(rows, D, device=dev) * rng.choice([0.1, 1.0, 30.0]) + rng.choice([0.0, 5.0])
        w, b = torch.rand(D, device=dev) + 0.5, torch.randn(D, device=dev)
        ref = torch.nn.functional.layer_norm(x, (D,), w, b, 1e-6)
        o = torch.empty(rows, D, device=dev, dtype=torch.bfloat16)
        ops.layernorm(x, w, b, o)
        e = rel(o, ref)
        if not (e < 6e-3):
            bad += 1
            print("LN FAIL", (rows, D), e, flush=True)
torch.cuda.synchronize()
print("FUZZ", "FAILED" if bad else "clean", bad, n)
