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
n = {"gemm": 0, "attn": 0, "ln": 0, "conv": 0, "masks": 0, "cast": 0}

def rel(a, b):
    return ((a.float() - b.float()).abs().max() / (b.float().abs().max() + 1e-12)).item()

def attn_ref(qkv, B, S, H):
    q, k, v = qkv.float().view(B, S, 3, H, 64).permute(2, 0, 3, 1, 4)
    p = torch.softmax((q @ k.transpose(-1, -2)) * math.log(2.0), dim=-1)
    return (p @ v).permute(0, 2, 1, 3).reshape(B * S, H * 64)

t_end = time.time() + budget
while time.time() < t_end:
    kind = rng.choice(["gemm", "gemm", "attn", "attn", "ln", "conv", "masks", "cast"])
    n[kind] += 1
    if kind == "gemm":
        M = rng.choice([rng.randint(1, 300), rng.randint(1000, 1300), rng.randint(1, 9000), 1024, 2048, 2049, 255, 257])
        N = rng.choice([128, 256, 384, 512, 640, 768, 1024, 3072, 32, 64, 96, 160, 224])     # the last five: narrow-N kernel
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
    elif kind == "conv":       # 3x3 replicate-padded convolution (narrow and wide forms) against torch on the CPU
        H, W = rng.randint(1, 40), rng.randint(1, 40)
        Ci, Co = rng.choice([3, 20, 32, 33, 64, 96, 128]), rng.choice([1, 3, 32, 40, 64, 96, 128, 256])
        Cp = 32 if Ci <= 32 else (Ci + 63) // 64 * 64
        Np = (Co + 31) // 32 * 32
        x = torch.randn(H * W, Ci)
        w4 = torch.randn(Co, Ci, 3, 3) / (3.0 * Ci ** 0.5)
        img = torch.zeros(H * W, Cp)
        img[:, :Ci] = x
        img = img.bfloat16()
        if Cp == 32:
            rows_w = torch.zeros(Np, 10, 32)
            rows_w[:Co, :9, :Ci] = w4.permute(0, 2, 3, 1).reshape(Co, 9, Ci)
        else:
            rows_w = torch.zeros(Np, 3, 3, Cp)
            rows_w[:Co, :, :, :Ci] = w4.permute(0, 2, 3, 1)
        rows_w = rows_w.reshape(Np, -1).bfloat16().contiguous()
        bias = torch.zeros(Np)
        bias[:Co] = torch.randn(Co)
        out = torch.full((H * W, Np), float("nan"), device=dev)
        ops.conv3x3(img.to(dev), H, W, Cp, rows_w.to(dev), bias.to(dev), out)
        xr = img.float()[:, :Ci].reshape(1, H, W, Ci).permute(0, 3, 1, 2)
        ref = torch.nn.functional.conv2d(torch.nn.functional.pad(xr, (1, 1, 1, 1), mode="replicate"), w4.bfloat16().float(),
                                         bias[:Co])[0].permute(1, 2, 0).reshape(H * W, Co)
        e = rel(out[:, :Co].cpu(), ref)
        if not (e < 3e-5) or not bool((out[:, Co:] == 0).all()):
            bad += 1
            print("CONV FAIL", (H, W, Ci, Co), e, flush=True)
    elif kind == "masks":      # fast decisions == exact arithmetic (csrc/post.hip) on random shapes and thresholds
        F, H, W = rng.randint(1, 5), rng.randint(1, 70), rng.choice([1, 2, 3, 5, 64, 341, 342, 406, 448, 1024, rng.randint(1, 1024)])
        lp = torch.randn(F, H, W, 3, device=dev)
        z = torch.exp(0.3 * torch.randn(F, H, W, device=dev)) if rng.random() < 0.5 else \
            1.0 + 0.015 * torch.arange(W, device=dev).view(1, 1, W) + 0.015 * torch.arange(H, device=dev).view(1, H, 1) + 1e-4 * torch.randn(F, H, W, device=dev)
        if rng.random() < 0.3:
            r = torch.rand(F, H, W, device=dev)
            z = torch.where(r < 0.01, torch.full_like(z, float("nan")), z)
            z = torch.where((r > 0.02) & (r < 0.03), torch.zeros_like(z), z)
            z = torch.where((r > 0.03) & (r < 0.04), torch.full_like(z, float("inf")), z)
        lp[..., 2] = z
        conf = (torch.randn(F, H, W, 1, device=dev) * 3).contiguous()
        thr, rtol = rng.choice([(0.1, 0.03), (0.5, 0.0), (0.9, 1.0), (0.01, 0.001)])
        os.environ.pop("PI3_MASKS_EXACT_ONLY", None)
        a = ops.compute_masks(conf, lp.contiguous(), thr, rtol)
        os.environ["PI3_MASKS_EXACT_ONLY"] = "1"
        b = ops.compute_masks(conf, lp.contiguous(), thr, rtol)
        os.environ.pop("PI3_MASKS_EXACT_ONLY", None)
        if not torch.equal(a, b):
            bad += 1
            print("MASKS FAIL", (F, H, W), thr, rtol, int((a != b).sum()), flush=True)
    elif kind == "cast":
        rows, cols = rng.randint(1, 5000), 4 * rng.randint(1, 600)
        in_cols = 4 * rng.randint(1, cols // 4)
        x = torch.randn(rows, in_cols + 4 * rng.randint(0, 3), device=dev)
        out = torch.full((rows, cols + 8), float("nan"), device=dev, dtype=torch.bfloat16)
        ops.cast_rows(x, out, cols=cols, in_cols=in_cols)
        ok = torch.equal(out[:, :in_cols], x[:, :in_cols].bfloat16()) and bool((out[:, in_cols:cols] == 0).all()) \
            and bool(out[:, cols:].float().isnan().all())
        if not ok:
            bad += 1
            print("CAST FAIL", (rows, cols, in_cols), flush=True)
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
