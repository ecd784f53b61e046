"""Interleaved A/B timing of the global-attention launch (S = 64 300, 16 heads, bounded-score path) across several builds
of the library loaded side by side in ONE process (cards of the pool differ by +-4 %, so only same-process rounds compare).

    python tools/dev_attn_ab.py name=path.so[:knob=v,...] [name=path.so ...] [--rounds 6] [--online-max] [--k2-ready] [--outlier-key]

Timing-only ablation builds (-DA64_ABL=n) compute wrong results by construction: never the product library."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

args = [a for a in sys.argv[1:] if "=" in a]
rounds = int(sys.argv[sys.argv.index("--rounds") + 1]) if "--rounds" in sys.argv else 6
online = "--online-max" in sys.argv
S, H = 64300, 16
dev = torch.device("cuda:0")
torch.manual_seed(0)
qkv = (torch.randn(S, 3 * H * 64, device=dev) * 0.5).bfloat16()
if "--outlier-key" in sys.argv:       # one key of every head with 8 x the norm: |q| max|k| leaves the a-priori bound, the scores stay moderate
    qkv.view(S, 3, H, 64)[1234, 1] *= 8.0
out = torch.empty(S, H * 64, device=dev, dtype=torch.bfloat16)
k2 = torch.zeros(H, device=dev)
k2_ready = int("--k2-ready" in sys.argv)       # max |k|^2 per head handed over as the fused qkv epilogue does (no pre-pass in the timing)
if k2_ready:
    kk = qkv.float().view(S, 3, H, 64)[:, 1]
    k2 = (kk * kk).sum(-1).amax(dim=0).contiguous()
libs, knobs = {}, {}
for a in args:
    name, path = a.split("=", 1)
    path, _, kn = path.partition(":")          # name=path.so[:knob=value,knob=value]: knobs set before every launch group
    knobs[name] = [(k.split("=")[0].encode(), int(k.split("=")[1])) for k in kn.split(",") if k]
    lib = ctypes.CDLL(os.path.abspath(path))
    vp, l, i = ctypes.c_void_p, ctypes.c_long, ctypes.c_int
    lib.pi3_attention.argtypes = [vp, vp, vp, l, l, vp, l, l, i, i, i, i, i, vp, i, vp]
    lib.pi3_set_knob.argtypes = [ctypes.c_char_p, l]
    if online:
        lib.pi3_set_knob(b"attn_nomax", 0)
    libs[name] = lib


def setk(name):
    for k, v in knobs[name]:
        libs[name].pi3_set_knob(k, v)


def launch(lib):
    base, ts = qkv.data_ptr(), qkv.stride(0)
    rc = lib.pi3_attention(base, base + 2 * H * 64, base + 4 * H * 64, ts, S * ts, out.data_ptr(), out.stride(0),
                           S * out.stride(0), 1, S, H, 64, 0, k2.data_ptr(), k2_ready, torch.cuda.current_stream().cuda_stream)
    assert rc == 0, rc


times = {n: [] for n in libs}
for n, lib in libs.items():
    setk(n)
    for _ in range(2):
        launch(lib)
torch.cuda.synchronize()
for r in range(rounds):
    for n, lib in libs.items():
        setk(n)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            launch(lib)
        e1.record()
        torch.cuda.synchronize()
        times[n].append(e0.elapsed_time(e1) / 3)
base = None
for n, t in times.items():
    t = sorted(t)
    med = t[len(t) // 2]
    base = base or med
    print(f"{n:14s} median {med:7.3f} ms  min {t[0]:7.3f}  max {t[-1]:7.3f}   {med / base:6.3f} x first   "
          f"{4.0 * H * S * S * 64 / med / 1e9:7.1f} TF/s-equivalent")

if "--check" in sys.argv:       # every variant against the fp32 softmax on a strided sample of rows
    import math
    rows = torch.arange(0, S, 997, device=dev)
    x = qkv.float().view(S, 3, H, 64)
    q, k, v = x[rows, 0].transpose(0, 1), x[:, 1].transpose(0, 1), x[:, 2].transpose(0, 1)
    ref = (torch.softmax(q @ k.transpose(-1, -2) * math.log(2.0), -1) @ v).transpose(0, 1).reshape(len(rows), H * 64)
    for n, lib in libs.items():
        setk(n)
        out.zero_()
        launch(lib)
        torch.cuda.synchronize()
        err = (out[rows].float() - ref).abs()
        print(f"check {n:14s} max|d| {err.max().item():.3e} mean|d| {err.mean().item():.3e} (ref max {ref.abs().max().item():.3f})")

if "--paths" in sys.argv:       # which softmax loop the waves' results came from (pi3_attention_path_counters), one launch per variant
    cnt = torch.zeros(2, 2, 32, device=dev, dtype=torch.int32)
    for n, lib in libs.items():
        setk(n)
        cnt.zero_()
        lib.pi3_attention_path_counters.argtypes = [ctypes.c_void_p]
        lib.pi3_attention_path_counters(cnt.data_ptr())
        launch(lib)
        torch.cuda.synchronize()
        lib.pi3_attention_path_counters(None)
        w = cnt.sum(-1)[0].tolist()
        print(f"paths {n:14s} bounded-score waves {w[0]}  online-max waves {w[1]} (= {w[1] // 8} workgroups of {(S + 511) // 512 * H})")
