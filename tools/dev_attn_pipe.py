"""A/B of the software-pipelined attention kernel (PI3_ATTN_PIPE=1) against the three-phase one: correctness on small
and ragged shapes (both bounded-score and online-max waves) + timing at the north-star shape.  One process per variant
(the knob is read once)."""
import math, os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    import torch
    from pi3_slam_amd import ops
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    def attn_ref(qkv, B, S, H):
        q, k, v = qkv.double().view(B, S, 3, H, 64).permute(2, 0, 3, 1, 4)
        p = torch.softmax((q @ k.transpose(-1, -2)) * math.log(2.0), dim=-1)
        return (p @ v).permute(0, 2, 1, 3).reshape(B * S, H * 64)
    for (B, S, H, scale) in [(1, 4096, 2, 0.5), (1, 4096 + 44, 2, 0.5), (2, 4500, 1, 0.5), (1, 4160, 2, 4.0), (1, 5000, 1, 0.5)]:
        qkv = torch.randn(B * S, 3 * H * 64, device=dev) * scale
        if S == 5000:      # one spiked key row forces the online-max path and a rescale
            qkv[3333, H * 64:2 * H * 64] *= 30
        qkv[:, :H * 64] *= ops.QSCALE * 2
        qkv = qkv.bfloat16()
        out = torch.empty(B * S, H * 64, device=dev, dtype=torch.bfloat16)
        ops.attention(qkv, out, B, S, H)
        r = attn_ref(qkv, B, S, H)
        print(f"check {(B, S, H, scale)} max rel err {((out.double() - r).abs().max() / r.abs().max()).item():.2e}", flush=True)
    B, S, H = 1, 64300, 16
    qkv = (torch.randn(B * S, 3 * H * 64, device=dev) * 0.5).bfloat16()
    out = torch.empty(B * S, H * 64, device=dev, dtype=torch.bfloat16)
    for _ in range(3): ops.attention(qkv, out, B, S, H)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 10
    e0.record()
    for _ in range(n): ops.attention(qkv, out, B, S, H)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print(f"PIPE={os.environ.get('PI3_ATTN_PIPE','0')} attn S={S}: {ms:.3f} ms  {4.0*B*H*S*S*64/ms/1e9:.1f} TF/s", flush=True)
else:
    for pipe in ("0", "1", "0", "1"):
        env = dict(os.environ, PI3_ATTN_PIPE=pipe)
        subprocess.run([sys.executable, __file__, "run"], env=env, check=False, timeout=300)
