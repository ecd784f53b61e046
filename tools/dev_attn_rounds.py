"""Does the global attention pay for a partial last round of workgroups?  S = 57 344 is 7.00 rounds of 256 workgroups
(112 query blocks x 16 heads), 64 300 is 7.875 (run as 8), 65 536 is 8.00: time per (rounds x key tiles) against time
per S^2 tells whether splitting the keys of the last round's workgroups could recover anything."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pi3_slam_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
H = 16
res = {}
for rnd in range(3):
    for S in (57344, 64300, 65536, 61440):
        qkv = (torch.randn(S, 3 * H * 64, device=dev) * 0.5).bfloat16()
        out = torch.empty(S, H * 64, device=dev, dtype=torch.bfloat16)
        kk = qkv.view(1, S, 3, H, 64)[:, :, 1].float()
        k2 = (kk * kk).sum(-1).amax(1).reshape(-1).contiguous()
        for _ in range(2):
            ops.attention(qkv, out, 1, S, H, k2max=k2)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(6):
            ops.attention(qkv, out, 1, S, H, k2max=k2)
        e1.record()
        torch.cuda.synchronize()
        res.setdefault(S, []).append(e0.elapsed_time(e1) / 6)
for S, v in res.items():
    t = sorted(v)[len(v) // 2]
    nqb, nt = (S + 511) // 512, (S + 63) // 64
    rounds = nqb * H / 256
    print(f"S={S}: {t:.3f} ms  workgroups {nqb * H} = {rounds:.3f} rounds, {nt} key tiles;  us per (ceil-round x tile) "
          f"{1e3 * t / (-(-nqb * H // 256) * nt):.4f}   us per (exact-round x tile) {1e3 * t / (rounds * nt):.4f}")
