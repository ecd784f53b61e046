"""Race screen for the long-sequence attention kernel (asm LDS-DMA, counted waits, one barrier per tile): repeated
launches on the same operands must agree bit for bit, on both softmax paths, at the north-star size and at ragged ones."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pi3_slam_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(3)
bad = 0
for (B, S, H, scale, reps) in [(1, 64300, 16, 0.5, 12), (1, 64300, 16, 2.5, 6), (1, 8191, 4, 0.5, 100), (1, 4097, 2, 2.5, 100),
                               (2, 5000, 3, 1.0, 60), (100, 643, 16, 0.5, 40)]:
    qkv = torch.randn(B * S, 3 * H * 64, device=dev)
    qkv[:, :2 * H * 64] *= scale
    qkv = qkv.bfloat16()
    first = None
    for r in range(reps):
        out = torch.empty(B * S, H * 64, device=dev, dtype=torch.bfloat16)
        ops.attention(qkv, out, B, S, H)
        if first is None:
            first = out
        elif not torch.equal(out, first):
            bad += 1
            print("MISMATCH", (B, S, H, scale), "rep", r, (out.float() - first.float()).abs().max().item())
            break
    print("case", (B, S, H, scale), "done", flush=True)
print("ATTENTION RACE SCREEN", "FAILED" if bad else "clean", bad)
