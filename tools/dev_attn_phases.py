"""Per-workgroup phases of the frame-wise attention (100 x 643 tokens, 16 heads): prologue (Q fragments + tile 0 staged),
key sweep, store tail, in s_memrealtime ticks.  Needs the stamped development build of the library:
    cd pi3_slam_amd/csrc && hipcc ... -DPI3_ATTN_STAMPS -c attn64.hip ... -> libpi3slam_hip_stamps.so   (tools/README.md)
    PI3_LIB_PATH=pi3_slam_amd/libpi3slam_hip_stamps.so python tools/dev_attn_phases.py 2> phases.log
The library prints `PHASES wg ...` lines for a sample of workgroups on its fourth launch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pi3_slam_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
B, S, H = 100, 643, 16
qkv = torch.randn(B * S, 3 * H * 64, device=dev); qkv[:, :H * 64] *= ops.QSCALE * 2; qkv = qkv.bfloat16()
out = torch.empty(B * S, H * 64, device=dev, dtype=torch.bfloat16)
k = qkv.view(B, S, 3, H, 64)[:, :, 1].float()
k2 = (k * k).sum(-1).amax(1).reshape(-1).contiguous()
for _ in range(6):
    ops.attention(qkv, out, B, S, H, k2max=k2)
torch.cuda.synchronize()
print("done")
