"""Race screen for a sync-structure edit of gemm256 (cdna guide, §5: 'screen it for races over many runs at several
sizes'): every shape is run many times on fresh random operands and each result must be bit-identical to the first run
of the same operands, and within bf16 tolerance of an fp32 reference."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pi3_slam_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(1)
bad = 0
for (M, N, K, reps) in [(64300, 3072, 1024, 30), (64300, 1024, 4096, 20), (64300, 4096, 1024, 20), (5000, 256, 4096, 100),
                        (2049, 1024, 1024, 100), (1500, 512, 128, 200), (1024, 256, 64, 200), (70000, 256, 192, 50)]:
    for trial in range(3):
        a = torch.randn(M, K, device=dev).bfloat16()
        w = (torch.randn(N, K, device=dev) / math.sqrt(K)).bfloat16()
        bias = torch.randn(N, device=dev)
        first = None
        for r in range(reps):
            out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
            ops.gemm(a, w, out, bias=bias, act=ops.ACT_GELU if trial == 1 else 0)
            if first is None:
                first = out
                if M <= 5000:
                    ref = a.float() @ w.float().T + bias
                    if trial == 1:
                        ref = torch.nn.functional.gelu(ref)
                    err = ((out.float() - ref).abs().max() / ref.abs().max()).item()
                    if err > 6e-3:
                        bad += 1
                        print("TOLERANCE", (M, N, K), err)
            elif not torch.equal(out, first):
                bad += 1
                print("MISMATCH", (M, N, K), "trial", trial, "rep", r, (out.float() - first.float()).abs().max().item())
                break
        # f32 output with residual (proj / fc2 form)
        x0 = torch.randn(M, N, device=dev)
        gamma = torch.rand(N, device=dev)
        first = None
        for r in range(max(3, reps // 4)):
            x = x0.clone()
            ops.gemm(a, w, x, bias=bias, gamma=gamma, resid=x)
            if first is None:
                first = x
            elif not torch.equal(x, first):
                bad += 1
                print("MISMATCH f32", (M, N, K), "trial", trial, "rep", r)
                break
    print("shape", (M, N, K), "done", flush=True)
print("RACE SCREEN", "FAILED" if bad else "clean", bad)
