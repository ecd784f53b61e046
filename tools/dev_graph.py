"""hipGraph capture of the per-chunk forward: equality with the eager run and timing (full model, 100 frames)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pi3_slam_amd.engine import Pi3Engine
from pi3_slam_amd.weights import Pi3Config
dev = "cuda:0"
eng = Pi3Engine(Pi3Config(), dev)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
g = torch.Generator(device=dev).manual_seed(0)
a = torch.rand(1, N, 3, 308, 406, device=dev, generator=g)
b = torch.rand(1, N, 3, 308, 406, device=dev, generator=g)
ref_a = {k: v.clone() for k, v in eng.forward(a).items()}
ref_b = {k: v.clone() for k, v in eng.forward(b).items()}
out = eng.forward_graphed(a)
torch.cuda.synchronize()
print("capture ok; a equal:", all(torch.equal(out[k], ref_a[k]) for k in ref_a))
out = eng.forward_graphed(b)
torch.cuda.synchronize()
print("replay  b equal:", all(torch.equal(out[k], ref_b[k]) for k in ref_b))
for name, fn in (("eager", lambda: eng.forward(a)), ("graph", lambda: eng.forward_graphed(a))):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter(); c0 = time.process_time()
    for _ in range(4): fn()
    t_launch = time.perf_counter() - t0
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{name}: {dt / 4 * 1e3:.1f} ms per chunk, host launch time {t_launch / 4 * 1e3:.1f} ms, cpu {(time.process_time() - c0) / 4 * 1e3:.1f} ms")
