"""Per-chunk forward: eager launches against the captured hipGraph (config 5), alone on the device."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pi3_slam_amd.engine import Pi3Engine
from pi3_slam_amd.weights import Pi3Config

dev = torch.device("cuda:0")
eng = Pi3Engine(Pi3Config(), str(dev))
imgs = torch.rand(1, 100, 3, 308, 406, device=dev)


def timeit(fn, n=5):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n, t_host / n * 1e3


for rep in range(2):
    g, hg = timeit(lambda: eng.forward(imgs))
    print(f"eager   : {g:8.2f} ms GPU per chunk, {hg:7.2f} ms host to queue")
    g, hg = timeit(lambda: eng.forward_graphed(imgs))
    print(f"graphed : {g:8.2f} ms GPU per chunk, {hg:7.2f} ms host to queue")
