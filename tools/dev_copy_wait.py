"""Host wait of small transfers on a high-priority stream while a chunk forward runs on the main stream."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pi3_slam_amd.engine import Pi3Engine
from pi3_slam_amd.weights import Pi3Config

dev = torch.device("cuda:0")
eng = Pi3Engine(Pi3Config(), str(dev))
imgs = torch.rand(1, 100, 3, 308, 406, device=dev)
eng.forward(imgs); torch.cuda.synchronize()
side = torch.cuda.Stream(dev, priority=-1)
host = torch.randn(20, 200, 3).half()
pinned = torch.empty(20, 200, 3, dtype=torch.float16).pin_memory()


def t(fn):
    t0 = time.perf_counter(); r = fn(); return (time.perf_counter() - t0) * 1e3, r


for rep in range(3):
    eng.forward(imgs)                       # ~410 ms of queued GPU work on the main stream
    time.sleep(0.05)
    with torch.cuda.stream(side):
        a, d1 = t(lambda: host.to(dev))                                   # pageable H2D
        b, d2 = t(lambda: pinned.to(dev, non_blocking=True))              # pinned, async
        c, d3 = t(lambda: d2.float() * 2)                                 # a kernel
        d, h1 = t(lambda: d3.cpu())                                       # pageable D2H (waits for the kernel)
        hp = torch.empty(20, 200, 3, pin_memory=True)
        e, _ = t(lambda: (hp.copy_(d3, non_blocking=True), side.synchronize()))   # pinned D2H + stream wait
        f, d4 = t(lambda: host.to(dev))                                   # pageable H2D again
    torch.cuda.synchronize()
    print(f"pageable H2D {a:7.2f} ms | pinned H2D {b:6.2f} | kernel {c:6.2f} | pageable D2H {d:7.2f} | pinned D2H+sync {e:7.2f} | pageable H2D {f:7.2f}")
