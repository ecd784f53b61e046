import os, sys, torch
sys.path.insert(0, "/root/repo")
from pi3_slam_amd.moge import MoGeEngine
dev = torch.device("cuda:0")
img = torch.rand(3, 308, 406, device=dev)
for dt in (torch.float16, torch.bfloat16):
    eng = MoGeEngine.from_pretrained("recipe", str(dev), dtype=dt)
    for _ in range(3): eng.infer(img)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): eng.infer(img)
    e1.record(); torch.cuda.synchronize()
    print(dt, f"{e0.elapsed_time(e1)/10:.3f} ms per infer (eager, alone)")
