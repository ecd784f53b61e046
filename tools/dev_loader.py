"""Debug: loader throughput on the GPU box (fork of a GPU process, pin thread, decode rate)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from PIL import Image
from torch.utils.data import DataLoader
from bench import synthetic_frames_u8
from pi3_slam_amd.image_io import ChunkImageDataset
torch.zeros(1, device="cuda:0")
print("cpus", len(os.sched_getaffinity(0)), os.cpu_count())
d = "/tmp/dev_loader_frames"; os.makedirs(d, exist_ok=True)
fr = synthetic_frames_u8(100, 384, 512, 5).numpy()
files = []
t0 = time.time()
for i in range(660):
    p = f"{d}/frame_{i:05d}.png"
    if not os.path.exists(p):
        if i < 100: Image.fromarray(fr[i]).save(p, compress_level=1)
        else: os.link(files[i % 100], p)
    files.append(p)
print("write", round(time.time() - t0, 2))
t0 = time.time()
for i in range(20): np.array(Image.open(files[i]).convert("RGB"), dtype=np.uint8)
print("decode ms/frame in main", (time.time() - t0) / 20 * 1e3)
ds = ChunkImageDataset(files, 100, 20, (308, 406), decode_only=True)
for nw, pin in ((8, False), (8, True), (4, True)):
    t0 = time.time()
    loader = DataLoader(ds, batch_size=1, shuffle=False, num_workers=nw, pin_memory=pin, persistent_workers=True, prefetch_factor=2)
    it = iter(loader)
    print(nw, pin, "iter created", round(time.time() - t0, 2))
    for i, b in enumerate(it):
        print(nw, pin, "item", i, round(time.time() - t0, 2))
    del it, loader
