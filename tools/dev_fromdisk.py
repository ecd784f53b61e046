"""Debug: where the wall time of process_and_save() from PNG files goes."""
import os, sys, time, shutil, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from PIL import Image
from bench import synthetic_frames_u8, CL, OV, KP, SRC_H, SRC_W
from pi3_slam_amd.chunk_creator import OfflineChunkCreator, OfflineCreatorConfig
from pi3_slam_amd.engine import Pi3Engine
from pi3_slam_amd.moge import MoGeEngine
from pi3_slam_amd.weights import Pi3Config
dev = "cuda:0"
engine, moge = Pi3Engine(Pi3Config(), dev), MoGeEngine.from_pretrained("recipe", dev)
tmp = tempfile.mkdtemp(prefix="pi3_disk_")
fr = synthetic_frames_u8(100, SRC_H, SRC_W, 5).numpy()
files = []
for i in range(660):
    p = os.path.join(tmp, f"frame_{i:05d}.png")
    if i < 100: Image.fromarray(fr[i]).save(p, compress_level=1)
    else: os.link(files[i % 100], p)
    files.append(p)
os.environ["PI3_TRACE"] = "1"
for workers, pin, ov in ((8, True, True),):
    cc = OfflineCreatorConfig(model_path="recipe", output_dir=os.path.join(tmp, f"out{workers}{pin}{ov}"), chunk_length=CL, overlap=OV,
                              device=dev, do_metric_depth=True, keypoint_type="grid", max_num_keypoints=KP,
                              num_loader_workers=workers, pin_memory=pin, device_resize=True, overlap_stages=ov)
    cr = OfflineChunkCreator(cc, model=engine, moge_model=moge)
    cr.process_and_save(files[:120])
    t0 = time.time()
    cr.process_and_save(files)
    print(f"### workers={workers} pin={pin} overlap={ov}: {time.time() - t0:.2f} s  {cr.last_run}", flush=True)
shutil.rmtree(tmp, ignore_errors=True)
