"""Timing / sanity of the bundle adjustment at the north-star chunk size (100 views, 200 keypoints per view)."""
import os, sys, time, tempfile, shutil
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pi3_slam_amd.chunk_creator import OfflineChunkCreator, OfflineCreatorConfig
from pi3_slam_amd.engine import Pi3Engine
from pi3_slam_amd.weights import Pi3Config
from pi3_slam_amd.bundle_adjust import bundle_adjust_chunk, PER_CHUNK, AFTER_ALIGNMENT
dev = "cuda:0"
small = Pi3Config(dim=128, enc_depth=1, dec_depth=2, head_depth=1, cam_dim=128, pos_grid=5)
eng = Pi3Engine(small if len(sys.argv) < 2 else Pi3Config(), dev)
cfg = OfflineCreatorConfig(model_path="recipe", output_dir="/tmp/pi3_ba_full", chunk_length=100, overlap=20,
                           do_metric_depth=False, keypoint_type="grid", max_num_keypoints=200, num_loader_workers=0)
cr = OfflineChunkCreator(cfg, model=eng, moge_model=None)
cr.target_size = (308, 406)
imgs = torch.rand(1, 100, 3, 308, 406, generator=torch.Generator().manual_seed(0))
chunk = cr._process_single_chunk(imgs, [[f"f{i}.png"] for i in range(100)])
for name, settings in (("per-chunk (10 it, Huber 2)", PER_CHUNK), ("after-alignment settings (50 it, Huber 3)", AFTER_ALIGNMENT)):
    c = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in chunk.items()}
    torch.cuda.synchronize(); t0 = time.time()
    info = bundle_adjust_chunk(c, 406, 308, 5, dev, settings)
    torch.cuda.synchronize(); dt = time.time() - t0
    uv, valid, _ = c["_observations"]
    print(f"{name}: {dt*1e3:.1f} ms  observations {int(valid.sum())}  {info}")
    print("   pose change max", (c["camera_poses"] - chunk["camera_poses"]).abs().max().item(),
          " point change max", (c["points"].float() - chunk["points"].float()).abs().max().item())
