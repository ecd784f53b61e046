"""Online sliding-window stream (config 5 shape) on one GPU: hipGraph replay against eager launches, same files."""
import os, shutil, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from PIL import Image
from pi3_slam_amd.engine import Pi3Engine
from pi3_slam_amd.moge import MoGeEngine
from pi3_slam_amd.online import Pi3SLAMOnline
from pi3_slam_amd.weights import Pi3Config

import collections, threading, traceback
samples = collections.Counter()
main_id = threading.get_ident()
def sampler():
    while True:
        time.sleep(0.02)
        fr = sys._current_frames().get(main_id)
        if fr is None:
            continue
        st = traceback.extract_stack(fr, limit=12)
        names = [f.name for f in st]
        if "_consume" in names:
            i = names.index("_consume")
            samples[" <- ".join(f"{f.filename.split('/')[-1]}:{f.lineno}" for f in reversed(st[i:]))] += 1
threading.Thread(target=sampler, daemon=True).start()
n_frames = int(sys.argv[1]) if len(sys.argv) > 1 else 1200
dev = "cuda:0"
eng = Pi3Engine(Pi3Config(), dev)
moge = MoGeEngine.from_pretrained("recipe", dev)
tmp = tempfile.mkdtemp(prefix="pi3_online_dev_")
try:
    rng = np.random.default_rng(0)
    files = []
    for i in range(n_frames):
        p = os.path.join(tmp, f"f_{i:05d}.png")
        if i < 100:
            Image.fromarray(rng.integers(0, 256, (384, 512, 3), dtype=np.uint8)).save(p, compress_level=1)
        else:
            os.link(files[i % 100], p)
        files.append(p)
    for graph in ((os.environ.get("PI3_DEV_GRAPH", "1") == "1"),):
        slam = Pi3SLAMOnline(model=eng, chunk_length=100, overlap=20, device=dev, keypoint_type="grid",
                             max_num_keypoints=200, do_metric_depth=True, moge_model=moge, hip_graph=graph,
                             output_dir=os.path.join(tmp, "out"), bundle_adjust=False, num_loader_workers=8)
        slam.process_chunks(files[:180])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = slam.process_chunks(files)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        st = slam.get_timing_statistics()
        gaps = [r["chunk"]["_metrics"].get("gap_before_forward_s", 0.0) * 1e3 for r in res]
        print("GAPS ms " + " ".join(f"{g:.1f}" for g in gaps), file=sys.stderr)
        print(f"RESULT hip_graph={graph}: {n_frames / dt:.1f} input frames/s, {len(res)} chunks, {dt / len(res) * 1e3:.1f} ms/chunk, "
              f"forward mean {st['pi3_forward']['mean_s'] * 1e3:.1f} ms, consume mean {st['consume_chunk']['mean_s'] * 1e3:.1f} ms",
              file=sys.stderr)
    for k, v in samples.most_common(10):
        print(f"SAMPLE {v * 20} ms  {k}", file=sys.stderr)
finally:
    shutil.rmtree(tmp, ignore_errors=True)
