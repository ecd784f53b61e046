"""Where the host waits while it aligns chunk c beside the forward of chunk c+1 (cProfile of the consumer side)."""
import cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pi3_slam_amd.alignment import align_and_refine_reconstructions, create_view_graph_matches
from pi3_slam_amd.chunk_creator import OfflineChunkCreator, OfflineCreatorConfig
from pi3_slam_amd.engine import Pi3Engine
from pi3_slam_amd.moge import MoGeEngine
from pi3_slam_amd.weights import Pi3Config

dev = "cuda:0"
eng = Pi3Engine(Pi3Config(), dev)
moge = MoGeEngine.from_pretrained("recipe", dev)
cc = OfflineCreatorConfig(model_path="recipe", output_dir="/tmp/pi3_dev_align", chunk_length=100, overlap=20, device=dev,
                          do_metric_depth=True, keypoint_type="grid", max_num_keypoints=200, device_resize=True)
cr = OfflineChunkCreator(cc, model=eng, moge_model=moge)
cr.target_size = (308, 406)
frames = torch.randint(0, 256, (100, 384, 512, 3), dtype=torch.uint8).pin_memory()
paths = [[f"f{i}.png"] for i in range(100)]
matches = create_view_graph_matches(100, 20)
align_stream = torch.cuda.Stream(dev, priority=-1)
prof = cProfile.Profile()
import traceback
_to, _cpu = torch.Tensor.to, torch.Tensor.cpu
def _wrap(fn, name):
    def w(self, *a, **k):
        t0 = time.perf_counter()
        r = fn(self, *a, **k)
        dt = (time.perf_counter() - t0) * 1e3
        if dt > 5:
            fr = traceback.extract_stack(limit=3)[0]
            print(f"SLOW {name} {dt:.1f} ms  {tuple(self.shape)} {self.dtype} {self.device} -> {tuple(r.shape)} {r.dtype} {r.device} pinned={self.is_pinned() if self.device.type == 'cpu' else '-'} at {fr.filename.split('/')[-1]}:{fr.lineno}", file=sys.stderr)
        return r
    return w
torch.Tensor.to = _wrap(_to, "to")
torch.Tensor.cpu = _wrap(_cpu, "cpu")
import collections, threading
samples = collections.Counter()
in_align = [False]
main_id = threading.get_ident()
def sampler():
    while True:
        time.sleep(0.02)
        if in_align[0]:
            fr = sys._current_frames().get(main_id)
            st = traceback.extract_stack(fr, limit=4)
            samples[" <- ".join(f"{f.filename.split('/')[-1]}:{f.lineno}" for f in reversed(st))] += 1
threading.Thread(target=sampler, daemon=True).start()
prev = None
waits = []
items = ({"frames": frames, "kind": "u8", "paths": paths, "meta": {"chunk_index": i}} for i in range(16))
for meta, chunk in cr.process_chunks(items):
    if prev is not None:
        t0 = time.perf_counter()
        if False:
            prof.enable()
        in_align[0] = True
        with torch.cuda.stream(align_stream):
            ok, _ = align_and_refine_reconstructions(prev, chunk, matches, device=dev)
        in_align[0] = False
        prof.disable()
        waits.append((time.perf_counter() - t0) * 1e3)
    prev = chunk
torch.cuda.synchronize()
print("align host ms per chunk:", [f"{w:.1f}" for w in waits], file=sys.stderr)
for k, v in samples.most_common(8):
    print(f"SAMPLE {v * 20} ms  {k}", file=sys.stderr)
s = io.StringIO()
pstats.Stats(prof, stream=s).sort_stats("cumulative").print_stats(35)
print(s.getvalue(), file=sys.stderr)
