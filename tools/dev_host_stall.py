"""Where a one-off host stall of the consumer side comes from (one 92 ms sample among 1-2 ms ones in
tests/test_pipeline_gpu.py::test_consumer_side_alignment_does_not_wait_for_the_running_forward; consume_ms_max 186 ms in
the driver's online-stream extra).  Times every candidate call made while a forward is running on the compute stream -
pinned-memory allocation (torch.empty(pin_memory=True) / Tensor.pin_memory), host<->device copies (Tensor.to / .cpu),
event / stream / device synchronisation, stream creation - and prints every call above 3 ms with its caller.

    python tools/dev_host_stall.py [chunks=60] [frames=60]
"""
import os
import sys
import time
import traceback

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from pi3_slam_amd.alignment import align_and_refine_reconstructions, create_view_graph_matches  # noqa: E402
from pi3_slam_amd.chunk_creator import OfflineChunkCreator, OfflineCreatorConfig  # noqa: E402
from pi3_slam_amd.engine import Pi3Engine  # noqa: E402
from pi3_slam_amd.weights import Pi3Config  # noqa: E402

N_CHUNKS = int(sys.argv[1]) if len(sys.argv) > 1 else 60
N = int(sys.argv[2]) if len(sys.argv) > 2 else 60
THRESH_MS = 3.0
log, phase = [], ["setup"]


def timed(fn, name):
    def w(*a, **k):
        t0 = time.perf_counter()
        r = fn(*a, **k)
        dt = (time.perf_counter() - t0) * 1e3
        if dt > THRESH_MS and phase[0] != "setup":
            st = traceback.extract_stack(limit=5)[:-1]
            where = " <- ".join(f"{f.filename.split('/')[-1]}:{f.lineno}" for f in reversed(st))
            what = ""
            if a and torch.is_tensor(a[0]):
                t = a[0]
                what = f" {tuple(t.shape)} {t.dtype} {t.device} pinned={t.is_pinned() if t.device.type == 'cpu' else '-'}"
            log.append((phase[0], name, dt, what, where))
        return r
    return w


_empty = torch.empty


def empty(*a, **k):
    if k.get("pin_memory"):
        return timed(_empty, "empty(pin_memory=True)")(*a, **k)
    return _empty(*a, **k)


torch.empty = empty
torch.Tensor.pin_memory = timed(torch.Tensor.pin_memory, "Tensor.pin_memory")
torch.Tensor.to = timed(torch.Tensor.to, "Tensor.to")
torch.Tensor.cpu = timed(torch.Tensor.cpu, "Tensor.cpu")
torch.Tensor.item = timed(torch.Tensor.item, "Tensor.item")
torch.cuda.Event.synchronize = timed(torch.cuda.Event.synchronize, "Event.synchronize")
torch.cuda.Stream.synchronize = timed(torch.cuda.Stream.synchronize, "Stream.synchronize")
torch.cuda.synchronize = timed(torch.cuda.synchronize, "cuda.synchronize")
torch.cuda.Stream.__new__ = timed(torch.cuda.Stream.__new__, "Stream()")

# garbage-collector pauses (a full collection walks every container object of the process)
import gc  # noqa: E402
import threading  # noqa: E402

_gc_t0 = [0.0]


def _gc_cb(ph, info):
    if ph == "start":
        _gc_t0[0] = time.perf_counter()
    else:
        dt = (time.perf_counter() - _gc_t0[0]) * 1e3
        if dt > THRESH_MS and phase[0] != "setup":
            log.append((phase[0], f"gc gen{info['generation']} ({info['collected']} collected)", dt, "", "garbage collector"))


gc.callbacks.append(_gc_cb)
# 2 ms stack sampler of the main thread during the alignment calls: what an unwrapped call was doing in a slow one
samples, main_id = [], threading.get_ident()


def _sampler():
    while True:
        time.sleep(0.002)
        if phase[0].startswith("align"):
            fr = sys._current_frames().get(main_id)
            if fr is not None:
                st = traceback.extract_stack(fr, limit=6)
                samples.append((phase[0], time.perf_counter(),
                                " <- ".join(f"{f.filename.split('/')[-1]}:{f.lineno}" for f in reversed(st))))


threading.Thread(target=_sampler, daemon=True).start()

dev = "cuda:0"
eng = Pi3Engine(Pi3Config(), dev)
cc = OfflineCreatorConfig(model_path="recipe", output_dir="/tmp/pi3_dev_stall", chunk_length=N, overlap=N // 5, device=dev,
                          do_metric_depth=False, keypoint_type="grid", max_num_keypoints=200, device_resize=True)
cr = OfflineChunkCreator(cc, model=eng, moge_model=None)
cr.target_size = (308, 406)
frames = torch.randint(0, 256, (N, 384, 512, 3), dtype=torch.uint8).pin_memory()
paths = [[f"f{i}.png"] for i in range(N)]
matches = create_view_graph_matches(N, N // 5)
side = torch.cuda.Stream(dev, priority=-1)
items = ({"frames": frames, "kind": "u8", "paths": paths, "meta": {"chunk_index": i}} for i in range(N_CHUNKS))
prev, waits, gaps = None, [], []
t_last = time.perf_counter()
it = cr.process_chunks(items)
i = 0
while True:
    phase[0] = f"creator[{i}]" if i >= 2 else "setup"
    t0 = time.perf_counter()
    try:
        meta, chunk = next(it)
    except StopIteration:
        break
    gaps.append((time.perf_counter() - t0) * 1e3)
    if prev is not None:
        phase[0] = f"align[{i}]" if i >= 2 else "setup"
        t0 = time.perf_counter()
        with torch.cuda.stream(side):
            ok, _ = align_and_refine_reconstructions(prev, chunk, matches, device=dev)
        waits.append((time.perf_counter() - t0) * 1e3)
    prev = chunk
    i += 1
torch.cuda.synchronize()
fwd = 1e3 * chunk["_metrics"]["infer_s"]
w = sorted(waits[2:])
print(f"forward {fwd:.1f} ms; align host wait ms: median {w[len(w) // 2]:.2f} p90 {w[int(0.9 * len(w))]:.2f} max {w[-1]:.2f}; "
      f"samples > 10 ms: {[round(x, 1) for x in waits[2:] if x > 10]}")
print(f"creator yield-to-yield host ms: median {sorted(gaps[2:])[len(gaps[2:]) // 2]:.1f} max {max(gaps[2:]):.1f}")
by = {}
for ph, name, dt, what, where in log:
    by.setdefault((ph.split('[')[0], name, where), []).append(dt)
print(f"calls above {THRESH_MS} ms while a forward was in flight (phase, call, count, total ms, max ms, caller chain):")
for (ph, name, where), v in sorted(by.items(), key=lambda kv: -max(kv[1])):
    print(f"  {ph:8s} {name:24s} n={len(v):3d} total {sum(v):8.1f} max {max(v):7.1f}   {where}")
for ph, name, dt, what, where in log:
    if ph.startswith("align") and dt > 10:
        print(f"  STALL {ph} {name} {dt:.1f} ms{what} at {where}")
slow = {f"align[{k + 1}]" for k, x in enumerate(waits) if x > 10 and k >= 1}
for ph in sorted(slow):
    rows = [(t, w_) for p_, t, w_ in samples if p_ == ph]
    print(f"  stack samples of {ph} ({len(rows)} x 2 ms):")
    last, n = None, 0
    for t, w_ in rows + [(0, None)]:
        if w_ != last:
            if last is not None:
                print(f"     {n * 2:4d} ms  {last}")
            last, n = w_, 0
        n += 1
