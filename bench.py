"""Headline benchmark of the hot path (BASELINE.json): frames/s end-to-end for chunk creation + overlap alignment at
chunk_length=100, overlap=20, 4:3 input -> 308x406 (what calculate_target_size makes of a 512x384 frame).

    python bench.py [--gpus N] [--steps K] [--warmup W]        (N > 1: launched by torch.distributed.run, one rank/GPU)

One "step" = one chunk of 100 synthetic frames already resident in HBM: pi3 forward -> masks -> MoGe metric scale (if a
MoGe engine is available) -> per-frame intrinsics (LM) -> grid keypoints (K=200) gather + fp16 pack -> D2H of the packed
chunk -> overlap Sim(3) alignment against the previous chunk (+ for N > 1 an RCCL all-gather of the boundary blocks and
the prefix composition).  Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CL, OV, H, W, KP = 100, 20, 308, 406, 200
PEAK_BF16_DENSE_TFLOPS = 2500.0   # MI355X dense bf16 MFMA peak (MI355X_MICROARCH.md)
# HBM-side bytes of ONE global-attention launch from rocprofv3 PMC passes (profiles/r01c_attention_pmc.csv):
# FETCH_SIZE 643 097 KB (x2: gfx950 reports half of a wide coalesced read stream) + WRITE_SIZE 128 601 KB.
# Algorithmic bytes are 527 MB (q, k, v read once + o written); L2 hit rate 95.8 % on the K/V re-reads.
ATTN_TRAFFIC_BYTES = (2 * 643096.75 + 128600.9) * 1024.0


def cpu_baseline(engine_cfg, n_frames: int = 4):
    """The oracle (CPU restatement of the reference forward, fp32 eager) timed on this box's host cores on a bounded
    sample of the same workload.  Weights are produced on the device by the same recipe and copied over (values do
    not matter for timing; this avoids a minute of numpy)."""
    from oracle import pi3_ref
    from pi3_slam_amd.weights import param_shapes, recipe_fill_device
    sd = {}
    for name, shape in param_shapes(engine_cfg).items():
        sd[name] = recipe_fill_device(name, shape, "cuda:0").cpu()
    imgs = torch.rand(1, n_frames, 3, H, W)
    threads = min(torch.get_num_threads(), len(os.sched_getaffinity(0)), 16)  # the GPU box grants 16 cores per GPU
    torch.set_num_threads(threads)
    t0 = time.perf_counter()
    pi3_ref.pi3_forward(sd, imgs, engine_cfg)
    dt = time.perf_counter() - t0
    return {"value": n_frames / dt, "unit": "frames/s", "cores": threads, "kind": "port",
            "sample": f"oracle pi3 forward only (no post-processing), {n_frames} frames {H}x{W}, fp32 eager torch CPU, "
                      f"{dt:.1f} s; global attention is quadratic in the frame count, so the CPU rate at 100 frames "
                      f"is lower still (BASELINE.md)"}


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with --nproc-per-node {args.gpus}")
    ndev = torch.cuda.device_count()
    backend = os.environ.get("PI3_DIST_BACKEND", "nccl")   # "gloo" only to rehearse the N > 1 logic on a 1-GPU box
    dev = torch.device(f"cuda:{local_rank % max(1, ndev) if backend == 'gloo' else local_rank}")
    torch.cuda.set_device(dev)
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)   # RCCL over xGMI
        else:
            dist.init_process_group(backend)
        # build the communicator now (untimed set-up, like weight initialisation): the first collective of a process
        # group pays the RCCL ring / xGMI topology discovery, which must not land in a timed step when --warmup is 0
        _w = torch.zeros(1, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(_w)
        dist.all_gather([torch.empty_like(_w) for _ in range(world)], _w)

    from pi3_slam_amd import ops
    from pi3_slam_amd.alignment import create_view_graph_matches, estimate_sim3
    from pi3_slam_amd.chunk_creator import OfflineChunkCreator, OfflineCreatorConfig
    from pi3_slam_amd.dist import (allgather_boundaries, compose_global, pack_boundary,
                                   relative_sim3_from_boundaries, unpack_boundary)
    from pi3_slam_amd.engine import Pi3Engine
    from pi3_slam_amd.weights import Pi3Config

    cfg = Pi3Config()
    engine = Pi3Engine(cfg, str(dev))
    moge = None
    try:
        from pi3_slam_amd.moge import MoGeEngine
        moge = MoGeEngine.from_pretrained("recipe", str(dev))
    except Exception as e:  # noqa: BLE001
        if rank == 0:
            print(f"[bench] MoGe engine unavailable ({e}); metric scaling is NOT in the timed region", file=sys.stderr)
    cc = OfflineCreatorConfig(model_path="recipe", output_dir="/tmp/pi3_bench_out", chunk_length=CL, overlap=OV,
                              device=str(dev), do_metric_depth=moge is not None, keypoint_type="grid",
                              max_num_keypoints=KP, num_loader_workers=0)
    creator = OfflineChunkCreator(cc, model=engine, moge_model=moge)
    creator.target_size = (H, W)

    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    frames = torch.rand(1, CL, 3, H, W, device=dev, generator=g)      # synthetic, resident in HBM
    paths = [[f"frame_{i:06d}.png"] for i in range(CL)]
    matches = create_view_graph_matches(CL, OV)
    attn_events = []
    prev = None

    def step(timed: bool):
        nonlocal prev
        # route the events through the creator's model call
        if timed:
            orig = creator.model.forward
            creator.model = _EventedModel(engine, attn_events)
        chunk = creator._process_single_chunk(frames, paths)
        if timed:
            creator.model = engine
        if world == 1:
            if prev is not None:
                out = estimate_sim3(prev, chunk, matches, str(dev))
                ops.sim3_apply(out[13:29].contiguous(), chunk["points"].to(dev, torch.float32).contiguous(),
                               chunk["camera_poses"].to(dev).contiguous())
        else:
            blocks = [unpack_boundary(b.cpu(), OV, KP)
                      for b in allgather_boundaries(pack_boundary(chunk, OV, KP), dev if backend == "nccl" else "cpu")]
            rel = [torch.eye(4, dtype=torch.float64, device=dev).reshape(16)]
            for r in range(1, world):
                rel.append(relative_sim3_from_boundaries(blocks[r - 1], blocks[r], OV, dev, chunk_length=CL)[13:29])
            compose_global(torch.stack(rel))
        prev = chunk

    class _EventedModel:
        def __init__(self, eng, events):
            self.eng, self.events = eng, events

        def __call__(self, imgs):
            return self.eng.forward(imgs, global_attn_events=self.events)

    for _ in range(args.warmup):
        step(False)

    def sync_all():
        torch.cuda.synchronize(dev)
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
            torch.cuda.synchronize(dev)

    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    sync_all()
    dt = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist
        tt = torch.tensor([dt], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    if rank == 0:
        T = (H // 14) * (W // 14) + cfg.n_dec_reg
        S = CL * T
        attn_flops = 4.0 * cfg.heads * float(S) * float(S) * 64.0          # one global-attention launch
        attn_ms = sum(a.elapsed_time(b) for a, b in attn_events) / max(1, len(attn_events))
        achieved = attn_flops / (attn_ms * 1e-3) / 1e12
        fl = engine.flops(1, CL, H, W)
        line = {
            "metric": "frames/sec end-to-end (chunk create+align), 512x384 cl=100 ov=20",
            "value": world * CL * args.steps / dt,
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "bf16",
            "data": "synthetic",
            "config": {"workload": "configs[1]-shaped synthetic chunk: 100 frames 308x406 (512x384 after "
                                   "calculate_target_size), cl=100 ov=20, grid keypoints K=200, recipe weights",
                       "chunks_per_step_per_gpu": 1, "parallelism": f"chunk-parallel x{world}",
                       "moge_metric_scale_in_timed_region": moge is not None,
                       "algorithmic_tflop_per_chunk": fl["total"] / 1e12},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": PEAK_BF16_DENSE_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_BF16_DENSE_TFLOPS, "traffic": ATTN_TRAFFIC_BYTES,
                         "kernel": "attn_fwd64_kernel<8, true, true> + key-norm pre-pass (global attention, S=64300, 16 heads, d=64)",
                         "launch_ms": attn_ms, "launches_timed": len(attn_events),
                         "end_to_end_tflops": fl["total"] * args.steps / dt / 1e12},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(cfg)
        _REAL_STDOUT.write(json.dumps(line) + "\n")
        _REAL_STDOUT.flush()
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    # stdout carries exactly ONE line, the JSON record of rank 0; the pipeline's progress prints (per-chunk inference
    # FPS etc., on every rank) go to stderr
    _REAL_STDOUT = sys.stdout
    sys.stdout = sys.stderr
    try:
        main()
    finally:
        sys.stdout = _REAL_STDOUT
