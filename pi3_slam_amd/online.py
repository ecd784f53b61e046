"""Online sliding-window reconstruction (BASELINE config 5): the per-chunk math of slam/online_reconstructor.py
(`Pi3SLAMOnline`, :230) without its spawn / mp.Queue / viser plumbing (out of scope as a
component, its math is the offline path's).

Frames stream through chunks of `chunk_length` with `overlap`.  Every chunk runs through OfflineChunkCreator's
pipelined device path (copy stream for the next chunk's upload + resize, compute stream, optional hipGraph replay of the
forward) and is aligned to its predecessor with the closed-form Sim(3) of the overlap views
(_align_chunk_online :1297-1339 -> align_and_refine_reconstructions) on a separate high-priority stream, so the
alignment of chunk k-1 rides beside the forward of chunk k.  One pass, no disk round trip.

Chunk-parallel (torch.distributed.run, one rank per GPU): chunk c is created on rank c % world; per wave of `world`
chunks the ranks exchange boundary blocks and transforms (dist.WaveAligner: RCCL all-gathers, each rank solves its own
T, prefix composition), every rank moves its chunk into the global frame, and rank 0 collects the chunks.  Chunks can
reach rank 0 in any order; `InOrderDrain` releases them strictly in chunk order, as the reference drains its inference
worker's out-of-order outputs (process_chunks_with_inference_worker / _process_ready_outputs_in_order, :761-920).

The class keeps the reference's constructor arguments and the result accessors that have a meaning here."""
from __future__ import annotations

import os
import time
from typing import Dict, Iterator, List, Optional, Tuple

import numpy as np
import torch

from .alignment import align_and_refine_reconstructions, create_view_graph_matches, transform_chunk
from .chunk_creator import OfflineChunkCreator, OfflineCreatorConfig
from .image_io import ChunkImageDataset, calculate_target_size
from .reconstructor import OfflineReconstructor


class InOrderDrain:
    """Reorder buffer: items arrive tagged with a chunk index in any order, pop_ready() yields the run of consecutive
    indices that starts at the next expected one (online_reconstructor.py:852-920)."""

    def __init__(self, first: int = 0):
        self.next_index = first
        self._pending: Dict[int, object] = {}

    def put(self, index: int, item) -> None:
        if index < self.next_index or index in self._pending:
            raise ValueError(f"chunk {index} was already delivered")
        self._pending[index] = item

    def pop_ready(self) -> Iterator[Tuple[int, object]]:
        while self.next_index in self._pending:
            item = self._pending.pop(self.next_index)
            self.next_index += 1
            yield self.next_index - 1, item

    def __len__(self) -> int:
        return len(self._pending)


class Pi3SLAMOnline:
    def __init__(self, model=None, chunk_length: int = 100, overlap: int = 10, device: str = "cuda",
                 conf_threshold: float = 0.5, undistortion_maps=None, cam_scale: float = 1.0,
                 visualization_port: int = 9090, estimate_camera_params: bool = False, keypoint_type: str = "grid",
                 max_num_keypoints: int = 512, keypoint_detection_threshold: float = 0.005,
                 save_chunk_reconstructions: bool = False, max_observations_per_track: int = 5,
                 do_metric_depth: bool = False, save_debug_projections: bool = False, model_path: Optional[str] = None,
                 use_inverse_depth: bool = False, moge_model=None, moge_model_path: Optional[str] = None,
                 hip_graph: bool = True, output_dir: Optional[str] = None, num_loader_workers: int = 0,
                 bundle_adjust: bool = True, reuse_overlap_encoder: bool = False):
        self.use_inverse_depth = bool(use_inverse_depth)   # online_reconstructor.py:246,1018,1235
        self.chunk_length, self.overlap = int(chunk_length), int(overlap)
        self.pixel_limit = 255000 // 2
        self.output_dir = output_dir or os.path.join("/tmp", f"pi3_online_{os.getpid()}")
        cfg = OfflineCreatorConfig(
            model_path=model_path or "recipe", output_dir=self.output_dir, chunk_length=self.chunk_length,
            overlap=self.overlap, device=device, do_metric_depth=do_metric_depth, keypoint_type=keypoint_type,
            max_num_keypoints=max_num_keypoints, keypoint_detection_threshold=keypoint_detection_threshold,
            estimate_camera_params=estimate_camera_params, num_loader_workers=num_loader_workers,
            pin_memory=num_loader_workers > 0, moge_model_path=moge_model_path, device_resize=True, hip_graph=hip_graph,
            reuse_overlap_encoder=reuse_overlap_encoder)
        self._creator = OfflineChunkCreator(cfg, model=model, moge_model=moge_model)
        self._creator.undistortion_maps = undistortion_maps
        self.rank, self.world = self._creator.rank, self._creator.world
        self.device = self._creator.device
        self.model = self._creator.model
        self.max_observations_per_track = max_observations_per_track
        # the two pytheia refinement stages of the reference (ChunkPTRecon BA, prior-constrained BA after alignment) on
        # the device (bundle_adjust.py); the chunk-parallel branch runs them as a sequential chain over the ranks
        self.bundle_adjust = bool(bundle_adjust)
        self.chunk_reconstructions: List[Dict] = []     # aligned chunk dicts (the reference keeps pytheia objects)
        self.alignment_infos: List[Optional[Dict]] = []
        self.timestamps: List[int] = []
        self._timing: Dict[str, List[float]] = {}
        self._matches = create_view_graph_matches(self.chunk_length, self.overlap)
        self._align_stream = torch.cuda.Stream(self.device, priority=-1)

    # ------------------------------------------------------------------ timing table (online_reconstructor.py:1096-1119)
    def _record_timing(self, name: str, duration_s: float) -> None:
        self._timing.setdefault(name, []).append(float(duration_s))

    def get_timing_statistics(self) -> Dict[str, Dict[str, float]]:
        return {k: {"count": len(v), "total_s": float(np.sum(v)), "mean_s": float(np.mean(v)), "max_s": float(np.max(v))}
                for k, v in self._timing.items() if v}

    def print_timing_statistics(self) -> None:
        for k, st in self.get_timing_statistics().items():
            print(f"   {k:24s} n={st['count']:4d} mean={st['mean_s'] * 1e3:8.1f} ms total={st['total_s']:.2f} s")

    # ------------------------------------------------------------------ one chunk (reference surface)
    def _process_chunk_with_images(self, chunk_images: torch.Tensor, chunk_paths: List) -> Dict:
        """pi3 forward + masks + metric scale + intrinsics + keypoint gather (online_reconstructor.py:1128-1294), then
        the alignment with the previous chunk; blocking form for one chunk."""
        t0 = time.time()
        chunk = self._creator._process_single_chunk(chunk_images, chunk_paths)
        self._record_timing("create_chunk", time.time() - t0)
        return self._consume(chunk)

    def _ba_args(self, chunk: Dict) -> Optional[Dict]:
        if not self.bundle_adjust or chunk.get("keypoints") is None:
            return None
        return {"width": int(chunk.get("original_width", 1920)), "height": int(chunk.get("original_height", 1080)),
                "max_observations_per_track": self.max_observations_per_track,
                "settings": {"inverse_depth": self.use_inverse_depth}}

    def _refine_new_chunk(self, chunk: Dict) -> None:
        args = self._ba_args(chunk)
        if args is None:
            return
        from .bundle_adjust import PER_CHUNK, bundle_adjust_chunk
        t0 = time.time()
        with torch.cuda.stream(self._align_stream):
            bundle_adjust_chunk(chunk, args["width"], args["height"], args["max_observations_per_track"],
                                str(self.device), dict(PER_CHUNK, **args["settings"]))
        self._record_timing("bundle_adjust_chunk", time.time() - t0)

    def _consume(self, chunk: Dict) -> Dict:
        """Sequential consumer: refine the chunk, align it to the previous one, account the new frames."""
        self._refine_new_chunk(chunk)
        t0 = time.time()
        info = self._align_chunk_online(chunk)
        self._record_timing("align_chunk", time.time() - t0)
        self._count_frames(chunk)
        return {"chunk": chunk, "transformation": info}

    def _count_frames(self, chunk: Dict) -> None:
        n_new = chunk["camera_poses"].shape[0] - (self.overlap if len(self.chunk_reconstructions) > 1 else 0)
        self.timestamps.extend(range(len(self.timestamps), len(self.timestamps) + max(0, int(n_new))))

    def _align_chunk_online(self, chunk: Dict) -> np.ndarray:
        self.chunk_reconstructions.append(chunk)
        if len(self.chunk_reconstructions) == 1:
            self.alignment_infos.append(None)
            return np.eye(4)
        with torch.cuda.stream(self._align_stream):     # beside, not behind, the next chunk's forward
            ok, info = align_and_refine_reconstructions(self.chunk_reconstructions[-2], chunk, self._matches,
                                                        use_inverse_depth=self.use_inverse_depth, device=str(self.device),
                                                        bundle_adjust=self._ba_args(chunk))
        self.alignment_infos.append(info if ok else None)
        if not ok:
            print(f"   ❌ Alignment failed for chunk {len(self.chunk_reconstructions) - 1}")
            return np.eye(4)
        return info["sim3_summary"]["global_matrix"].numpy().astype(np.float64)

    # ------------------------------------------------------------------ whole stream
    def _items(self, ds: ChunkImageDataset, indices: List[int]):
        und = self._creator.undistortion_maps
        nw = self._creator.config.num_loader_workers
        if nw > 0:      # decode threads, two chunks ahead, pinned staging (image_io.ThreadedChunkLoader)
            from .image_io import ThreadedChunkLoader
            loader = ThreadedChunkLoader(ds, indices, threads=max(nw, 4), depth=2)
            source = ((indices[i], {k: (v[0] if k != "chunk_paths" else v) for k, v in b.items()})
                      for i, b in enumerate(loader))
        else:
            source = ((i, ds[i]) for i in indices)
        for idx, item in source:
            s, e = int(item["start_idx"]), int(item["end_idx"])
            print(f"\n📦 Processing chunk {idx + 1}/{len(ds)}: frames {s + 1}-{e}")
            yield {"frames": item["chunk_u8"], "kind": "u8_undist" if und is not None else "u8",
                   "paths": item["chunk_paths"][0], "meta": {"chunk_index": idx, "start_idx": s, "end_idx": e}}

    def process_chunks(self, image_paths: List[str]) -> List[Dict]:
        """start_background_loader + process_chunks_with_inference_worker (:620-920) in one process per GPU: the
        stages of consecutive chunks overlap on streams instead of processes."""
        self._creator.target_size = calculate_target_size(image_paths[0], pixel_limit=self.pixel_limit)
        ds = ChunkImageDataset(image_paths, self.chunk_length, self.overlap, self._creator.target_size, decode_only=True)
        import torch.distributed as _dist
        if self.world > 1 or (_dist.is_available() and _dist.is_initialized()):
            return self._process_chunks_distributed(ds)
        results, t_start, frames_before = [], time.time(), len(self.timestamps)
        for meta, chunk in self._creator.process_chunks(self._items(ds, list(range(len(ds))))):
            t0 = time.time()
            results.append(self._consume(chunk))
            self._record_timing("consume_chunk", time.time() - t0)
            self._record_timing("pi3_forward", chunk["_metrics"]["infer_s"])
        self._report(t_start, frames_before)
        return results

    def _process_chunks_distributed(self, ds: ChunkImageDataset) -> List[Dict]:
        """Chunk c on rank c % world; wave by wave: create (pipelined per rank), exchange, transform, collect on rank 0
        in chunk order."""
        import torch.distributed as dist

        from .dist import WaveAligner, chain_payload, chain_step, gather_objects
        rank, world, n = self.rank, self.world, len(ds)
        aligner = WaveAligner(rank, world, self.overlap, self.chunk_length, str(self.device))
        prev_payload: Optional[Dict] = None        # bundle adjustment on: the refined predecessor (dist.chain_step)
        mine = list(range(rank, n, world))
        stream = self._creator.process_chunks(self._items(ds, mine))
        drain = InOrderDrain()
        results, t_start, frames_before = [], time.time(), len(self.timestamps)
        keep = ("points", "colors", "keypoints", "masks", "camera_poses", "image_paths", "intrinsics", "_metrics")
        for w0 in range(0, n, world):
            c = w0 + rank
            chunk = None
            if c < n:
                meta, chunk = next(stream)
                assert meta["chunk_index"] == c
            if chunk is not None:
                self._refine_new_chunk(chunk)
            if self.bundle_adjust:
                # the reference's refinement is a sequential chain (align to the REFINED predecessor, then adjust with
                # its poses as priors): the ranks of a wave take turns, the refined chunk travels on; the next wave's
                # forward is already queued on this rank's compute stream and runs meanwhile
                my_ok, my_G = True, torch.eye(4, dtype=torch.float64)
                for r in range(min(world, n - w0)):
                    pay = None
                    if r == rank:
                        if w0 + r > 0:
                            with torch.cuda.stream(self._align_stream):
                                my_ok, info = align_and_refine_reconstructions(
                                    prev_payload, chunk, self._matches, device=str(self.device),
                                    use_inverse_depth=self.use_inverse_depth, bundle_adjust=self._ba_args(chunk))
                            if my_ok:
                                my_G = info["sim3_summary"]["global_matrix"]
                        pay = chain_payload(chunk)
                    prev_payload = chain_step(pay, r)
                oks, Gs = {rank: my_ok}, {rank: my_G}
            else:
                with torch.cuda.stream(self._align_stream):
                    Gs, oks = aligner.step(chunk, w0, n)
                    if chunk is not None:
                        transform_chunk(chunk, Gs[rank], device=str(self.device), absolute=True)
            payload = None if chunk is None else (c, {k: chunk[k] for k in keep if k in chunk}, bool(oks[rank]),
                                                  Gs[rank])
            parts = gather_objects(payload)
            if rank == 0:
                for part in reversed(parts):                  # arrival order is not chunk order: the drain restores it
                    if part is not None:
                        drain.put(part[0], part)
                for idx, (_, ch, ok, G) in drain.pop_ready():
                    self.chunk_reconstructions.append(ch)
                    self.alignment_infos.append({"success": ok, "global_matrix": G} if idx > 0 else None)
                    if not ok:
                        print(f"   ❌ Alignment failed for chunk {idx}: it stays in its own frame")
                    self._count_frames(ch)
                    results.append({"chunk": ch, "transformation": G.numpy()})
        dist.barrier()
        if rank == 0:
            assert len(drain) == 0 and drain.next_index == n
            self._report(t_start, frames_before)
        return results

    def _report(self, t_start: float, frames_before: int) -> None:
        self.print_timing_statistics()
        dt = max(1e-6, time.time() - t_start)
        n = len(self.timestamps) - frames_before
        print(f"\n⏱️ Overall performance: {n} frames in {dt:.2f}s  ->  average {n / dt:.2f} FPS")

    # ------------------------------------------------------------------ accessors / export
    def get_chunk_reconstructions(self):
        return self.chunk_reconstructions

    def get_latest_reconstruction(self):
        return self.chunk_reconstructions[-1] if self.chunk_reconstructions else None

    def get_reconstruction_count(self):
        return len(self.chunk_reconstructions)

    def get_statistics(self) -> Dict:
        return {"num_chunks": len(self.chunk_reconstructions), "num_frames": len(self.timestamps),
                "timing": self.get_timing_statistics()}

    def _exporter(self) -> OfflineReconstructor:
        rec = OfflineReconstructor.__new__(OfflineReconstructor)
        rec.reconstructions, rec.output_dir = self.chunk_reconstructions, self.output_dir
        return rec

    def save_trajectory_tum(self, save_path: str, timestamps: Optional[List[float]] = None,
                            integer_timestamp: bool = False) -> None:
        self._exporter()._save_trajectory_tum(save_path, integer_timestamp=integer_timestamp)

    def save_final_result(self, save_path: str, max_points: int = 1000000) -> None:
        from .reconstructor import write_ply
        pts, cols = self._exporter()._extract_points_colors()
        if pts.shape[0] > max_points:
            sel = np.random.default_rng(0).choice(pts.shape[0], max_points, replace=False)
            pts, cols = pts[sel], (cols[sel] if cols.size else cols)
        write_ply(pts, cols if cols.size else np.ones_like(pts), save_path)
