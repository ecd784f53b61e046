"""Online sliding-window reconstruction (BASELINE config 5): the per-chunk math of slam/online_reconstructor.py
(`Pi3SLAMOnline`, :230) without its process / queue / viser plumbing (out of scope as a component,
its math is the offline path's).  Frames stream through chunks of `chunk_length` with `overlap`; every chunk goes through
the same device path as OfflineChunkCreator._process_single_chunk (optionally replayed as a captured hipGraph) and is
aligned to the previous, already aligned chunk with the closed-form Sim(3) of the overlap views
(_align_chunk_online :1297-1339 -> align_and_refine_reconstructions), in one pass and without the disk round trip of
the offline two-stage flow.  The class keeps the reference's constructor arguments and result accessors that have a
meaning here."""
from __future__ import annotations

import os
import time
from typing import Dict, List, Optional

import numpy as np
import torch

from .alignment import align_and_refine_reconstructions, create_view_graph_matches
from .chunk_creator import OfflineChunkCreator, OfflineCreatorConfig
from .image_io import ChunkImageDataset, calculate_target_size, ingest_frames_device
from .reconstructor import OfflineReconstructor


class Pi3SLAMOnline:
    def __init__(self, model=None, chunk_length: int = 100, overlap: int = 10, device: str = "cuda",
                 conf_threshold: float = 0.5, undistortion_maps=None, cam_scale: float = 1.0,
                 visualization_port: int = 9090, estimate_camera_params: bool = False, keypoint_type: str = "grid",
                 max_num_keypoints: int = 512, keypoint_detection_threshold: float = 0.005,
                 save_chunk_reconstructions: bool = False, max_observations_per_track: int = 5,
                 do_metric_depth: bool = False, save_debug_projections: bool = False, model_path: Optional[str] = None,
                 use_inverse_depth: bool = False, moge_model=None, moge_model_path: Optional[str] = None,
                 hip_graph: bool = True, output_dir: Optional[str] = None):
        self.chunk_length, self.overlap = int(chunk_length), int(overlap)
        self.pixel_limit = 255000 // 2
        self.output_dir = output_dir or os.path.join("/tmp", f"pi3_online_{os.getpid()}")
        cfg = OfflineCreatorConfig(
            model_path=model_path or "recipe", output_dir=self.output_dir, chunk_length=self.chunk_length,
            overlap=self.overlap, device=device, do_metric_depth=do_metric_depth, keypoint_type=keypoint_type,
            max_num_keypoints=max_num_keypoints, keypoint_detection_threshold=keypoint_detection_threshold,
            estimate_camera_params=estimate_camera_params, num_loader_workers=0, pin_memory=False,
            moge_model_path=moge_model_path, device_resize=True, hip_graph=hip_graph)
        self._creator = OfflineChunkCreator(cfg, model=model, moge_model=moge_model)
        self._creator.undistortion_maps = undistortion_maps
        self.device = self._creator.device
        self.model = self._creator.model
        self.max_observations_per_track = max_observations_per_track
        self.chunk_reconstructions: List[Dict] = []     # aligned chunk dicts (the reference keeps pytheia objects)
        self.alignment_infos: List[Optional[Dict]] = []
        self.timestamps: List[int] = []
        self._timing: Dict[str, List[float]] = {}
        self._matches = create_view_graph_matches(self.chunk_length, self.overlap)

    # ------------------------------------------------------------------ timing table (online_reconstructor.py:1096-1119)
    def _record_timing(self, name: str, duration_s: float) -> None:
        self._timing.setdefault(name, []).append(float(duration_s))

    def get_timing_statistics(self) -> Dict[str, Dict[str, float]]:
        return {k: {"count": len(v), "total_s": float(np.sum(v)), "mean_s": float(np.mean(v)), "max_s": float(np.max(v))}
                for k, v in self._timing.items() if v}

    def print_timing_statistics(self) -> None:
        for k, st in self.get_timing_statistics().items():
            print(f"   {k:24s} n={st['count']:4d} mean={st['mean_s'] * 1e3:8.1f} ms total={st['total_s']:.2f} s")

    # ------------------------------------------------------------------ one chunk
    def _process_chunk_with_images(self, chunk_images: torch.Tensor, chunk_paths: List) -> Dict:
        """pi3 forward + masks + metric scale + intrinsics + keypoint gather (online_reconstructor.py:1128-1294), then
        the alignment with the previous chunk."""
        t0 = time.time()
        chunk = self._creator._process_single_chunk(chunk_images, chunk_paths)
        self._record_timing("create_chunk", time.time() - t0)
        t0 = time.time()
        info = self._align_chunk_online(chunk)
        self._record_timing("align_chunk", time.time() - t0)
        n_new = chunk["camera_poses"].shape[0] - (self.overlap if len(self.chunk_reconstructions) > 1 else 0)
        self.timestamps.extend(range(len(self.timestamps), len(self.timestamps) + max(0, int(n_new))))
        return {"chunk": chunk, "transformation": info}

    def _align_chunk_online(self, chunk: Dict) -> np.ndarray:
        self.chunk_reconstructions.append(chunk)
        if len(self.chunk_reconstructions) == 1:
            self.alignment_infos.append(None)
            return np.eye(4)
        ok, info = align_and_refine_reconstructions(self.chunk_reconstructions[-2], chunk, self._matches,
                                                    device=str(self.device))
        self.alignment_infos.append(info if ok else None)
        if not ok:
            print(f"   ❌ Alignment failed for chunk {len(self.chunk_reconstructions) - 1}")
            return np.eye(4)
        return info["sim3_summary"]["matrix"].numpy().astype(np.float64)

    # ------------------------------------------------------------------ whole stream
    def process_chunks(self, image_paths: List[str]) -> List[Dict]:
        """Synchronous form of start_background_loader + process_chunks_with_background_loader (:620-759)."""
        self._creator.target_size = calculate_target_size(image_paths[0], pixel_limit=self.pixel_limit)
        ds = ChunkImageDataset(image_paths, self.chunk_length, self.overlap, self._creator.target_size, decode_only=True)
        results, t_start, frames_before = [], time.time(), len(self.timestamps)
        for idx in range(len(ds)):
            item = ds[idx]
            frames = item["chunk_u8"].to(self.device, non_blocking=True)
            und = self._creator.undistortion_maps
            imgs = (und.undistort_frames_device(frames, self._creator.target_size) if und is not None
                    else ingest_frames_device(frames, self._creator.target_size))[None]
            print(f"\n📦 Processing chunk {idx + 1}/{len(ds)}: frames {int(item['start_idx']) + 1}-{int(item['end_idx'])}")
            kpx = self._creator.keypoint_extractor
            if kpx is not None and hasattr(kpx, "reseed"):
                kpx.reseed(idx)          # same per-chunk keypoint subset as process_and_save
            t0 = time.time()
            results.append(self._process_chunk_with_images(imgs, item["chunk_paths"][0]))
            self._record_timing("process_chunk", time.time() - t0)
        self.print_timing_statistics()
        dt = max(1e-6, time.time() - t_start)
        n = len(self.timestamps) - frames_before
        print(f"\n⏱️ Overall performance: {n} frames in {dt:.2f}s  ->  average {n / dt:.2f} FPS")
        return results

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
