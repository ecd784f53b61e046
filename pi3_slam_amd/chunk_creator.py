"""OfflineChunkCreator: drop-in mirror of slam/offline_chunk_creator.py (same config dataclass, same methods, same
chunk_*.pt / chunks_manifest.json / chunk_metadata.json layout), with every arithmetic step on the MI355X:

    frames (1,N,3,H,W) --H2D--> Pi3Engine.forward --> masks --> MoGe metric scale --> intrinsics (LM per frame)
                                --> grid keypoints --> gather + fp16 pack --> D2H of N*K*~20 B --> torch.save

The reference copies the four dense maps (350 MB at N=100) to the host and samples them there
(offline_chunk_creator.py:204-213, 228); here only the packed per-keypoint tensors cross PCIe.
Error conventions are the reference's: model / keypoint / intrinsics / save failures are printed and the run goes on
with the feature disabled (offline_chunk_creator.py:77-79, 92-94, 199-201, 242-243, 330-331).
"""
from __future__ import annotations

import json
import os
import time
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import torch
from torch.utils.data import DataLoader

from . import ops
from .engine import Pi3Engine
from .image_io import ChunkImageDataset, calculate_target_size, ingest_frames_device
from .undistortion import create_undistortion_maps
from .keypoints import create_keypoint_extractor
from .weights import Pi3Config


@dataclass
class OfflineCreatorConfig:
    """Same fields and defaults as slam/offline_chunk_creator.py:36-50, plus the knobs the reference hard-codes."""
    model_path: str
    output_dir: str
    chunk_length: int = 100
    overlap: int = 10
    device: str = "cuda"
    do_metric_depth: bool = True
    keypoint_type: str = "aliked"  # 'aliked' | 'grid' | 'none'
    max_num_keypoints: int = 512
    keypoint_detection_threshold: float = 0.005
    estimate_camera_params: bool = True
    num_loader_workers: int = 2
    pin_memory: bool = True
    cam_dist_path: Optional[str] = None
    # --- additions (the reference hard-codes "Ruicheng/moge-2-vits-normal" and an unseeded device randperm)
    moge_model_path: Optional[str] = None   # local MoGe-2 model.pt, or "recipe" for synthetic weights
    keypoint_seed: Optional[int] = 0
    device_resize: bool = False             # loader workers only decode; Resize + ToTensor run on the GPU (bit-identical)
    hip_graph: bool = False                 # replay the per-chunk pi3 forward as one captured hipGraph per chunk shape


def _uv_tables(H: int, W: int, device) -> Tuple[torch.Tensor, torch.Tensor]:
    """normalized_view_plane_uv (utils/geometry_torch.py:39-51) as two 1-D fp32 tables (constants per image size)."""
    ar = W / H
    sx = ar / (1 + ar ** 2) ** 0.5
    sy = 1 / (1 + ar ** 2) ** 0.5
    u = torch.linspace(-sx * (W - 1) / W, sx * (W - 1) / W, W, dtype=torch.float32)
    v = torch.linspace(-sy * (H - 1) / H, sy * (H - 1) / H, H, dtype=torch.float32)
    return u.to(device), v.to(device)


class OfflineChunkCreator:
    """Create per-chunk pi3 results (with MoGe scaling and keypoints) and save them to disk."""

    def __init__(self, config: OfflineCreatorConfig, model: Optional[Pi3Engine] = None, moge_model=None):
        self.config = config
        os.makedirs(self.config.output_dir, exist_ok=True)
        self.chunks_dir = os.path.join(self.config.output_dir, "chunks")
        os.makedirs(self.chunks_dir, exist_ok=True)
        from .dist import ensure_process_group, local_device_index
        self.rank, self.world = ensure_process_group()          # (0, 1) unless launched under torch.distributed.run
        dev = self.config.device if self.config.device != "cuda" else f"cuda:{local_device_index()}"
        if not str(dev).startswith("cuda"):
            raise RuntimeError("this build runs the hot path on an MI355X only; there is no CPU path (device='cuda')")
        self.device = torch.device(dev)

        # pi3 (offline_chunk_creator.py:65): a local checkpoint directory/file, or "recipe" for synthetic weights
        if model is not None:
            self.model = model
        elif self.config.model_path == "recipe":
            self.model = Pi3Engine(Pi3Config(), str(self.device))
        else:
            self.model = Pi3Engine.from_pretrained(self.config.model_path, str(self.device))

        # MoGe (offline_chunk_creator.py:70-79): failure disables metric scaling, it does not abort
        self.moge_model = moge_model
        if self.moge_model is None and self.config.do_metric_depth:
            try:
                from .moge import MoGeEngine
                if not self.config.moge_model_path:
                    raise FileNotFoundError("no local MoGe-2 checkpoint configured (moge_model_path)")
                self.moge_model = MoGeEngine.from_pretrained(self.config.moge_model_path, str(self.device))
                print("   MoGe loaded for metric scaling")
            except Exception as e:  # noqa: BLE001 - same degrade-don't-crash policy as the reference
                print(f"⚠️  Failed to initialize MoGe: {e}. Continuing without metric depth.")
                self.moge_model = None

        self.keypoint_extractor = None
        if self.config.keypoint_type and self.config.keypoint_type.lower() != "none":
            try:
                self.keypoint_extractor = create_keypoint_extractor(
                    keypoint_type=self.config.keypoint_type, max_num_keypoints=self.config.max_num_keypoints,
                    detection_threshold=self.config.keypoint_detection_threshold, device=str(self.device),
                    seed=self.config.keypoint_seed)
                print(f"   Keypoint extractor: {self.config.keypoint_type}")
            except Exception as e:  # noqa: BLE001
                print(f"⚠️  Failed to initialize keypoint extractor: {e}. Continuing without keypoints.")
                self.keypoint_extractor = None

        self.target_size: Optional[Tuple[int, int]] = None
        # Undistortion maps (optional), offline_chunk_creator.py:98-112: built and applied on the device
        self.undistortion_maps = None
        if getattr(self.config, "cam_dist_path", None):
            try:
                if os.path.exists(self.config.cam_dist_path):
                    print(f"🔧 Creating undistortion maps from: {self.config.cam_dist_path}")
                    self.undistortion_maps = create_undistortion_maps(self.config.cam_dist_path, str(self.device))
                    if self.undistortion_maps is not None:
                        print("✅ Undistortion maps ready; images will be undistorted before Pi3 inference")
                    else:
                        print("⚠️  Failed to create undistortion maps; proceeding without undistortion")
                else:
                    print(f"⚠️  Calibration file not found: {self.config.cam_dist_path}")
            except Exception as e:  # noqa: BLE001
                print(f"⚠️  Undistortion map creation failed: {e}")

    # ------------------------------------------------------------------ device steps (same names as the reference)
    @staticmethod
    def _compute_masks(pi3_result: Dict[str, torch.Tensor]) -> torch.Tensor:
        """(B, N, H, W) bool — offline_chunk_creator.py:114-119."""
        conf, lp = pi3_result["conf"], pi3_result["local_points"]
        B, N, H, W = lp.shape[:4]
        m = ops.compute_masks(conf.reshape(B * N, H, W, 1).contiguous(), lp.reshape(B * N, H, W, 3).contiguous())
        return m.view(B, N, H, W).bool()

    @staticmethod
    def _get_scale_factor_for_pi3(moge_metric_depth: torch.Tensor, pi3_metric_depth: torch.Tensor,
                                  mask: torch.Tensor) -> torch.Tensor:
        """0-dim device tensor — offline_chunk_creator.py:121-127.  pi3_metric_depth may be a strided view of
        local_points[..., 2]."""
        assert pi3_metric_depth.stride(-1) in (1, 3)
        n = moge_metric_depth.numel()
        m8 = mask.contiguous().view(torch.uint8) if mask.dtype == torch.bool else mask.contiguous()
        out = ops.masked_ratio_median(moge_metric_depth.contiguous(), pi3_metric_depth, pi3_metric_depth.stride(-1),
                                      m8, n)
        return out[0]

    @staticmethod
    def _interpolate_world_points_for_keypoints(result_dense: Dict, keypoints: torch.Tensor) -> Dict[str, torch.Tensor]:
        """offline_chunk_creator.py:129-159 (+ the fp16 pack of :231-241), on the device."""
        dev = result_dense["points"].device
        masks = result_dense["masks"]
        m8 = masks.contiguous().view(torch.uint8) if masks.dtype == torch.bool else masks.contiguous()
        images = result_dense.get("images")
        return ops.gather_keypoints(result_dense["points"].contiguous(), result_dense["local_points"].contiguous(),
                                    result_dense["conf"].contiguous(), m8,
                                    images.contiguous() if images is not None else None,
                                    keypoints.to(dev, torch.float32).contiguous())

    def _estimate_camera_parameters(self, pi3_result: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        """estimate_camera_parameters_single_chunk (utils/camera_estimation.py:101-132) on the device."""
        lp = pi3_result["local_points"][0].contiguous()
        conf = pi3_result["conf"][0].contiguous()
        H, W = lp.shape[1:3]
        uvx, uvy = _uv_tables(H, W, lp.device)
        r = ops.focal_shift(lp, conf, uvx, uvy)
        fxy = r["fxfycxcy"]
        return dict(intrinsics=r["intrinsics"], focal=r["focal"][None], shift=r["shift"][None], fx=fxy[:, 0][None],
                    fy=fxy[:, 1][None], cx=fxy[:, 2][None], cy=fxy[:, 3][None])

    # ------------------------------------------------------------------ one chunk
    def _process_single_chunk(self, chunk_images: torch.Tensor, chunk_paths: List[str]) -> Dict:
        """Run pi3, optional MoGe scaling, intrinsics, keypoints and the gather for one chunk
        (offline_chunk_creator.py:161-256).  chunk_images: (1, N, 3, H, W) fp32 in [0, 1], CPU or device."""
        assert chunk_images.ndim == 5, "Expected (B=1, N, C, H, W) tensor for chunk images"
        num_frames = int(chunk_images.shape[1])
        t0 = time.time()
        imgs_dev = chunk_images.to(self.device, non_blocking=True)
        if self.config.hip_graph and hasattr(self.model, "forward_graphed"):
            pi3_result = self.model.forward_graphed(imgs_dev)     # static outputs: consumed before the next chunk
        else:
            pi3_result = self.model(imgs_dev)
        torch.cuda.synchronize(self.device)
        dt_inf = max(1e-6, time.time() - t0)
        fps = num_frames / dt_inf if num_frames > 0 else 0.0
        print(f"   ⏱️ Inference: {dt_inf:.3f}s for {num_frames} frames  ->  {fps:.2f} FPS")
        _metrics = {"infer_s": float(dt_inf), "num_frames": int(num_frames), "fps": float(fps)}

        masks = self._compute_masks(pi3_result)[0]

        if self.moge_model is not None:
            infer = (self.moge_model.infer_graphed if self.config.hip_graph and hasattr(self.moge_model, "infer_graphed")
                     else self.moge_model.infer)
            moge_depth = infer(imgs_dev[0, 0])["depth"]
            pi3_depth = pi3_result["local_points"][0, 0][..., 2]
            scale = self._get_scale_factor_for_pi3(moge_depth, pi3_depth, masks[0])
            ops.apply_scale(scale.reshape(1), pi3_result["local_points"], pi3_result["points"],
                            pi3_result["camera_poses"])

        camera_params = None
        if self.config.estimate_camera_params:
            try:
                camera_params = self._estimate_camera_parameters(pi3_result)
            except Exception as e:  # noqa: BLE001
                print(f"⚠️  Camera parameter estimation failed: {e}")
                camera_params = None

        result: Dict = {"camera_poses": pi3_result["camera_poses"][0].cpu(), "image_paths": chunk_paths,
                        "_metrics": _metrics}
        if camera_params is not None:
            result["camera_params"] = {k: v.cpu() for k, v in camera_params.items()}
        if self.target_size is not None:
            result["original_width"] = self.target_size[1]
            result["original_height"] = self.target_size[0]

        done = False
        if self.keypoint_extractor is not None:
            try:
                kp_res = self.keypoint_extractor.extract(chunk_images)
                dense = dict(points=pi3_result["points"][0], local_points=pi3_result["local_points"][0],
                             conf=pi3_result["conf"][0], masks=masks, images=imgs_dev[0])
                interp = self._interpolate_world_points_for_keypoints(dense, kp_res["keypoints"])
                result["points"] = interp["points"].cpu()
                result["local_points"] = interp["local_points"].cpu()
                result["conf"] = interp["conf"].cpu()
                result["masks"] = interp["masks"].cpu()
                result["keypoints"] = interp["keypoints"].cpu()
                result["descriptors"] = kp_res["descriptors"].to(torch.float16)
                result["scores"] = kp_res["scores"].to(torch.float16)
                result["colors"] = interp["colors"].cpu()
                done = True
            except Exception as e:  # noqa: BLE001
                print(f"⚠️  Keypoint extraction failed: {e}")
        if not done:  # dense maps are stored instead (offline_chunk_creator.py:204-209)
            result["points"] = pi3_result["points"][0].cpu()
            result["local_points"] = pi3_result["local_points"][0].cpu()
            result["conf"] = pi3_result["conf"][0].cpu()
            result["masks"] = masks.cpu()
        if "camera_params" in result and result["camera_params"] is not None:
            result["intrinsics"] = result["camera_params"].get("intrinsics", None)
        return result

    # ------------------------------------------------------------------ whole sequence
    def process_and_save(self, image_paths: List[str]) -> List[str]:
        """offline_chunk_creator.py:258-371."""
        if not image_paths:
            raise ValueError("image_paths is empty")
        self.target_size = calculate_target_size(image_paths[0], pixel_limit=255000 // 2)
        print(f"Target size: {self.target_size}")
        undist = self.undistortion_maps
        dataset = ChunkImageDataset(image_paths, self.config.chunk_length, self.config.overlap, self.target_size,
                                    decode_only=self.config.device_resize or undist is not None)
        nw = self.config.num_loader_workers
        # chunk-parallel over the GPUs of the node: chunk c is created by rank c % world; chunks do not
        # depend on each other (each takes its metric scale from its own first frame)
        my_chunks = list(range(self.rank, len(dataset), self.world))
        shard = dataset if self.world == 1 else torch.utils.data.Subset(dataset, my_chunks)
        loader = DataLoader(shard, batch_size=1, shuffle=False, num_workers=nw, pin_memory=self.config.pin_memory,
                            persistent_workers=nw > 0 and len(my_chunks) > 0, prefetch_factor=1 if nw > 0 else None)
        saved_files: List[str] = []
        manifest: List[Dict] = []
        print(f"🔄 Processing {len(my_chunks)} of {len(dataset)} chunks (rank {self.rank}/{self.world})...")
        infer_times, infer_frames, per_chunk_fps = [], [], []
        for local_idx, batch in enumerate(loader):
            chunk_idx = my_chunks[local_idx]
            if self.keypoint_extractor is not None and hasattr(self.keypoint_extractor, "reseed"):
                self.keypoint_extractor.reseed(chunk_idx)     # the random grid subset does not depend on the sharding
            start_idx = int(batch["start_idx"].item())
            end_idx = int(batch["end_idx"].item())
            if undist is not None:     # remap + ToTensor on the GPU (datasets/image_datasets.py:192-199)
                frames = batch["chunk_u8"][0].to(self.device, non_blocking=True)
                chunk_images = undist.undistort_frames_device(frames, self.target_size)[None]
            elif self.config.device_resize:
                frames = batch["chunk_u8"][0].to(self.device, non_blocking=True)
                chunk_images = ingest_frames_device(frames, self.target_size)[None]
            else:
                chunk_images = batch["chunk"]
            chunk_paths = batch["chunk_paths"][0]
            print(f"📦 Chunk {chunk_idx + 1}/{len(dataset)}: frames {start_idx + 1}-{end_idx}")
            chunk_result = self._process_single_chunk(chunk_images, chunk_paths)
            m = chunk_result.get("_metrics", {})
            if m:
                infer_times.append(float(m.get("infer_s", 0.0)))
                infer_frames.append(int(m.get("num_frames", 0)))
                per_chunk_fps.append(float(m.get("fps", 0.0)))
            out_name = f"chunk_{chunk_idx:06d}.pt"
            out_path = os.path.join(self.chunks_dir, out_name)
            chunk_result["chunk_index"] = chunk_idx
            chunk_result["start_idx"] = start_idx
            chunk_result["end_idx"] = end_idx
            try:
                torch.save(chunk_result, out_path)
                saved_files.append(out_path)
                manifest.append({"chunk_index": chunk_idx, "file": out_name, "start_idx": start_idx,
                                 "end_idx": end_idx, "num_frames": len(chunk_paths),
                                 "image_paths": chunk_paths})
                print(f"   💾 Saved: {out_path}")
            except Exception as e:  # noqa: BLE001
                print(f"❌ Failed to save chunk {chunk_idx}: {e}")
        try:
            total_time, total_frames = sum(infer_times), sum(infer_frames)
            overall = (total_frames / total_time) if total_time > 0 else 0.0
            steady = sorted(f for f, n in zip(per_chunk_fps, infer_frames) if n == self.config.chunk_length)
            print(f"\n⏱️ Overall inference: {total_frames} frames in {total_time:.3f}s  ->  {overall:.2f} FPS (weighted)")
            if steady:
                print(f"   Steady-state FPS (full {self.config.chunk_length}-frame chunks, median): "
                      f"{steady[len(steady) // 2]:.2f} FPS")
        except Exception:  # noqa: BLE001
            pass
        if self.world > 1:   # rank 0 writes the manifest of all ranks' chunks
            from .dist import gather_objects
            parts = gather_objects(manifest)
            if self.rank != 0:
                torch.distributed.barrier()
                print(f"✅ Completed. Saved {len(saved_files)} chunks to {self.chunks_dir}")
                return saved_files
            manifest = sorted((m for part in parts for m in part), key=lambda m: m["chunk_index"])
        try:
            with open(os.path.join(self.config.output_dir, "chunks_manifest.json"), "w") as f:
                json.dump(manifest, f, indent=2)
        except Exception as e:  # noqa: BLE001
            print(f"⚠️  Failed to write manifest: {e}")
        try:
            metadata = {"chunk_length": int(self.config.chunk_length), "overlap": int(self.config.overlap),
                        "target_size": list(self.target_size) if self.target_size is not None else None}
            with open(os.path.join(self.config.output_dir, "chunk_metadata.json"), "w") as f:
                json.dump(metadata, f, indent=2)
        except Exception as e:  # noqa: BLE001
            print(f"⚠️  Failed to write chunk metadata: {e}")
        if self.world > 1:
            torch.distributed.barrier()     # metadata is on disk before any rank starts stage 2
        print(f"✅ Completed. Saved {len(saved_files)} chunks to {self.chunks_dir}")
        return saved_files
