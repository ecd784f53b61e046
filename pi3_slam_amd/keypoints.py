"""Grid keypoints: host-side mirror of utils/keypoint_extraction.py:32-255 (GridKeypointExtractor) + factory :394-419.

Only index generation lives here (a K x 2 table per frame); colours and the 3-D gather run on the device
(csrc/post.hip: pi3_gather_keypoints).  ALIKED is a third-party CNN the reference itself treats as optional
(falls back to grid, keypoint_extraction.py:407-411) and is out of scope: asking for it
selects the grid extractor with the same warning.
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np
import torch

from .hostmem import full_host, zeros_host


class GridKeypointExtractor:
    def __init__(self, max_num_keypoints: int = 512, grid_spacing: Optional[int] = None, device: str = "cuda",
                 seed: Optional[int] = 0):
        """seed: the reference subsamples an over-full grid with an unseeded device torch.randperm
        (keypoint_extraction.py:140-143); a seeded CPU generator makes the keypoint indices reproducible (and
        bit-comparable with the oracle).  seed=None keeps the global RNG like the reference."""
        self.device = device
        self.max_num_keypoints = max_num_keypoints
        self.grid_spacing = grid_spacing
        self._seed = seed
        self._gen = torch.Generator().manual_seed(seed) if seed is not None else None

    def reseed(self, chunk_index: int) -> None:
        """Restart the subsampling stream for a chunk: the keypoints of chunk c then do not depend on which chunks this
        process handled before it (chunk-parallel runs give the files of a single-process run)."""
        if self._seed is not None:
            self._gen.manual_seed(self._seed + 7919 * int(chunk_index))

    def _calculate_grid_spacing(self, H: int, W: int) -> int:
        if self.grid_spacing is not None:
            return self.grid_spacing
        margin = min(H, W) * 0.05
        eh, ew = H - 2 * margin, W - 2 * margin
        if eh <= 0 or ew <= 0:
            return max(H, W)
        spacing = int(np.sqrt((eh * ew) / self.max_num_keypoints))
        return max(8, min(spacing, min(H, W) // 4))

    def extract(self, images: torch.Tensor) -> Dict[str, torch.Tensor]:
        """images (N, C, H, W) or (1, N, C, H, W): only the shape is used.  -> CPU tensors keypoints (N, K, 2) f32
        (x, y), descriptors (N, K, 128) zeros, scores (N, K) ones."""
        if images.ndim == 5:
            images = images.squeeze(0)
        N, _, H, W = images.shape
        kps = []
        for _ in range(N):
            sp = self._calculate_grid_spacing(H, W)
            margin = min(H, W) * 0.05
            gx = torch.arange(margin, W - margin, sp)
            gy = torch.arange(margin, H - margin, sp)
            if len(gx) == 0 or len(gy) == 0:
                coords = torch.tensor([[W // 2, H // 2]], dtype=torch.float32)
            else:
                yy, xx = torch.meshgrid(gy, gx, indexing="ij")
                coords = torch.stack([xx.flatten(), yy.flatten()], dim=-1)
            if len(coords) > self.max_num_keypoints:
                idx = torch.randperm(len(coords), generator=self._gen)[: self.max_num_keypoints]
                coords = coords[idx]
            kps.append(coords)
        keypoints = torch.stack(kps, dim=0).to(torch.float32)
        K = keypoints.shape[1]
        # "constant": descriptors are all zero and scores all one (keypoint_extraction.py:150-151); the chunk creator
        # then writes its fp16 copies from cached constants instead of converting 2.5 M floats per chunk on the host
        # (calloc'ed / numpy-filled: a torch.zeros of 2.5 M floats per chunk is an OpenMP region on the launch path, hostmem.py)
        return dict(keypoints=keypoints, descriptors=zeros_host((N, K, 128), torch.float32),
                    scores=full_host((N, K), 1.0, torch.float32), constant=True)


def create_keypoint_extractor(keypoint_type: str = "grid", max_num_keypoints: int = 512,
                              detection_threshold: float = 0.005, device: str = "cuda", seed: Optional[int] = 0):
    """utils/keypoint_extraction.py:394-419."""
    kt = (keypoint_type or "").lower()
    if kt == "aliked":
        print("⚠️  ALIKED (lightglue) is not available in this build, falling back to grid-based extraction")
        kt = "grid"
    if kt == "grid":
        return GridKeypointExtractor(max_num_keypoints=max_num_keypoints, device=device, seed=seed)
    raise ValueError(f"Unknown keypoint type: {keypoint_type}. Supported types: 'aliked', 'grid'")
