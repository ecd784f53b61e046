"""Frame ingest: mirror of utils/image_utils.py:13-50 (calculate_target_size) and datasets/image_datasets.py:13-210
(ChunkImageDataset) for image files.  PIL decode + bilinear resize is what torchvision's Resize does on a PIL image
(transforms.Resize(size) -> img.resize((W, H), BILINEAR)); ToTensor = uint8 HWC -> float CHW / 255.
Video / torchcodec / undistortion inputs are out of scope."""
from __future__ import annotations

import math
import os
from typing import List, Tuple

import numpy as np
import torch
from torch.utils.data import Dataset


def target_size_for(W_orig: int, H_orig: int, pixel_limit: int = 255000) -> Tuple[int, int]:
    scale = math.sqrt(pixel_limit / (W_orig * H_orig)) if W_orig * H_orig > 0 else 1
    W_t, H_t = W_orig * scale, H_orig * scale
    k, m = round(W_t / 14), round(H_t / 14)
    while (k * 14) * (m * 14) > pixel_limit:
        if k / m > W_t / H_t:
            k -= 1
        else:
            m -= 1
    return max(1, m) * 14, max(1, k) * 14  # (H, W)


def calculate_target_size(first_image_path: str, pixel_limit: int = 255000) -> Tuple[int, int]:
    from PIL import Image
    W_orig, H_orig = Image.open(first_image_path).convert("RGB").size
    return target_size_for(W_orig, H_orig, pixel_limit)


def chunk_indices(n_frames: int, chunk_length: int, overlap: int) -> List[Tuple[int, int]]:
    """datasets/image_datasets.py:40-47 (note: no early stop, a tail made only of overlap frames is kept)."""
    out, start = [], 0
    while start < n_frames:
        end = min(start + chunk_length, n_frames)
        if end - start >= 2:
            out.append((start, end))
        start += chunk_length - overlap
    return out


def load_image(path: str, target_size: Tuple[int, int]) -> torch.Tensor:
    from PIL import Image
    if not os.path.exists(path):
        raise ValueError(f"Image file not found: {path}")
    img = Image.open(path).convert("RGB").resize((target_size[1], target_size[0]), Image.BILINEAR)
    arr = np.array(img, dtype=np.uint8)  # writable copy
    return torch.from_numpy(arr).permute(2, 0, 1).to(torch.float32).div(255.0)


class ChunkImageDataset(Dataset):
    def __init__(self, image_paths: List[str], chunk_length: int, overlap: int, target_size: Tuple[int, int],
                 device: str = "cpu", undistortion_maps=None):
        if undistortion_maps is not None:
            raise NotImplementedError("undistortion maps are out of scope for this build")
        self.image_paths, self.chunk_length, self.overlap = image_paths, chunk_length, overlap
        self.target_size = target_size
        self.chunk_indices = chunk_indices(len(image_paths), chunk_length, overlap)

    def __len__(self):
        return len(self.chunk_indices)

    def __getitem__(self, idx):
        s, e = self.chunk_indices[idx]
        paths = self.image_paths[s:e]
        chunk = torch.stack([load_image(p, self.target_size) for p in paths])
        return {"chunk": chunk, "start_idx": torch.tensor([s]), "end_idx": torch.tensor([e]), "chunk_paths": [paths]}
