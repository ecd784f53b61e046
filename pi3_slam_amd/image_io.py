"""Frame ingest: mirror of utils/image_utils.py:13-50 (calculate_target_size) and datasets/image_datasets.py:13-210
(ChunkImageDataset) for image files.  PIL decode + bilinear resize is what torchvision's Resize does on a PIL image
(transforms.Resize(size) -> img.resize((W, H), BILINEAR)); ToTensor = uint8 HWC -> float CHW / 255.
Video / torchcodec inputs are out of scope; undistortion: pi3_slam_amd/undistortion.py."""
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


def resample_coeffs(in_size: int, out_size: int):
    """Tap tables of Pillow's 8-bit bilinear resample (Resample.c: precompute_coeffs + normalize_coeffs_8bpc) for one
    axis, vectorised: bounds int32 [out][2] = (first tap, tap count), coefs int32 [out][ksize] in 22-bit fixed point.
    Every operation is an IEEE double operation in Pillow's order, so the tables equal Pillow's bit for bit."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    ss = 1.0 / filterscale
    center = 0.0 + (np.arange(out_size, dtype=np.float64) + 0.5) * scale
    xmin = np.maximum(np.trunc(center - support + 0.5).astype(np.int64), 0)
    xmax = np.minimum(np.trunc(center + support + 0.5).astype(np.int64), in_size) - xmin
    taps = np.arange(ksize, dtype=np.int64)[None, :]
    live = taps < xmax[:, None]
    a = np.abs((taps + xmin[:, None] - center[:, None] + 0.5) * ss)
    w = np.where(live & (a < 1.0), 1.0 - a, 0.0)
    ww = np.zeros(out_size, dtype=np.float64)
    for t in range(ksize):              # sequential accumulation like the C loop (summation order matters in fp64)
        ww = ww + w[:, t]
    w = np.where(ww[:, None] != 0.0, w / np.where(ww == 0.0, 1.0, ww)[:, None], w)
    kk = np.trunc(0.5 + w * float(1 << 22)).astype(np.int32)   # weights of the triangle filter are never negative
    kk[~live] = 0
    bounds = np.stack([xmin, xmax], axis=1).astype(np.int32)
    return bounds, kk


_COEF_CACHE = {}


def ingest_frames_device(frames_u8: torch.Tensor, target_size: Tuple[int, int]) -> torch.Tensor:
    """Device form of Resize + ToTensor for a stack of decoded frames: uint8 [N,H0,W0,3] on the GPU -> float32
    [N,3,H,W], bit-identical to load_image() on each frame (HIP kernels, csrc/ingest.hip)."""
    from . import ops
    H1, W1 = int(target_size[0]), int(target_size[1])
    N, H0, W0 = frames_u8.shape[:3]
    key = (H0, W0, H1, W1, str(frames_u8.device))
    if key not in _COEF_CACHE:
        xb, xk = resample_coeffs(W0, W1)
        yb, yk = resample_coeffs(H0, H1)
        _COEF_CACHE[key] = tuple(torch.from_numpy(np.ascontiguousarray(t)).to(frames_u8.device)
                                 for t in (xb, xk, yb, yk))
    xb, xk, yb, yk = _COEF_CACHE[key]
    return ops.ingest_frames(frames_u8.contiguous(), H1, W1, xb, xk, yb, yk)


def decode_frames_u8(paths: List[str], pin: bool = True) -> torch.Tensor:
    """PIL decode only (host): uint8 [N,H0,W0,3], pinned when a GPU is present so the upload can overlap compute
    (pin=False inside DataLoader workers: the loader's own pin_memory thread does it)."""
    from PIL import Image
    frames = []
    for p in paths:
        if not os.path.exists(p):
            raise ValueError(f"Image file not found: {p}")
        frames.append(np.array(Image.open(p).convert("RGB"), dtype=np.uint8))
    if any(f.shape != frames[0].shape for f in frames):
        raise ValueError("frames of one chunk must share a size for the batched device resize")
    out = torch.from_numpy(np.stack(frames))
    return out.pin_memory() if (pin and torch.cuda.is_available()) else out


class ThreadedChunkLoader:
    """In-process replacement for the reference's DataLoader workers (slam/offline_chunk_creator.py:279-287): a producer
    thread walks the chunks, a thread pool decodes (and, for the host-resize path, resizes) the frames of a chunk in
    parallel - PIL's decoder and resampler release the GIL - straight into a pinned staging buffer, `depth` chunks
    ahead.  Why not worker processes: forking a process that holds a GPU context costs the device ~2 s of evicted
    queues at start-up on this stack, every batch crosses a shared-memory pickle and a second copy into pinned memory,
    and a chunk (100 frames) is ONE dataset item, so a worker decodes its frames one after the other.
    Yields the same dictionaries as DataLoader(batch_size=1) over ChunkImageDataset: 'chunk_u8' (1,N,H0,W0,3) uint8 or
    'chunk' (1,N,3,H,W) fp32, 'start_idx' / 'end_idx' (1,1) int64, 'chunk_paths' [[[p0],[p1],...]] (the nesting the
    reference's collate + pin pass produces on a GPU box)."""

    def __init__(self, dataset: "ChunkImageDataset", indices: List[int], threads: int = 8, depth: int = 2,
                 pin: bool = True):
        self.ds, self.indices, self.threads, self.depth = dataset, list(indices), max(1, threads), max(1, depth)
        self.pin = pin and torch.cuda.is_available()

    def __len__(self):
        return len(self.indices)

    def __iter__(self):
        import queue
        import threading
        from concurrent.futures import ThreadPoolExecutor
        from PIL import Image
        ds = self.ds
        q: "queue.Queue" = queue.Queue(maxsize=self.depth)
        stop = threading.Event()
        buffers: dict = {}            # (shape, dtype) -> free pinned staging tensors (returned when the consumer moves on)
        lock = threading.Lock()

        def staging(shape, dtype):
            with lock:
                free = buffers.setdefault((tuple(shape), dtype), [])
                if free:
                    return free.pop()
            t = torch.empty(shape, dtype=dtype)
            return t.pin_memory() if self.pin else t

        def decode_one(dst, j, path):
            if not os.path.exists(path):
                raise ValueError(f"Image file not found: {path}")
            img = Image.open(path).convert("RGB")
            if ds.decode_only:
                a = np.array(img, dtype=np.uint8)
                if tuple(a.shape) != tuple(dst.shape[1:]):
                    raise ValueError("frames of one chunk must share a size for the batched device resize")
                dst[j].copy_(torch.from_numpy(a))
            else:
                a = np.array(img.resize((ds.target_size[1], ds.target_size[0]), Image.BILINEAR), dtype=np.uint8)
                dst[j].copy_(torch.from_numpy(a).permute(2, 0, 1).to(torch.float32).div(255.0))

        def produce():
            try:
                with ThreadPoolExecutor(self.threads, thread_name_prefix="frame-decode") as pool:
                    for idx in self.indices:
                        if stop.is_set():
                            return
                        s, e = ds.chunk_indices[idx]
                        paths = ds.image_paths[s:e]
                        if ds.decode_only:
                            w0, h0 = Image.open(paths[0]).size
                            buf = staging((e - s, h0, w0, 3), torch.uint8)
                        else:
                            buf = staging((e - s, 3, ds.target_size[0], ds.target_size[1]), torch.float32)
                        list(pool.map(lambda jp: decode_one(buf, jp[0], jp[1]), enumerate(paths)))
                        item = {"chunk_u8" if ds.decode_only else "chunk": buf[None],
                                "start_idx": torch.tensor([[s]]), "end_idx": torch.tensor([[e]]),
                                "chunk_paths": [[[p] for p in paths]], "_staging": buf}
                        while not stop.is_set():
                            try:
                                q.put(item, timeout=0.1)
                                break
                            except queue.Full:
                                continue
                q.put(None)
            except BaseException as exc:  # noqa: BLE001 - delivered to the consumer
                q.put(exc)

        th = threading.Thread(target=produce, name="chunk-loader", daemon=True)
        th.start()
        held = []                      # staging buffers the consumer may still be uploading from
        try:
            while True:
                item = q.get()
                if item is None:
                    return
                if isinstance(item, BaseException):
                    raise item
                buf = item.pop("_staging")
                held.append(buf)
                if len(held) > self.depth + 2:     # two chunks further on, the upload of this one has long finished
                    old = held.pop(0)
                    with lock:
                        buffers.setdefault((tuple(old.shape), old.dtype), []).append(old)
                yield item
        finally:
            stop.set()


class ChunkImageDataset(Dataset):
    def __init__(self, image_paths: List[str], chunk_length: int, overlap: int, target_size: Tuple[int, int],
                 device: str = "cpu", undistortion_maps=None, decode_only: bool = False):
        if undistortion_maps is not None and not decode_only:
            raise NotImplementedError("undistortion runs on the GPU: build the dataset with decode_only=True and call "
                                      "UndistortionMaps.undistort_frames_device (pi3_slam_amd/undistortion.py)")
        self.image_paths, self.chunk_length, self.overlap = image_paths, chunk_length, overlap
        self.target_size = target_size
        self.decode_only = decode_only   # items carry the decoded uint8 frames ("chunk_u8"); the resize runs on the GPU
        self.chunk_indices = chunk_indices(len(image_paths), chunk_length, overlap)

    def __len__(self):
        return len(self.chunk_indices)

    def __getitem__(self, idx):
        s, e = self.chunk_indices[idx]
        paths = self.image_paths[s:e]
        if self.decode_only:
            return {"chunk_u8": decode_frames_u8(paths, pin=False), "start_idx": torch.tensor([s]),
                    "end_idx": torch.tensor([e]), "chunk_paths": [paths]}
        chunk = torch.stack([load_image(p, self.target_size) for p in paths])
        return {"chunk": chunk, "start_idx": torch.tensor([s]), "end_idx": torch.tensor([e]), "chunk_paths": [paths]}

    def load_chunk_device(self, idx, device="cuda"):
        """Same item as __getitem__ with the resize + ToTensor done on the GPU (frames cross PCIe as uint8)."""
        s, e = self.chunk_indices[idx]
        paths = self.image_paths[s:e]
        frames = decode_frames_u8(paths).to(device, non_blocking=True)
        chunk = ingest_frames_device(frames, self.target_size)
        return {"chunk": chunk, "start_idx": torch.tensor([s]), "end_idx": torch.tensor([e]), "chunk_paths": [paths]}
