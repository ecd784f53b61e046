"""Host-side byte moving for the consumer side of the pipeline, without ATen CPU operators.

Why (round 5, tools/dev_host_stall.py, gpurun_out/r5c): every ATen CPU operator over >= 32 768 elements (`clone`, `copy_`,
`.to(dtype)`, `zeros`) and every advanced index opens an OpenMP parallel region.  torch sizes its intra-op pool by the
HOST's core count; on a box whose CPU share is a cgroup quota (the GPU boxes: 16 of the host's cores) such a region was
measured to stall its caller for 85-110 ms once in ~25 chunks - the "one 92 ms sample among 1-2 ms ones" of round 4's
pipeline test and the 186 ms `consume_ms_max` of the online-stream extra.  The per-chunk host work is a few hundred KB:
plain memcpy / calloc on the calling thread is both faster and free of that hazard."""
from __future__ import annotations

import ctypes

import numpy as np
import torch

_NP = {torch.float16: np.float16, torch.float32: np.float32, torch.float64: np.float64, torch.uint8: np.uint8,
       torch.int16: np.int16, torch.int32: np.int32, torch.int64: np.int64, torch.bool: np.bool_}


def memcpy_into(dst: torch.Tensor, src: torch.Tensor) -> torch.Tensor:
    """dst[...] = src for two contiguous host tensors of equal byte size (dtype / shape may differ: a byte copy)."""
    n = src.numel() * src.element_size()
    assert dst.is_contiguous() and src.is_contiguous() and dst.numel() * dst.element_size() == n
    assert dst.device.type == "cpu" and src.device.type == "cpu"
    if n:
        ctypes.memmove(dst.data_ptr(), src.data_ptr(), n)
    return dst


def clone_host(t: torch.Tensor) -> torch.Tensor:
    """t.clone() for a host tensor (pinned or pageable) as one memcpy on the calling thread."""
    src = t if t.is_contiguous() else t.contiguous()
    return memcpy_into(torch.empty(src.shape, dtype=src.dtype), src)


def pinned_copy(t: torch.Tensor) -> torch.Tensor:
    """A pinned staging copy of a host tensor (torch's caching host allocator keeps the block until the copy queued from
    it has run)."""
    src = t if t.is_contiguous() else t.contiguous()
    return memcpy_into(torch.empty(src.shape, dtype=src.dtype, pin_memory=True), src)


def zeros_host(shape, dtype: torch.dtype) -> torch.Tensor:
    """torch.zeros(shape, dtype) through calloc: the pages are zero when first touched, nobody writes them."""
    return torch.from_numpy(np.zeros(tuple(shape), dtype=_NP[dtype]))


def full_host(shape, value, dtype: torch.dtype) -> torch.Tensor:
    return torch.from_numpy(np.full(tuple(shape), value, dtype=_NP[dtype]))
