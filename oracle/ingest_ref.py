"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): CPU restatement of the frame ingest of
datasets/image_datasets.py:186-208.

The reference resizes each decoded PIL frame with ``transforms.Resize(target_size)`` and converts it with ``ToTensor``.
On a PIL image torchvision's Resize is ``img.resize((W, H), Image.BILINEAR)``; the arithmetic therefore lives in the
third-party dependency Pillow (src/libImaging/Resample.c, ``ImagingResample`` for 8-bit images; this container has
Pillow 12.2.0, the reference pins no version), not in /root/reference.  It is restated here from the published
algorithm and pinned by running Pillow itself (tests/test_oracle_golden.py, tests/golden/ingest_*.npz made by
oracle/gen_golden_ingest.py):

  * per output coordinate the antialiased triangle filter: support = max(scale, 1), taps over
    [trunc(center - support + .5), trunc(center + support + .5)) clipped to the image, weights normalised in double;
  * weights quantised to 22-bit fixed point, round-half-away:  (int)(±0.5 + w * 2**22);
  * horizontal pass then vertical pass, each  clip8((2**21 + sum(pixel * k)) >> 22)  with a uint8 image in between;
  * ToTensor: uint8 HWC -> float32 CHW / 255.
"""
from __future__ import annotations

import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def resample_coeffs(in_size: int, out_size: int):
    """precompute_coeffs + normalize_coeffs_8bpc of Pillow's Resample.c for the bilinear (triangle) filter and the
    full-image box.  Returns bounds int32 [out][2] = (first tap, tap count) and kk int32 [out][ksize]."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = 0.0 + (xx + 0.5) * scale
        xmin = int(center - support + 0.5)          # C (int) cast: truncation toward zero
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = np.zeros(xmax, dtype=np.float64)
        ww = 0.0
        for x in range(xmax):
            a = (x + xmin - center + 0.5) * ss
            a = -a if a < 0.0 else a
            w[x] = 1.0 - a if a < 1.0 else 0.0
            ww += w[x]
        for x in range(xmax):
            if ww != 0.0:
                w[x] /= ww
            v = w[x] * (1 << PRECISION_BITS)
            kk[xx, x] = int(-0.5 + v) if w[x] < 0 else int(0.5 + v)
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def _pass(img: np.ndarray, bounds: np.ndarray, kk: np.ndarray, axis: int) -> np.ndarray:
    """One resampling pass over `axis` (0 = rows / vertical, 1 = columns / horizontal) of a uint8 HWC image."""
    src = np.moveaxis(img, axis, 0).astype(np.int64)
    out = np.empty((bounds.shape[0],) + src.shape[1:], dtype=np.uint8)
    for o in range(bounds.shape[0]):
        x0, n = int(bounds[o, 0]), int(bounds[o, 1])
        acc = np.tensordot(kk[o, :n].astype(np.int64), src[x0:x0 + n], axes=(0, 0)) + (1 << (PRECISION_BITS - 1))
        out[o] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return np.moveaxis(out, 0, axis)


def resize_bilinear_u8(img: np.ndarray, out_hw) -> np.ndarray:
    """PIL ``Image.resize((W, H), BILINEAR)`` on a uint8 HWC array: horizontal pass first, then vertical."""
    H1, W1 = out_hw
    xb, xk = resample_coeffs(img.shape[1], W1)
    yb, yk = resample_coeffs(img.shape[0], H1)
    return _pass(_pass(img, xb, xk, 1), yb, yk, 0)


def ingest_frames(frames_u8: np.ndarray, out_hw) -> np.ndarray:
    """uint8 [N][H0][W0][3] -> float32 [N][3][H][W] exactly as Resize + ToTensor of the reference's loader."""
    out = np.stack([resize_bilinear_u8(f, out_hw) for f in frames_u8])
    return (out.astype(np.float32) / np.float32(255.0)).transpose(0, 3, 1, 2).copy()
