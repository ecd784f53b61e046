"""Tensor-level wrappers over the C-ABI (one Python function per entry point of include/pi3slam_hip.h).

Each wrapper only validates shapes/dtypes, allocates the output through torch's allocator and forwards raw device
pointers + the current stream.  No arithmetic happens here.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Optional

import torch

from . import lib as _L

BF16, F32 = 0, 1
ACT_NONE, ACT_GELU, ACT_RELU = 0, 1, 2
QSCALE = 0.125 * math.log2(math.e)  # head_dim^-0.5 * log2(e), folded into q (attn.hip works in the exp2 domain)


def _dt(t: torch.Tensor) -> int:
    if t.dtype == torch.bfloat16:
        return BF16
    if t.dtype == torch.float32:
        return F32
    raise _L.Pi3HipError(f"unsupported dtype {t.dtype}")


def gemm(a: torch.Tensor, w: torch.Tensor, out: torch.Tensor, *, M: Optional[int] = None, K: Optional[int] = None,
         bias: Optional[torch.Tensor] = None, gamma: Optional[torch.Tensor] = None,
         resid: Optional[torch.Tensor] = None, act: int = ACT_NONE, rpg: int = 0, gstride: int = 0, goff: int = 0,
         addtab: Optional[torch.Tensor] = None, qscale: float = 1.0, qcols: int = 0) -> torch.Tensor:
    """out[orow(m), :N] = resid + gamma * act((a[m] . w^T + bias) * qscale[n < qcols]) + addtab[m % rpg].

    a: [>=M, lda] (bf16 or f32), w: [N, ldw] same dtype, out: 2-D bf16/f32 with row stride out.stride(0).
    """
    lib = _L.load()
    assert a.dim() == 2 and w.dim() == 2 and out.dim() == 2 and a.dtype == w.dtype
    assert a.stride(1) == 1 and w.stride(1) == 1 and out.stride(1) == 1
    M = a.shape[0] if M is None else M
    K = w.shape[1] if K is None else K
    N = w.shape[0]
    rc = lib.pi3_gemm(a.data_ptr(), a.stride(0), w.data_ptr(), w.stride(0), M, N, K, _dt(a),
                      _L.ptr(bias), _L.ptr(gamma), _L.ptr(resid), resid.stride(0) if resid is not None else 0,
                      out.data_ptr(), out.stride(0), _dt(out), act, rpg, gstride, goff,
                      _L.ptr(addtab), addtab.stride(0) if addtab is not None else 0, float(qscale), int(qcols),
                      _L.stream_ptr())
    _L.check(rc, "pi3_gemm")
    return out


def attention(qkv: torch.Tensor, out: torch.Tensor, B: int, S: int, H: int) -> torch.Tensor:
    """qkv: packed [B*S, 3*H*64] bf16 (q pre-scaled by QSCALE); out: [B*S, H*64] bf16."""
    lib = _L.load()
    assert qkv.dtype == torch.bfloat16 and out.dtype == torch.bfloat16 and qkv.dim() == 2 and out.dim() == 2
    assert qkv.shape[0] >= B * S and qkv.shape[1] == 3 * H * 64 and qkv.stride(1) == 1 and out.stride(1) == 1
    ts = qkv.stride(0)
    base = qkv.data_ptr()
    rc = lib.pi3_attention(base, base + 2 * H * 64, base + 4 * H * 64, ts, S * ts, out.data_ptr(), out.stride(0),
                           S * out.stride(0), B, S, H, 64, _L.stream_ptr())
    _L.check(rc, "pi3_attention")
    return out


def layernorm(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, out: torch.Tensor, eps: float = 1e-6, *,
              rows: Optional[int] = None, T: int = 0, nspecial: int = 0,
              special: Optional[torch.Tensor] = None) -> torch.Tensor:
    lib = _L.load()
    assert x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1 and out.stride(1) == 1
    rows = x.shape[0] if rows is None else rows
    rc = lib.pi3_layernorm(x.data_ptr(), x.stride(0), rows, x.shape[1], w.data_ptr(), b.data_ptr(), float(eps),
                           out.data_ptr(), out.stride(0), _dt(out), T, nspecial, _L.ptr(special), _L.stream_ptr())
    _L.check(rc, "pi3_layernorm")
    return out


def qknorm_rope(qkv: torch.Tensor, rows: int, H: int, T: int, pos: Optional[torch.Tensor],
                cs: Optional[torch.Tensor], qw=None, qb=None, kw=None, kb=None, eps: float = 1e-6,
                qscale: float = QSCALE, do_rope: bool = True) -> torch.Tensor:
    lib = _L.load()
    assert qkv.dtype == torch.bfloat16 and qkv.is_contiguous() and qkv.shape[1] == 3 * H * 64
    if pos is not None:
        assert pos.dtype == torch.int32 and pos.is_contiguous()
    rc = lib.pi3_qknorm_rope(qkv.data_ptr(), rows, H, T, _L.ptr(pos), _L.ptr(cs), _L.ptr(qw), _L.ptr(qb),
                             _L.ptr(kw), _L.ptr(kb), float(eps), float(qscale), int(do_rope), _L.stream_ptr())
    _L.check(rc, "pi3_qknorm_rope")
    return qkv


def cast_rows(x: torch.Tensor, out: torch.Tensor, rows: Optional[int] = None, cols: Optional[int] = None):
    lib = _L.load()
    assert x.dtype == torch.float32 and x.stride(1) == 1 and out.stride(1) == 1
    rows = x.shape[0] if rows is None else rows
    cols = x.shape[1] if cols is None else cols
    rc = lib.pi3_cast_rows(x.data_ptr(), x.stride(0), out.data_ptr(), out.stride(0), rows, cols, _dt(out),
                           _L.stream_ptr())
    _L.check(rc, "pi3_cast_rows")
    return out


def patch_gather(imgs: torch.Tensor, out: torch.Tensor, mean, std) -> torch.Tensor:
    """imgs: [F, 3, H, W] f32 -> out: [F*P, KP] bf16 normalised patch rows."""
    lib = _L.load()
    assert imgs.dtype == torch.float32 and imgs.is_contiguous() and out.dtype == torch.bfloat16 and out.is_contiguous()
    F, _, H, W = imgs.shape
    m3 = (C.c_float * 3)(*[float(v) for v in mean])
    s3 = (C.c_float * 3)(*[float(v) for v in std])
    rc = lib.pi3_patch_gather(imgs.data_ptr(), F, H, W, out.data_ptr(), out.shape[1], m3, s3, _L.stream_ptr())
    _L.check(rc, "pi3_patch_gather")
    return out


def resample_grid(src: torch.Tensor, wy: torch.Tensor, wx: torch.Tensor) -> torch.Tensor:
    """src [Mi, Mj, D] f32, wy [oh, Mi], wx [ow, Mj] -> [oh, ow, D]."""
    lib = _L.load()
    Mi, Mj, D = src.shape
    oh, ow = wy.shape[0], wx.shape[0]
    dst = torch.empty(oh, ow, D, device=src.device, dtype=torch.float32)
    rc = lib.pi3_resample_grid(src.data_ptr(), Mi, Mj, D, wy.data_ptr(), wx.data_ptr(), oh, ow, dst.data_ptr(),
                               _L.stream_ptr())
    _L.check(rc, "pi3_resample_grid")
    return dst


def fill_tokens(x: torch.Tensor, F: int, T: int, t0: int, vals: torch.Tensor) -> None:
    lib = _L.load()
    assert x.dtype == torch.float32 and vals.dtype == torch.float32 and vals.is_contiguous()
    rc = lib.pi3_fill_tokens(x.data_ptr(), F, T, x.shape[1], t0, vals.shape[0], vals.data_ptr(), _L.stream_ptr())
    _L.check(rc, "pi3_fill_tokens")


def recipe_fill(out: torch.Tensor, seed: int, offset: float, scale: float) -> torch.Tensor:
    lib = _L.load()
    assert out.is_contiguous()
    rc = lib.pi3_recipe_fill(out.data_ptr(), out.numel(), C.c_ulonglong(seed), float(offset), float(scale), _dt(out),
                             _L.stream_ptr())
    _L.check(rc, "pi3_recipe_fill")
    return out


def unpatchify_points(pfeat: torch.Tensor, cfeat: torch.Tensor, poses: torch.Tensor, F: int, H: int, W: int, T: int,
                      tok_off: int, local_points: torch.Tensor, points: torch.Tensor, conf: torch.Tensor) -> None:
    lib = _L.load()
    for t in (pfeat, cfeat, poses, local_points, points, conf):
        assert t.dtype == torch.float32
    assert poses.is_contiguous() and local_points.is_contiguous() and points.is_contiguous() and conf.is_contiguous()
    rc = lib.pi3_unpatchify_points(pfeat.data_ptr(), pfeat.stride(0), cfeat.data_ptr(), cfeat.stride(0),
                                   poses.data_ptr(), F, H, W, T, tok_off, local_points.data_ptr(), points.data_ptr(),
                                   conf.data_ptr(), _L.stream_ptr())
    _L.check(rc, "pi3_unpatchify_points")


def camera_tail(feat: torch.Tensor, T: int, tok_off: int, F: int, P: int, w: dict, poses: torch.Tensor) -> None:
    """feat: [F*T, C] fp32 (output of the ResConv blocks); patch tokens start at row tok_off of each frame."""
    lib = _L.load()
    assert feat.dtype == torch.float32 and feat.stride(1) == 1 and poses.is_contiguous()
    ld = feat.stride(0)
    base = feat.data_ptr() + tok_off * ld * 4
    g = lambda n: w[n].data_ptr()
    rc = lib.pi3_camera_tail(base, ld, T * ld, F, P, feat.shape[1],
                             g("camera_head.more_mlps.0.weight"), g("camera_head.more_mlps.0.bias"),
                             g("camera_head.more_mlps.2.weight"), g("camera_head.more_mlps.2.bias"),
                             g("camera_head.fc_t.weight"), g("camera_head.fc_t.bias"),
                             g("camera_head.fc_rot.weight"), g("camera_head.fc_rot.bias"),
                             poses.data_ptr(), _L.stream_ptr())
    _L.check(rc, "pi3_camera_tail")
