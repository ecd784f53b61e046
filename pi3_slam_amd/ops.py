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

BF16, F32, F16 = 0, 1, 2
_16BIT = (torch.bfloat16, torch.float16)
ACT_NONE, ACT_GELU, ACT_RELU = 0, 1, 2
QSCALE = 0.125 * math.log2(math.e)  # head_dim^-0.5 * log2(e), folded into q (attn.hip works in the exp2 domain)


def _dt(t: torch.Tensor) -> int:
    if t.dtype == torch.bfloat16:
        return BF16
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.float16:       # IEEE half: the MoGe path (the reference runs it under fp16 autocast)
        return F16
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


def gemm_qkv(a: torch.Tensor, w: torch.Tensor, qkv: torch.Tensor, *, M: int, H: int, bias: Optional[torch.Tensor],
             T: int, pos: Optional[torch.Tensor] = None, cs: Optional[torch.Tensor] = None, qw=None, qb=None, kw=None,
             kb=None, eps: float = 1e-5, qscale: float = QSCALE, k2max: Optional[torch.Tensor] = None,
             attn_B: int = 0, attn_S: int = 0) -> torch.Tensor:
    """qkv projection + per-head q/k LayerNorm(64) + RoPE-2D + softmax-scale fold (+ max |k|^2 per (batch, head) into
    k2max) in one launch where the 256x256 kernel applies, else projection + the stand-alone passes."""
    lib = _L.load()
    assert a.dtype == torch.bfloat16 and w.dtype == torch.bfloat16 and qkv.dtype == torch.bfloat16
    assert a.stride(1) == 1 and w.stride(1) == 1 and qkv.is_contiguous() and qkv.shape[1] == 3 * H * 64
    assert w.shape[0] == 3 * H * 64
    if pos is not None:
        assert pos.dtype == torch.int32 and pos.is_contiguous() and cs is not None
    rc = lib.pi3_gemm_qkv(a.data_ptr(), a.stride(0), w.data_ptr(), w.stride(0), M, w.shape[1], H, _L.ptr(bias),
                          qkv.data_ptr(), qkv.stride(0), T, _L.ptr(pos), _L.ptr(cs) if pos is not None else None,
                          _L.ptr(qw), _L.ptr(qb), _L.ptr(kw), _L.ptr(kb), float(eps), float(qscale), _L.ptr(k2max),
                          attn_B, attn_S, _L.stream_ptr())
    _L.check(rc, "pi3_gemm_qkv")
    return qkv


def attention(qkv: torch.Tensor, out: torch.Tensor, B: int, S: int, H: int,
              k2max: Optional[torch.Tensor] = None) -> torch.Tensor:
    """qkv: packed [B*S, 3*H*64] bf16 (q pre-scaled by QSCALE); out: [B*S, H*64] bf16.
    k2max: f32 [B*H] already holding max_s |k|^2 per (batch, head) (written by the fused qkv epilogue); None -> a
    workspace is taken from torch's allocator and the call fills it with its own pre-pass."""
    lib = _L.load()
    assert qkv.dtype in _16BIT and out.dtype == qkv.dtype and qkv.dim() == 2 and out.dim() == 2
    assert qkv.shape[0] >= B * S and qkv.shape[1] == 3 * H * 64 and qkv.stride(1) == 1 and out.stride(1) == 1
    ts = qkv.stride(0)
    base = qkv.data_ptr()
    ready = 1
    if k2max is None:
        ready = 0
        k2max = torch.empty(B * H, device=qkv.device, dtype=torch.float32) \
            if S >= 4096 and qkv.dtype == torch.bfloat16 else None
    else:
        assert k2max.dtype == torch.float32 and k2max.numel() >= B * H and k2max.is_contiguous()
    rc = lib.pi3_attention(base, base + 2 * H * 64, base + 4 * H * 64, ts, S * ts, out.data_ptr(), out.stride(0),
                           S * out.stride(0), B, S, H, 64, _dt(qkv), _L.ptr(k2max), ready, _L.stream_ptr())
    _L.check(rc, "pi3_attention")
    return out


def attention_path_counters(counters: Optional[torch.Tensor]) -> None:
    """Register (or, with None, remove) the diagnostic counter block of pi3_attention's 64-row kernel: int32 [2, 2, 32]
    on the device, zeroed by the caller = [eight-wave (global) | four-wave (frame-wise)] x [bounded-score loop |
    online-max loop] x 32 slots; `counters.sum(-1)` are waves.  Synchronise before changing it."""
    lib = _L.load()
    if counters is not None:
        assert counters.is_cuda and counters.dtype == torch.int32 and counters.numel() == 128 and counters.is_contiguous()
    _L.check(lib.pi3_attention_path_counters(_L.ptr(counters)), "pi3_attention_path_counters")


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


_ROPE_DT = {torch.bfloat16: 0, torch.float32: 1, torch.float16: 2}


def rope_2d(tokens: torch.Tensor, positions: torch.Tensor, base: float, fwd: float) -> None:
    """`curope.rope_2d(tokens, positions, base, fwd)` (pi3/models/curope/curope.cpp:49-68): IN PLACE on tokens
    (B, N, H, D) - any strides with the last dim contiguous - with positions (B, N, 2) int64 = (y, x).  The argument
    checks and their messages are the reference's TORCH_CHECKs (curope.cpp:54-59, kernels.cu:11-14, 91-94), raised as
    RuntimeError like a failing TORCH_CHECK is; one is relaxed: kernels.cu:91 also wants stride(2) == D, which the
    transposed view `cuRoPE2D.forward` makes of a contiguous (B, heads, N, D) tensor does not have."""
    lib = _L.load()
    if tokens.dim() != 4:
        raise RuntimeError("tokens must have 4 dimensions")
    if positions.dim() != 3:
        raise RuntimeError("positions must have 3 dimensions")
    if tokens.size(0) != positions.size(0):
        raise RuntimeError("batch size differs between tokens & positions")
    if tokens.size(1) != positions.size(1):
        raise RuntimeError("seq_length differs between tokens & positions")
    if positions.size(2) != 2:
        raise RuntimeError("positions.shape[2] must be equal to 2")
    if tokens.is_cuda != positions.is_cuda:
        raise RuntimeError("tokens and positions are not on the same device")
    if not tokens.is_cuda:
        raise _L.Pi3HipError("pi3_rope_2d runs on the GPU only (the reference's rope_2d_cpu is restated in oracle/)")
    B, N, H, D = tokens.shape
    if tokens.stride(3) != 1:
        raise RuntimeError("tokens are not contiguous")
    if not positions.is_contiguous():
        raise RuntimeError("positions are not contiguous")
    if D % 4 != 0:
        raise RuntimeError("token dim must be multiple of 4")
    if tokens.dtype not in _ROPE_DT or positions.dtype != torch.int64:
        raise RuntimeError(f"rope_2d: unsupported dtypes {tokens.dtype} / {positions.dtype}")
    rc = lib.pi3_rope_2d(tokens.data_ptr(), positions.data_ptr(), B, N, H, D, tokens.stride(0), tokens.stride(1),
                         tokens.stride(2), float(base), float(fwd), _ROPE_DT[tokens.dtype], _L.stream_ptr())
    _L.check(rc, "pi3_rope_2d")


class cuRoPE2D(torch.nn.Module):
    """Drop-in for `models.curope.cuRoPE2D` (pi3/models/curope/curope2d.py:33-40; selected by pos_embed.py:104-106 when
    the extension is importable): forward(tokens (B, heads, ntokens, dim), positions (B, ntokens, 2)) rotates `tokens` in
    place through the transposed view and returns it.  Inference only (the reference's autograd wrapper calls the same
    entry with fwd = -F0 for the backward pass; `rope_2d(..., -F0)` is available for that)."""

    def __init__(self, freq: float = 100.0, F0: float = 1.0):
        super().__init__()
        self.base = freq
        self.F0 = F0

    def forward(self, tokens: torch.Tensor, positions: torch.Tensor) -> torch.Tensor:
        rope_2d(tokens.transpose(1, 2), positions, self.base, self.F0)
        return tokens


def cast_rows(x: torch.Tensor, out: torch.Tensor, rows: Optional[int] = None, cols: Optional[int] = None,
              in_cols: Optional[int] = None):
    """out[r, :cols] = x[r, :cols]; with in_cols < cols only x[r, :in_cols] is read and out[r, in_cols:cols] = 0."""
    lib = _L.load()
    assert x.dtype == torch.float32 and x.stride(1) == 1 and out.stride(1) == 1
    rows = x.shape[0] if rows is None else rows
    cols = x.shape[1] if cols is None else cols
    in_cols = cols if in_cols is None else in_cols
    assert in_cols <= x.shape[1] and cols <= out.shape[1]
    rc = lib.pi3_cast_rows_pad(x.data_ptr(), x.stride(0), in_cols, out.data_ptr(), out.stride(0), rows, cols, _dt(out),
                               _L.stream_ptr())
    _L.check(rc, "pi3_cast_rows")
    return out


def patch_gather(imgs: torch.Tensor, out: torch.Tensor, mean, std) -> torch.Tensor:
    """imgs: [F, 3, H, W] f32 -> out: [F*P, KP] bf16 normalised patch rows."""
    lib = _L.load()
    assert imgs.dtype == torch.float32 and imgs.is_contiguous() and out.dtype in _16BIT and out.is_contiguous()
    F, _, H, W = imgs.shape
    m3 = (C.c_float * 3)(*[float(v) for v in mean])
    s3 = (C.c_float * 3)(*[float(v) for v in std])
    rc = lib.pi3_patch_gather(imgs.data_ptr(), F, H, W, out.data_ptr(), out.shape[1], _dt(out), m3, s3, _L.stream_ptr())
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
    assert out.is_contiguous() and out.dtype != torch.float16      # bf16 / f32 only (half weights: fill f32, then .half())
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


# ----------------------------------------------------------------------------------------------- post-processing
def compute_masks(conf: torch.Tensor, local_points: torch.Tensor, conf_thr: float = 0.1,
                  rtol: float = 0.03) -> torch.Tensor:
    """conf [F,H,W,1] or [F,H,W], local_points [F,H,W,3] (f32, contiguous) -> uint8 mask [F,H,W]."""
    lib = _L.load()
    F, H, W = local_points.shape[:3]
    assert conf.is_contiguous() and local_points.is_contiguous() and conf.numel() == F * H * W
    out = torch.empty(F, H, W, device=conf.device, dtype=torch.uint8)
    rc = lib.pi3_compute_masks(conf.data_ptr(), local_points.data_ptr(), F, H, W, float(conf_thr), float(rtol),
                               out.data_ptr(), _L.stream_ptr())
    _L.check(rc, "pi3_compute_masks")
    return out


def masked_ratio_median(num: torch.Tensor, den: torch.Tensor, den_stride: int, mask: torch.Tensor,
                        n: int) -> torch.Tensor:
    """-> device tensor [median, count] (f32).  den is addressed as den_ptr[i * den_stride]."""
    lib = _L.load()
    assert num.dtype == torch.float32 and den.dtype == torch.float32 and mask.dtype == torch.uint8
    out = torch.empty(2, device=num.device, dtype=torch.float32)
    rc = lib.pi3_masked_ratio_median(num.data_ptr(), den.data_ptr(), den_stride, mask.data_ptr(), n, out.data_ptr(),
                                     _L.stream_ptr())
    _L.check(rc, "pi3_masked_ratio_median")
    return out


def apply_scale(scale_dev: torch.Tensor, local_points: torch.Tensor, points: torch.Tensor,
                poses: torch.Tensor) -> None:
    lib = _L.load()
    assert local_points.is_contiguous() and points.is_contiguous() and poses.is_contiguous()
    F = poses.numel() // 16
    rc = lib.pi3_apply_scale(scale_dev.data_ptr(), local_points.data_ptr(), points.data_ptr(), local_points.numel(),
                             poses.data_ptr(), F, _L.stream_ptr())
    _L.check(rc, "pi3_apply_scale")


def gather_keypoints(points, local_points, conf, masks, images, keypoints):
    """Dense maps [F,H,W,*] + keypoints f32 [F,K,2] -> dict of packed per-keypoint tensors (fp16 / bool)."""
    lib = _L.load()
    F, H, W = points.shape[:3]
    K = keypoints.shape[1]
    dev = points.device
    for t in (points, local_points, conf, masks, keypoints):
        assert t.is_contiguous()
    assert keypoints.dtype == torch.float32 and masks.dtype == torch.uint8
    o_points = torch.empty(F, K, 3, device=dev, dtype=torch.float16)
    o_local = torch.empty(F, K, 3, device=dev, dtype=torch.float16)
    o_conf = torch.empty(F, K, 1, device=dev, dtype=torch.float16)
    o_mask = torch.empty(F, K, 1, device=dev, dtype=torch.uint8)
    o_kps = torch.empty(F, K, 2, device=dev, dtype=torch.float16)
    o_colors = torch.empty(F, K, 3, device=dev, dtype=torch.float16) if images is not None else None
    if images is not None:
        assert images.is_contiguous() and images.dtype == torch.float32 and tuple(images.shape) == (F, 3, H, W)
    rc = lib.pi3_gather_keypoints(points.data_ptr(), local_points.data_ptr(), conf.data_ptr(), masks.data_ptr(),
                                  _L.ptr(images), keypoints.data_ptr(), F, H, W, K, o_points.data_ptr(),
                                  o_local.data_ptr(), o_conf.data_ptr(), o_mask.data_ptr(), _L.ptr(o_colors),
                                  o_kps.data_ptr(), _L.stream_ptr())
    _L.check(rc, "pi3_gather_keypoints")
    return dict(points=o_points, local_points=o_local, conf=o_conf, masks=o_mask.bool(), colors=o_colors,
                keypoints=o_kps)


# ----------------------------------------------------------------------------------------------- Sim(3) alignment
def sim3_match_keypoints(kp_ref: torch.Tensor, kp_qry: torch.Tensor) -> torch.Tensor:
    """kp_*: f16 [ov, K, 2] -> int32 [ov, K] (index into the ref view's keypoints or -1)."""
    lib = _L.load()
    assert kp_ref.dtype == torch.float16 and kp_qry.dtype == torch.float16 and kp_ref.shape == kp_qry.shape
    kp_ref, kp_qry = kp_ref.contiguous(), kp_qry.contiguous()
    ov, K = kp_ref.shape[:2]
    idx = torch.empty(ov, K, device=kp_ref.device, dtype=torch.int32)
    rc = lib.pi3_sim3_match_keypoints(kp_ref.data_ptr(), kp_qry.data_ptr(), ov, K, idx.data_ptr(), _L.stream_ptr())
    _L.check(rc, "pi3_sim3_match_keypoints")
    return idx


def sim3_umeyama(pts_ref: torch.Tensor, pts_qry: torch.Tensor, idx: torch.Tensor, last_ref_pose: torch.Tensor,
                 w_ref: Optional[torch.Tensor] = None, w_qry: Optional[torch.Tensor] = None,
                 use_filter: bool = True) -> torch.Tensor:
    """-> f64 device tensor [33]: s, R(9), t(3), M(16), n_used, n_common, median, rms.
    w_ref / w_qry [ov, K]: uint8 validity (a pair takes part or not) or float32 weights (the weighted Umeyama:
    pair weight = w_ref[ref track] * w_qry[qry keypoint]); both of one kind."""
    lib = _L.load()
    assert pts_ref.dtype == pts_qry.dtype and pts_ref.dtype in (torch.float16, torch.float32) and idx.dtype == torch.int32
    pts_ref, pts_qry, idx = pts_ref.contiguous(), pts_qry.contiguous(), idx.contiguous()
    flags = int(bool(use_filter)) | (2 if pts_ref.dtype == torch.float32 else 0)
    assert last_ref_pose.dtype == torch.float32
    last_ref_pose = last_ref_pose.contiguous()
    ov, K = idx.shape
    assert pts_ref.numel() == pts_qry.numel() == ov * K * 3
    kinds = {w.dtype for w in (w_ref, w_qry) if w is not None}
    assert kinds <= {torch.uint8} or kinds <= {torch.float32}, "w_ref / w_qry: both uint8 validity or both float32 weights"
    for w in (w_ref, w_qry):
        assert w is None or (w.is_contiguous() and w.numel() == ov * K and w.device == idx.device)
    out = torch.empty(33, device=idx.device, dtype=torch.float64)
    fn, name = ((lib.pi3_sim3_umeyama_weighted, "pi3_sim3_umeyama_weighted") if torch.float32 in kinds
                else (lib.pi3_sim3_umeyama, "pi3_sim3_umeyama"))
    rc = fn(pts_ref.data_ptr(), pts_qry.data_ptr(), idx.data_ptr(), _L.ptr(w_ref), _L.ptr(w_qry),
            ov, K, last_ref_pose.data_ptr(), flags, out.data_ptr(), _L.stream_ptr())
    _L.check(rc, name)
    return out


def sim3_apply(M4: torch.Tensor, pts: Optional[torch.Tensor], poses: Optional[torch.Tensor]) -> None:
    lib = _L.load()
    assert M4.dtype == torch.float64 and M4.numel() >= 16
    n = 0 if pts is None else pts.numel() // 3
    F = 0 if poses is None else poses.numel() // 16
    rc = lib.pi3_sim3_apply(M4.data_ptr(), _L.ptr(pts), n, _L.ptr(poses), F, _L.stream_ptr())
    _L.check(rc, "pi3_sim3_apply")


def sim3_compose_prefix(T: torch.Tensor) -> torch.Tensor:
    lib = _L.load()
    assert T.dtype == torch.float64 and T.is_contiguous()
    n = T.numel() // 16
    G = torch.empty_like(T)
    rc = lib.pi3_sim3_compose_prefix(T.data_ptr(), G.data_ptr(), n, _L.stream_ptr())
    _L.check(rc, "pi3_sim3_compose_prefix")
    return G


def focal_shift(local_points: torch.Tensor, conf: Optional[torch.Tensor], uvx: torch.Tensor, uvy: torch.Tensor,
                conf_thr: float = 0.1, mask: Optional[torch.Tensor] = None):
    """local_points [F,H,W,3], conf [F,H,W,(1)] f32 (or mask uint8 [F,H,W]) -> dict(focal [F], shift [F],
    fxfycxcy [F,4], intrinsics [F,3,3])."""
    lib = _L.load()
    F, H, W = local_points.shape[:3]
    assert local_points.is_contiguous() and uvx.numel() == W and uvy.numel() == H
    assert (conf is not None and conf.is_contiguous()) or (mask is not None and mask.is_contiguous())
    dev = local_points.device
    focal = torch.empty(F, device=dev, dtype=torch.float32)
    shift = torch.empty(F, device=dev, dtype=torch.float32)
    fxy = torch.empty(F, 4, device=dev, dtype=torch.float32)
    K = torch.empty(F, 3, 3, device=dev, dtype=torch.float32)
    rc = lib.pi3_focal_shift(local_points.data_ptr(), _L.ptr(conf), _L.ptr(mask), uvx.data_ptr(), uvy.data_ptr(), F, H, W,
                             float(conf_thr), focal.data_ptr(), shift.data_ptr(), fxy.data_ptr(), K.data_ptr(),
                             _L.stream_ptr())
    _L.check(rc, "pi3_focal_shift")
    return dict(focal=focal, shift=shift, fxfycxcy=fxy, intrinsics=K)


# ----------------------------------------------------------------------------------------------- MoGe conv pyramid
def conv3x3(img: torch.Tensor, H: int, W: int, C: int, wgt: torch.Tensor, bias: Optional[torch.Tensor],
            out: torch.Tensor, resid: Optional[torch.Tensor] = None, act: int = ACT_NONE) -> torch.Tensor:
    """img bf16 NHWC [H*W, ldc] (one image), wgt bf16 [N, 9*C] (C % 64 == 0) or [N, 10*32] (C == 32: ten tap slots, the
    tenth zero); out [H*W, >=N] f32/bf16."""
    lib = _L.load()
    assert img.dtype in _16BIT and wgt.dtype == img.dtype and wgt.shape[1] == (320 if C == 32 else 9 * C)
    rc = lib.pi3_conv3x3(img.data_ptr(), img.stride(0), 1, H, W, C, wgt.data_ptr(), wgt.shape[0], _L.ptr(bias),
                         _L.ptr(resid), resid.stride(0) if resid is not None else 0, out.data_ptr(), out.stride(0),
                         _dt(img), _dt(out), act, _L.stream_ptr())
    _L.check(rc, "pi3_conv3x3")
    return out


def groupnorm_stats(x: torch.Tensor, HW: int, C: int, G: int, stats: torch.Tensor) -> None:
    lib = _L.load()
    assert x.dtype == torch.float32 and stats.dtype == torch.float64 and stats.numel() >= 2 * G
    n_ws = int(lib.pi3_groupnorm_ws_doubles(1, HW, C))
    ws = torch.empty(n_ws, device=x.device, dtype=torch.float64)     # per-block channel partials (deterministic sum)
    rc = lib.pi3_groupnorm_stats(x.data_ptr(), x.stride(0), 1, HW, C, G, stats.data_ptr(), ws.data_ptr(), n_ws,
                                 _L.stream_ptr())
    _L.check(rc, "pi3_groupnorm_stats")


ACT_LEAKY, ACT_SILU, ACT_ELU = 3, 4, 5


def groupnorm_apply(x, HW, C, Cpad, G, stats, gamma, beta, eps, act, out) -> None:
    """G = 0: no normalisation; gamma / beta None: no affine (InstanceNorm2d)."""
    lib = _L.load()
    assert out.dtype in _16BIT and out.shape[1] >= Cpad
    rc = lib.pi3_groupnorm_apply(x.data_ptr(), x.stride(0), 1, HW, C, Cpad, G, _L.ptr(stats), _L.ptr(gamma),
                                 _L.ptr(beta), float(eps), act, out.data_ptr(), out.stride(0), _dt(out), _L.stream_ptr())
    _L.check(rc, "pi3_groupnorm_apply")


def add_rows(x: torch.Tensor, y: torch.Tensor, rows: int, C: int) -> None:
    lib = _L.load()
    assert x.dtype == torch.float32 and y.dtype == torch.float32
    rc = lib.pi3_add_rows(x.data_ptr(), x.stride(0), y.data_ptr(), y.stride(0), rows, C, _L.stream_ptr())
    _L.check(rc, "pi3_add_rows")


def convt_scatter(g: torch.Tensor, H: int, W: int, Cout: int, Cs: int, Cpad: int, out: torch.Tensor) -> None:
    lib = _L.load()
    assert g.dtype == torch.float32 and out.dtype in _16BIT
    rc = lib.pi3_convt_scatter(g.data_ptr(), g.stride(0), 1, H, W, Cout, Cs, Cpad, out.data_ptr(), out.stride(0),
                               _dt(out), _L.stream_ptr())
    _L.check(rc, "pi3_convt_scatter")


def uv_affine(x, H, W, C, w, wofs, bias, uvx, uvy, accumulate: bool) -> None:
    lib = _L.load()
    assert x.dtype == torch.float32 and w.dtype == torch.float32 and w.stride(1) == 1
    rc = lib.pi3_uv_affine(x.data_ptr(), x.stride(0), 1, H, W, C, w.data_ptr(), w.stride(0), wofs, _L.ptr(bias),
                           uvx.data_ptr(), uvy.data_ptr(), int(accumulate), _L.stream_ptr())
    _L.check(rc, "pi3_uv_affine")


def resize_taps(src, sstr, C, ys, yw, xs, xw, oh, ow, dst, dstr) -> None:
    lib = _L.load()
    assert src.dtype == torch.float32 and dst.dtype == torch.float32 and ys.dtype == torch.int32
    rc = lib.pi3_resize_taps(src.data_ptr(), sstr[0], sstr[1], sstr[2], C, ys.data_ptr(), yw.data_ptr(),
                             xs.data_ptr(), xw.data_ptr(), oh, ow, dst.data_ptr(), dstr[0], dstr[1], dstr[2],
                             _L.stream_ptr())
    _L.check(rc, "pi3_resize_taps")


def dense_vec(x, W, b, act, y) -> None:
    lib = _L.load()
    assert W.dtype == torch.float32 and W.is_contiguous() and x.is_contiguous()
    rc = lib.pi3_dense_vec(x.data_ptr(), W.data_ptr(), _L.ptr(b), W.shape[1], W.shape[0], act, y.data_ptr(),
                           _L.stream_ptr())
    _L.check(rc, "pi3_dense_vec")


def moge_remap(pts, mask_logit, n, remap, mask) -> None:
    lib = _L.load()
    rc = lib.pi3_moge_remap(pts.data_ptr(), _L.ptr(mask_logit), n, remap, mask.data_ptr(), _L.stream_ptr())
    _L.check(rc, "pi3_moge_remap")


def moge_depth(pts, shift, log_scale, mask, n, depth) -> None:
    lib = _L.load()
    rc = lib.pi3_moge_depth(pts.data_ptr(), shift.data_ptr(), _L.ptr(log_scale), mask.data_ptr(), n,
                            depth.data_ptr(), _L.stream_ptr())
    _L.check(rc, "pi3_moge_depth")


def project_observations(points: torch.Tensor, poses: torch.Tensor, intrinsics: torch.Tensor, W: int, H: int,
                         max_after: int):
    """points f16 [N,K,3], poses f32 [N,4,4], intrinsics f32 [N,3,3] -> (uv f32 [N,N,K,2], valid bool [N,N,K])."""
    lib = _L.load()
    N, K = points.shape[:2]
    assert points.dtype == torch.float16 and points.is_contiguous()
    poses, intrinsics = poses.float().contiguous(), intrinsics.float().contiguous()
    uv = torch.zeros(N, N, K, 2, device=points.device, dtype=torch.float32)
    valid = torch.empty(N, N, K, device=points.device, dtype=torch.uint8)
    rc = lib.pi3_project_observations(points.data_ptr(), poses.data_ptr(), intrinsics.data_ptr(), N, K, W, H,
                                      max_after, uv.data_ptr(), valid.data_ptr(), _L.stream_ptr())
    _L.check(rc, "pi3_project_observations")
    return uv, valid.bool()


def ingest_frames(frames_u8: torch.Tensor, H1: int, W1: int, xb: torch.Tensor, xk: torch.Tensor, yb: torch.Tensor,
                  yk: torch.Tensor) -> torch.Tensor:
    """uint8 [N,H0,W0,3] -> float32 [N,3,H1,W1]; bounds / coefs are int32 device tensors from image_io.resample_coeffs."""
    lib = _L.load()
    assert frames_u8.dtype == torch.uint8 and frames_u8.is_contiguous() and frames_u8.shape[-1] == 3
    N, H0, W0 = frames_u8.shape[:3]
    for t in (xb, xk, yb, yk):
        assert t.dtype == torch.int32 and t.is_contiguous()
    assert xb.shape == (W1, 2) and yb.shape == (H1, 2) and xk.shape[0] == W1 and yk.shape[0] == H1
    tmp = torch.empty(N, H0, W1, 3, device=frames_u8.device, dtype=torch.uint8)
    dst = torch.empty(N, 3, H1, W1, device=frames_u8.device, dtype=torch.float32)
    rc = lib.pi3_ingest_frames(frames_u8.data_ptr(), N, H0, W0, H1, W1, xb.data_ptr(), xk.data_ptr(), xk.shape[1],
                               yb.data_ptr(), yk.data_ptr(), yk.shape[1], tmp.data_ptr(), dst.data_ptr(),
                               _L.stream_ptr())
    _L.check(rc, "pi3_ingest_frames")
    return dst


def undistort_maps(params16, model: int, H: int, W: int, device):
    """params16: 16 python floats (see include/pi3slam_hip.h) -> (map_x, map_y) f32 [H,W] on `device`."""
    import ctypes
    lib = _L.load()
    arr = (ctypes.c_double * 16)(*[float(v) for v in params16])
    mx = torch.empty(H, W, device=device, dtype=torch.float32)
    my = torch.empty(H, W, device=device, dtype=torch.float32)
    rc = lib.pi3_undistort_maps(ctypes.cast(arr, ctypes.c_void_p), int(model), H, W, mx.data_ptr(), my.data_ptr(),
                                _L.stream_ptr())
    _L.check(rc, "pi3_undistort_maps")
    return mx, my


def remap_bilinear_u8(frames_u8: torch.Tensor, map_x: torch.Tensor, map_y: torch.Tensor) -> torch.Tensor:
    """uint8 [N,H0,W0,3] + f32 maps [H,W] -> float32 [N,3,H,W] (cv2.remap INTER_LINEAR, border 0, then / 255)."""
    lib = _L.load()
    assert frames_u8.dtype == torch.uint8 and frames_u8.is_contiguous() and frames_u8.shape[-1] == 3
    assert map_x.dtype == torch.float32 and map_y.dtype == torch.float32 and map_x.shape == map_y.shape
    N, H0, W0 = frames_u8.shape[:3]
    H, W = map_x.shape
    dst = torch.empty(N, 3, H, W, device=frames_u8.device, dtype=torch.float32)
    rc = lib.pi3_remap_bilinear_u8(frames_u8.data_ptr(), N, H0, W0, map_x.contiguous().data_ptr(),
                                   map_y.contiguous().data_ptr(), H, W, dst.data_ptr(), _L.stream_ptr())
    _L.check(rc, "pi3_remap_bilinear_u8")
    return dst


# ----------------------------------------------------------------------------------------------- bundle adjustment
def bundle_adjust(points: torch.Tensor, poses: torch.Tensor, intr: torch.Tensor, uv: torch.Tensor, valid: torch.Tensor,
                  huber_width: float, max_iters: int, prior_R: Optional[torch.Tensor] = None,
                  prior_C: Optional[torch.Tensor] = None, prior_flag: Optional[torch.Tensor] = None,
                  sqrt_info_rot: float = 0.0, sqrt_info_pos: float = 0.0, homogeneous: bool = False,
                  inverse_depth: bool = False) -> torch.Tensor:
    """points f64 [N*K,3] and poses f64 [N,12] (R world->camera | centre) refined in place; intr f64 [N,4];
    uv f32 [N,N,K,2], valid u8 [N,N,K].  -> summary f64 [11] (device).  homogeneous: Theia's default point
    parametrization (pi3_bundle_adjust_homogeneous) instead of Euclidean steps; inverse_depth: one inverse depth per track
    along its reference keypoint's ray (pi3_bundle_adjust_inverse_depth; points are snapped onto those rays)."""
    assert not (homogeneous and inverse_depth)
    lib = _L.load()
    N, _, K = valid.shape
    for t in (points, poses, intr):
        assert t.dtype == torch.float64 and t.is_contiguous()
    assert uv.dtype == torch.float32 and uv.is_contiguous() and valid.dtype == torch.uint8 and valid.is_contiguous()
    assert tuple(points.shape) == (N * K, 3) and tuple(poses.shape) == (N, 12) and tuple(intr.shape) == (N, 4)
    uvT = uv.permute(0, 2, 1, 3).contiguous()          # [source][keypoint][target]: a track's cameras are contiguous
    validT = valid.permute(0, 2, 1).contiguous()
    n_ws = int(lib.pi3_ba_workspace_doubles(N, K))
    ws = torch.empty(n_ws, device=points.device, dtype=torch.float64)
    summary = torch.empty(11, device=points.device, dtype=torch.float64)
    if prior_flag is not None:
        assert prior_flag.dtype == torch.uint8 and prior_R.dtype == torch.float64 and prior_C.dtype == torch.float64
        prior_R, prior_C, prior_flag = prior_R.contiguous(), prior_C.contiguous(), prior_flag.contiguous()
    fn, fn_name = ((lib.pi3_bundle_adjust_inverse_depth, "pi3_bundle_adjust_inverse_depth") if inverse_depth else
                   (lib.pi3_bundle_adjust_homogeneous, "pi3_bundle_adjust_homogeneous") if homogeneous else
                   (lib.pi3_bundle_adjust, "pi3_bundle_adjust"))
    rc = fn(points.data_ptr(), poses.data_ptr(), intr.data_ptr(), uv.data_ptr(), valid.data_ptr(),
            uvT.data_ptr(), validT.data_ptr(), N, K, float(huber_width), int(max_iters),
            _L.ptr(prior_R), _L.ptr(prior_C), _L.ptr(prior_flag), float(sqrt_info_rot),
            float(sqrt_info_pos), summary.data_ptr(), ws.data_ptr(), n_ws, _L.stream_ptr())
    _L.check(rc, fn_name)
    return summary


def ba_outlier_tracks(points: torch.Tensor, poses: torch.Tensor, intr: torch.Tensor, uv: torch.Tensor,
                      valid: torch.Tensor, max_px: float, min_angle_deg: float) -> torch.Tensor:
    lib = _L.load()
    N, _, K = valid.shape
    est = torch.empty(N * K, device=points.device, dtype=torch.uint8)
    rc = lib.pi3_ba_outlier_tracks(points.data_ptr(), poses.data_ptr(), intr.data_ptr(), uv.data_ptr(), valid.data_ptr(),
                                   N, K, float(max_px), float(min_angle_deg), est.data_ptr(), _L.stream_ptr())
    _L.check(rc, "pi3_ba_outlier_tracks")
    return est.view(N, K).bool()


# ---------------------------------------------------------------------------------------------------- device guard
# Every wrapper launches on torch's CURRENT stream, i.e. on the current device.  A tensor that lives on another card
# (e.g. 'cuda:0' data in a rank bound to cuda:3) would hand that card's pointers to a kernel running elsewhere: fail
# loudly instead.  Checked on the first tensor argument of each call (all tensors of a call share a device by contract).
def _guarded(fn):
    import functools

    @functools.wraps(fn)
    def wrapper(*args, **kwargs):
        for a in args:
            if isinstance(a, torch.Tensor):
                if not a.is_cuda:
                    raise _L.Pi3HipError(f"{fn.__name__}: expected a device tensor, got {a.device}")
                if a.device.index != torch.cuda.current_device():
                    raise _L.Pi3HipError(
                        f"{fn.__name__}: tensor on {a.device} but the current device (and stream) is "
                        f"cuda:{torch.cuda.current_device()}; wrap the call in torch.cuda.device(t.device)")
                break
        return fn(*args, **kwargs)
    return wrapper


for _name, _fn in list(globals().items()):
    if isinstance(_fn, type(_guarded)) and getattr(_fn, "__module__", None) == __name__ and not _name.startswith("_") \
            and _name not in ("undistort_maps",):          # plain functions only (cuRoPE2D is a module class)
        globals()[_name] = _guarded(_fn)
del _name, _fn
