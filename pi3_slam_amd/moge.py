"""MoGeEngine: the MoGe-2 metric-depth forward used for the per-chunk metric scale
(slam/offline_chunk_creator.py:70-79, 182-187 — only infer(img)['depth'] is consumed).

Mirror of MoGeModel.from_pretrained / infer (moge/model/v2.py:66-97, 181-290) driven by the checkpoint's
`model_config` dict, with every arithmetic step in C-ABI kernels: the DINOv2 encoder reuses the pi3 transformer kernels
(vit.py), 3x3 replicate-padded convolutions run as implicit GEMMs on MFMA (pi3_conv3x3), 1x1 convolutions as pi3_gemm,
GroupNorm / transposed-conv scatter / UV planes / resizes / remap in csrc/moge.hip, focal-shift recovery in the same LM
kernel as the pi3 intrinsics.  Activations: NHWC fp32 [H*W, ld] + 16-bit NHWC staging images (channel stride % 64 == 0).

Supported config space (moge/model/modules.py:18-254, every option an UP-sampling ConvStack can carry): resamplers
conv_transpose / pixel_shuffle / nearest / bilinear; activations relu / leaky_relu / silu / elu; res-block norms
group_norm / layer_norm / instance_norm / none; dim_times_res_block_hidden >= 1; identity input and output blocks.  The
down-sampling resamplers (pixel_unshuffle / avg_pool / max_pool) cannot occur in the 5-level up-sampling pyramid of
MoGeModel.forward (v2.py:141-150) and raise NotImplementedError at construction.  The released
"Ruicheng/moge-2-vits-normal" checkpoint (weights and config) is not available offline; `from_pretrained("recipe")`
builds SYNTHETIC_CONFIG with recipe weights, and a local model.pt ({'model_config', 'model'}) is loaded as the reference
does (v2.py:80-95).  The reference runs this forward under fp16 autocast (v2.py:228); so does this engine since round 5:
IEEE-half operands on the f16 matrix-core forms with fp32 accumulation and fp32 maps between the layers (`dtype=torch.bfloat16`
selects the bf16 forms of rounds 1-4).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import numpy as np
import torch

from . import ops
from .recipe import fnv1a64, recipe_tensor
from .vit import run_block
from .weights import IMAGE_MEAN, IMAGE_STD

BACKBONES = {  # moge/model/dinov2/hub/backbones.py
    "dinov2_vits14": dict(dim=384, depth=12, heads=6, n_reg=0, antialias=False, offset=0.1),
    "dinov2_vitb14": dict(dim=768, depth=12, heads=12, n_reg=0, antialias=False, offset=0.1),
    "dinov2_vitl14": dict(dim=1024, depth=24, heads=16, n_reg=0, antialias=False, offset=0.1),
    "dinov2_vits14_reg": dict(dim=384, depth=12, heads=6, n_reg=4, antialias=True, offset=0.0),
    "dinov2_vitb14_reg": dict(dim=768, depth=12, heads=12, n_reg=4, antialias=True, offset=0.0),
    "dinov2_vitl14_reg": dict(dim=1024, depth=24, heads=16, n_reg=4, antialias=True, offset=0.0),
}

SYNTHETIC_CONFIG = dict(
    encoder=dict(backbone="dinov2_vits14", intermediate_layers=4, dim_out=384),
    neck=dict(dim_in=[386, 2, 2, 2, 2], dim_res_blocks=[384, 256, 128, 64, 32], dim_out=[256, 128, 64, 32, 32],
              resamplers=["conv_transpose"] * 4, num_res_blocks=[1, 1, 1, 1, 1]),
    points_head=dict(dim_in=[256, 128, 64, 32, 32], dim_res_blocks=[128, 64, 32, 32, 32],
                     dim_out=[None, None, None, None, 3], resamplers=["conv_transpose"] * 4,
                     num_res_blocks=[1, 1, 1, 1, 1]),
    mask_head=dict(dim_in=[256, 128, 64, 32, 32], dim_res_blocks=[64, 32, 32, 32, 32],
                   dim_out=[None, None, None, None, 1], resamplers=["conv_transpose"] * 4,
                   num_res_blocks=[1, 1, 1, 1, 1]),
    scale_head=dict(dims=[384, 128, 1]), remap_output="exp", num_tokens_range=[1200, 3600])

REMAP = {"linear": 0, "exp": 1, "sinh": 2, "sinh_exp": 3}


def _lst(v, n):
    return list(v) if isinstance(v, (list, tuple)) else [v] * n


def _up(x, m):
    return (x + m - 1) // m * m


def _npad(c):
    """Row stride of an fp32 NHWC map / output columns of a GEMM or convolution: the narrow-N kernels (gemm.hip:
    gemm_narrow_kernel) take any multiple of 32, so a 32-channel map of the finest pyramid levels is stored as 32-float
    rows, not padded to the 128-column tile of the big kernels (4x the bytes on ~900 k pixels)."""
    return _up(c, 32)


def _cpad(c):
    """Channel stride of a bf16 staging image that a 3x3 convolution reads: one 64-channel block per tap, or - for maps of
    up to 32 channels - 32 channels with two taps per K-step (pi3_conv3x3, C == 32)."""
    return 32 if c <= 32 else _up(c, 64)


ACTS = {"relu": 2, "leaky_relu": 3, "silu": 4, "elu": 5}          # ops.ACT_* codes of pi3_groupnorm_apply
NORMS = ("group_norm", "layer_norm", "instance_norm", "none")
UPSAMPLERS = ("conv_transpose", "pixel_shuffle", "nearest", "bilinear")


def _stack_options(cfg: Dict):
    act = cfg.get("activation", "relu")
    in_norm, hid_norm = cfg.get("res_block_in_norm", "layer_norm"), cfg.get("res_block_hidden_norm", "group_norm")
    mult = int(cfg.get("dim_times_res_block_hidden", 1))
    if act not in ACTS:
        raise NotImplementedError(f"activation '{act}' is not defined by the reference (modules.py:36-45)")
    for k in (in_norm, hid_norm):
        if k not in NORMS:
            raise NotImplementedError(f"norm '{k}' is not defined by the reference (modules.py:47-56)")
    if mult < 1:
        raise NotImplementedError("dim_times_res_block_hidden must be >= 1")
    return act, in_norm, hid_norm, mult


def stack_shapes(name: str, cfg: Dict) -> Dict[str, tuple]:
    """state_dict entries of one ConvStack (moge/model/modules.py:195-240)."""
    dims = cfg["dim_res_blocks"]
    n = len(dims)
    dim_in, dim_out = _lst(cfg["dim_in"], n), _lst(cfg["dim_out"], n)
    res = _lst(cfg["resamplers"], n - 1)
    nres = cfg.get("num_res_blocks", 1)
    _, in_norm, hid_norm, mult = _stack_options(cfg)
    for r in res:
        if r not in UPSAMPLERS:
            raise NotImplementedError(f"resampler '{r}' does not up-sample: it cannot occur in MoGeModel's pyramid")
    out: Dict[str, tuple] = {}
    for i in range(n):
        if dim_in[i] is not None:
            out[f"{name}.input_blocks.{i}.weight"] = (dims[i], dim_in[i], 1, 1)
            out[f"{name}.input_blocks.{i}.bias"] = (dims[i],)
    for i in range(n - 1):
        p, C, Cn = f"{name}.resamplers.{i}", dims[i], dims[i + 1]
        if res[i] == "conv_transpose":      # ConvTranspose2d(C, Cn, 2, 2) ; Conv2d(Cn, Cn, 3)      (modules.py:159-164)
            out[f"{p}.0.weight"], out[f"{p}.0.bias"] = (C, Cn, 2, 2), (Cn,)
            out[f"{p}.1.weight"], out[f"{p}.1.bias"] = (Cn, Cn, 3, 3), (Cn,)
        elif res[i] == "pixel_shuffle":     # Conv2d(C, 4 Cn, 3) ; PixelShuffle(2) ; Conv2d(Cn, Cn, 3)  (:146-153)
            out[f"{p}.0.weight"], out[f"{p}.0.bias"] = (4 * Cn, C, 3, 3), (4 * Cn,)
            out[f"{p}.2.weight"], out[f"{p}.2.bias"] = (Cn, Cn, 3, 3), (Cn,)
        else:                               # Upsample(2, nearest | bilinear) ; Conv2d(C, Cn, 3)        (:154-158)
            out[f"{p}.1.weight"], out[f"{p}.1.bias"] = (Cn, C, 3, 3), (Cn,)
    for i in range(n):
        hid = mult * dims[i]
        for j in range(nres[i] if isinstance(nres, list) else nres):
            p = f"{name}.res_blocks.{i}.{j}.layers"
            if in_norm in ("group_norm", "layer_norm"):
                out[f"{p}.0.weight"], out[f"{p}.0.bias"] = (dims[i],), (dims[i],)
            if hid_norm in ("group_norm", "layer_norm"):
                out[f"{p}.3.weight"], out[f"{p}.3.bias"] = (hid,), (hid,)
            out[f"{p}.2.weight"], out[f"{p}.2.bias"] = (hid, dims[i], 3, 3), (hid,)
            out[f"{p}.5.weight"], out[f"{p}.5.bias"] = (dims[i], hid, 3, 3), (dims[i],)
    for i in range(n):
        if dim_out[i] is not None:
            out[f"{name}.output_blocks.{i}.weight"] = (dim_out[i], dims[i], 1, 1)
            out[f"{name}.output_blocks.{i}.bias"] = (dim_out[i],)
    return out


def moge_param_shapes(cfg: Dict) -> Dict[str, tuple]:
    bb = BACKBONES[cfg["encoder"]["backbone"]]
    D = bb["dim"]
    pre = "encoder.backbone"
    out: Dict[str, tuple] = {f"{pre}.cls_token": (1, 1, D), f"{pre}.pos_embed": (1, 1370, D), f"{pre}.mask_token": (1, D)}
    if bb["n_reg"]:
        out[f"{pre}.register_tokens"] = (1, bb["n_reg"], D)
    out[f"{pre}.patch_embed.proj.weight"] = (D, 3, 14, 14)
    out[f"{pre}.patch_embed.proj.bias"] = (D,)
    for i in range(bb["depth"]):
        b = f"{pre}.blocks.{i}"
        out.update({f"{b}.norm1.weight": (D,), f"{b}.norm1.bias": (D,), f"{b}.attn.qkv.weight": (3 * D, D),
                    f"{b}.attn.qkv.bias": (3 * D,), f"{b}.attn.proj.weight": (D, D), f"{b}.attn.proj.bias": (D,),
                    f"{b}.ls1.gamma": (D,), f"{b}.norm2.weight": (D,), f"{b}.norm2.bias": (D,),
                    f"{b}.mlp.fc1.weight": (4 * D, D), f"{b}.mlp.fc1.bias": (4 * D,), f"{b}.mlp.fc2.weight": (D, 4 * D),
                    f"{b}.mlp.fc2.bias": (D,), f"{b}.ls2.gamma": (D,)})
    out[f"{pre}.norm.weight"] = (D,)
    out[f"{pre}.norm.bias"] = (D,)
    n_int = cfg["encoder"]["intermediate_layers"]
    n_int = n_int if isinstance(n_int, int) else len(n_int)
    for i in range(n_int):
        out[f"encoder.output_projections.{i}.weight"] = (cfg["encoder"]["dim_out"], D, 1, 1)
        out[f"encoder.output_projections.{i}.bias"] = (cfg["encoder"]["dim_out"],)
    for head in ("neck", "points_head", "mask_head"):
        if cfg.get(head):
            out.update(stack_shapes(head, cfg[head]))
    if cfg.get("scale_head"):
        dims = cfg["scale_head"]["dims"]
        for li in range(len(dims) - 1):
            out[f"scale_head.{2 * li}.weight"] = (dims[li + 1], dims[li])
            out[f"scale_head.{2 * li}.bias"] = (dims[li + 1],)
    return out


def moge_recipe_params(name: str, shape) -> tuple:
    """(offset, scale) of the recipe for a MoGe parameter (see recipe.py); mask logits are biased positive so the
    synthetic model produces a usable mask."""
    leaf = name.split(".")[-1]
    if name == "mask_head.output_blocks.4.bias" or (name.startswith("mask_head.output_blocks") and leaf == "bias"):
        return 1.5, 0.1
    if leaf == "gamma":
        return 0.15, 0.05
    if leaf in ("cls_token", "pos_embed", "mask_token", "register_tokens"):
        return 0.0, 0.05
    if leaf == "bias":
        return (0.0, 0.05) if len(shape) == 1 and (".norm" in name or ".layers.0." in name or ".layers.3." in name) \
            else (0.0, 0.02)
    if len(shape) == 1:
        return 1.0, 0.1
    if ".resamplers." in name and name.endswith(".0.weight") and shape[2] == 2:
        fan_in = shape[0]                       # ConvTranspose2d weight is [in, out, kh, kw]
    else:
        fan_in = int(np.prod(shape[1:]))
    gain = 0.06 if name.startswith("points_head.output_blocks") else (0.3 if name.startswith("scale_head") else 1.0)
    return 0.0, gain * math.sqrt(3.0 / fan_in)


def recipe_state_dict_cpu(cfg: Dict) -> Dict[str, torch.Tensor]:
    out = {}
    for name, shape in moge_param_shapes(cfg).items():
        off, sc = moge_recipe_params(name, shape)
        out[name] = torch.from_numpy(recipe_tensor("moge." + name, shape, off, sc))
    out["encoder.image_mean"] = torch.tensor(IMAGE_MEAN).view(1, 3, 1, 1)
    out["encoder.image_std"] = torch.tensor(IMAGE_STD).view(1, 3, 1, 1)
    return out


# ---------------------------------------------------------------------------------------------------- host tap tables
def bicubic_taps_dense(in_size: int, out_size: int, scale: float) -> np.ndarray:
    """[out, in] matrix of F.interpolate(mode='bicubic', antialias=False, align_corners=False): cubic convolution
    A = -0.75 on src = scale*(i+0.5)-0.5, neighbours clamped (ATen upsample_bicubic2d)."""
    A = -0.75
    W = np.zeros((out_size, in_size), dtype=np.float32)
    for i in range(out_size):
        real = np.float32(scale) * np.float32(i + 0.5) - np.float32(0.5)
        ix = int(np.floor(real))
        t = np.float32(real - ix)

        def c1(x):
            return ((A + 2) * x - (A + 3)) * x * x + 1

        def c2(x):
            return ((A * x - 5 * A) * x + 8 * A) * x - 4 * A

        ws = [c2(t + 1.0), c1(t), c1(1.0 - t), c2(2.0 - t)]
        for k in range(4):
            j = min(max(ix - 1 + k, 0), in_size - 1)
            W[i, j] += np.float32(ws[k])
    return W


def linear_taps(in_size: int, out_size: int, antialias: bool, max_taps: int = 8):
    """(start/count int32 [out, 2], weights f32 [out, max_taps]) of F.interpolate(mode='bilinear',
    align_corners=False) with or without antialias along one axis."""
    sc = np.zeros((out_size, 2), dtype=np.int32)
    w = np.zeros((out_size, max_taps), dtype=np.float32)
    scale = np.float32(in_size) / np.float32(out_size)
    for i in range(out_size):
        if antialias:
            support = scale if scale >= 1.0 else np.float32(1.0)
            invscale = np.float32(1.0) / scale if scale >= 1.0 else np.float32(1.0)
            center = scale * np.float32(i + 0.5)
            xmin = max(0, int(center - support + np.float32(0.5)))
            xmax = min(in_size, int(center + support + np.float32(0.5)))
            j = np.arange(xmin, xmax, dtype=np.float32)
            ww = np.maximum(0.0, 1.0 - np.abs((j - center + np.float32(0.5)) * invscale)).astype(np.float32)
            tot = np.float32(ww.sum(dtype=np.float32))
            if tot != 0:
                ww = ww / tot
            assert len(ww) <= max_taps, "antialias support exceeds the tap table"
            sc[i] = (xmin, len(ww))
            w[i, :len(ww)] = ww
        else:
            src = max(scale * np.float32(i + 0.5) - np.float32(0.5), np.float32(0.0))
            i0 = min(int(src), in_size - 1)
            i1 = min(i0 + 1, in_size - 1)
            l1 = np.float32(src - i0)
            sc[i] = (i0, 2 if i1 != i0 else 1)
            if i1 != i0:
                w[i, 0], w[i, 1] = np.float32(1.0) - l1, l1
            else:
                w[i, 0] = 1.0
    return sc, w


class _Act:
    """NHWC fp32 activation map: tensor [H*W, ld] with C valid channels."""
    __slots__ = ("t", "C", "H", "W")

    def __init__(self, t, C, H, W):
        self.t, self.C, self.H, self.W = t, C, H, W


class MoGeEngine:
    def __init__(self, cfg: Dict, device: str = "cuda:0", state_dict: Optional[Dict[str, torch.Tensor]] = None,
                 dtype: torch.dtype = torch.float16):
        """dtype: the 16-bit format of the matrix-core operands (weights, staging images, q / k / v).  torch.float16 is
        what the reference computes in - MoGeModel.infer runs its forward under fp16 autocast (moge/model/v2.py:228) -
        and the default: v_mfma_*_f16 at the bf16 rate, fp32 accumulation, fp32 maps between the layers (where the
        reference's own are fp16).  torch.bfloat16 (rounds 1-4) stays selectable: 3 mantissa bits fewer, 8x the deviation."""
        assert dtype in (torch.float16, torch.bfloat16)
        self.dt16 = dtype
        self.cfg = cfg
        self.device = torch.device(device)
        torch.cuda.set_device(self.device)
        self.bb = BACKBONES[cfg["encoder"]["backbone"]]
        self.w: Dict[str, torch.Tensor] = {}
        shapes = moge_param_shapes(cfg)
        # first convolution of a pixel_shuffle resampler: its output channels are stored (dy, dx, co)-major so that the
        # PixelShuffle is the same scatter as the transposed convolution's (csrc/moge.hip: pi3_convt_scatter)
        self._ps_convs = set()
        for head in ("neck", "points_head", "mask_head", "normal_head"):
            if cfg.get(head):
                n = len(cfg[head]["dim_res_blocks"])
                for i, r in enumerate(_lst(cfg[head]["resamplers"], n - 1)):
                    if r == "pixel_shuffle":
                        self._ps_convs.add(f"{head}.resamplers.{i}.0")
        for name, shape in shapes.items():
            if state_dict is None:
                off, sc = moge_recipe_params(name, shape)
                t = torch.empty(shape, device=self.device, dtype=torch.float32)
                ops.recipe_fill(t, fnv1a64("moge." + name), off, sc)
            else:
                t = state_dict[name].to(self.device, torch.float32)
                assert tuple(t.shape) == tuple(shape), (name, tuple(t.shape), shape)
            self._install(name, t)
        self._consts = {}

    @classmethod
    def from_pretrained(cls, path: str, device: str = "cuda:0", dtype: torch.dtype = torch.float16) -> "MoGeEngine":
        """path == "recipe": synthetic config + recipe weights; otherwise a local model.pt like the reference's
        MoGeModel.from_pretrained (v2.py:80-95): {'model_config': dict, 'model': state_dict}."""
        if path == "recipe":
            return cls(SYNTHETIC_CONFIG, device, dtype=dtype)
        ckpt = torch.load(path, map_location="cpu", weights_only=True)
        return cls(ckpt["model_config"], device, ckpt["model"], dtype=dtype)

    # ------------------------------------------------------------------ weight layouts
    def _install(self, name: str, t: torch.Tensor) -> None:
        dev, bf = self.device, self.dt16
        if name.endswith("patch_embed.proj.weight"):
            D = t.shape[0]
            w = torch.zeros(D, 640, device=dev, dtype=bf)
            w[:, :588] = t.reshape(D, 588).to(bf)
            self.w[name] = w
        elif ".backbone.blocks." in name and name.endswith(("qkv.weight", "proj.weight", "fc1.weight", "fc2.weight")):
            self.w[name] = t.to(bf).contiguous()
        elif t.dim() == 4 and t.shape[2] == 3:            # 3x3 conv [Co, Ci, 3, 3] -> [Co_pad32, 3*3*Ci_pad64] / [Co_pad32, 10*32]
            if name.rsplit(".", 1)[0] in self._ps_convs:  # PixelShuffle(2): channel co*4 + q -> row q*Cn + co
                Cn = t.shape[0] // 4
                t = t.reshape(Cn, 4, *t.shape[1:]).transpose(0, 1).reshape(4 * Cn, *t.shape[1:])
            Co, Ci = t.shape[:2]
            if _cpad(Ci) == 32:       # ten tap slots of 32 channels (two taps per K-step), the tenth stays zero
                w = torch.zeros(_npad(Co), 10, 32, device=dev, dtype=bf)
                w[:Co, :9, :Ci] = t.permute(0, 2, 3, 1).reshape(Co, 9, Ci).to(bf)
            else:
                w = torch.zeros(_npad(Co), 3, 3, _up(Ci, 64), device=dev, dtype=bf)
                w[:Co, :, :, :Ci] = t.permute(0, 2, 3, 1).to(bf)
            self.w[name] = w.reshape(w.shape[0], -1).contiguous()
        elif t.dim() == 4 and t.shape[2] == 2:            # ConvTranspose2d [Ci, Co, 2, 2] -> [(dy*2+dx)*Co + co, Ci_pad64]
            Ci, Co = t.shape[:2]
            assert (4 * Co) % 32 == 0, "conv_transpose output channels must be a multiple of 8"
            w = torch.zeros(4 * Co, _up(Ci, 64), device=dev, dtype=bf)
            w[:, :Ci] = t.permute(2, 3, 1, 0).reshape(4 * Co, Ci).to(bf)
            self.w[name] = w.contiguous()
        elif t.dim() == 4:                                # 1x1 conv [Co, Ci, 1, 1]
            Co, Ci = t.shape[:2]
            self.w[name + "#f32"] = t.reshape(Co, Ci).contiguous()     # UV columns are applied in fp32
            w = torch.zeros(_npad(Co), _up(Ci, 64), device=dev, dtype=bf)
            w[:Co, :Ci] = t.reshape(Co, Ci).to(bf)
            self.w[name] = w.contiguous()
        elif ".resamplers." in name and name.endswith(".0.bias") and name.rsplit(".", 1)[0] in self._ps_convs:
            Cn = t.shape[0] // 4
            self.w[name] = t.reshape(Cn, 4).t().reshape(-1).contiguous()      # (dy, dx, co)-major like the weight rows
        elif ".resamplers." in name and name.endswith(".0.bias") and (name.rsplit(".", 1)[0] + ".weight") in self.w \
                and self.w[name.rsplit(".", 1)[0] + ".weight"].shape[0] == 4 * t.shape[0]:
            self.w[name] = t.repeat(4).contiguous()       # ConvTranspose2d: one bias per output channel, per (dy, dx)
        elif name.endswith(".bias") and (".input_blocks." in name or ".output_blocks." in name
                                         or ".res_blocks." in name and name.split(".")[-2] in ("2", "5")
                                         or ".resamplers." in name
                                         or name.startswith("encoder.output_projections")):
            b = torch.zeros(_npad(t.shape[0]), device=dev)
            b[: t.shape[0]] = t
            self.w[name] = b
        else:
            self.w[name] = t.contiguous()

    # ------------------------------------------------------------------ small helpers
    def _new(self, rows: int, cols: int, dtype=torch.float32) -> torch.Tensor:
        return torch.empty(rows, cols, device=self.device, dtype=dtype)

    def _to_bf16(self, a: _Act, conv: bool = False) -> torch.Tensor:
        """bf16 NHWC staging copy for a 1x1 convolution (channel stride = C rounded up to 64, the GEMM's K-step) or, with
        conv=True, for a 3x3 one (_cpad).  The fp32 maps keep exact zeros in their padded columns (zero-padded weights and
        biases); columns beyond the map's row are written as zeros."""
        Cp = _cpad(a.C) if conv else _up(a.C, 64)
        out = self._new(a.H * a.W, Cp, self.dt16)
        ops.cast_rows(a.t, out, rows=a.H * a.W, cols=Cp, in_cols=min(Cp, a.t.shape[1]))
        return out

    def _uv(self, H: int, W: int, ar: float):
        key = ("uv", H, W, round(ar, 9))
        if key not in self._consts:
            sx = ar / (1 + ar ** 2) ** 0.5
            sy = 1 / (1 + ar ** 2) ** 0.5
            u = torch.linspace(-sx * (W - 1) / W, sx * (W - 1) / W, W, dtype=torch.float32)
            v = torch.linspace(-sy * (H - 1) / H, sy * (H - 1) / H, H, dtype=torch.float32)
            self._consts[key] = (u.to(self.device), v.to(self.device))
        return self._consts[key]

    def _taps(self, n_in: int, n_out: int, antialias: bool):
        key = ("taps", n_in, n_out, antialias)
        if key not in self._consts:
            sc, w = linear_taps(n_in, n_out, antialias)
            self._consts[key] = (torch.from_numpy(sc).to(self.device), torch.from_numpy(w).to(self.device))
        return self._consts[key]

    # ------------------------------------------------------------------ ConvStack (modules.py:242-254)
    def _norm_act(self, src: _Act, key: str, norm: str, act: int) -> torch.Tensor:
        """norm + activation of a map -> bf16 NHWC staging image for the following 3x3 convolution."""
        HW, C = src.H * src.W, src.C
        Cp = _cpad(C)
        a = self._new(HW, Cp, self.dt16)
        if norm == "none":
            ops.groupnorm_apply(src.t, HW, C, Cp, 0, None, None, None, 1e-5, act, a)
            return a
        G = {"group_norm": C // 32, "layer_norm": 1, "instance_norm": C}[norm]
        stats = torch.empty(2 * G, device=self.device, dtype=torch.float64)
        ops.groupnorm_stats(src.t, HW, C, G, stats)
        affine = norm != "instance_norm"          # nn.InstanceNorm2d(C): affine=False, no parameters
        ops.groupnorm_apply(src.t, HW, C, Cp, G, stats, self.w[key + ".weight"] if affine else None,
                            self.w[key + ".bias"] if affine else None, 1e-5, act, a)
        return a

    def _res_block(self, p: str, x: _Act, in_norm: str, hid_norm: str, act: int, mult: int) -> None:
        """ResidualConvBlock (modules.py:18-68) with in == out channels: x += conv(act(norm(conv(act(norm(x))))))."""
        C, hid = x.C, mult * x.C
        a = self._norm_act(x, f"{p}.0", in_norm, act)
        h = _Act(self._new(x.H * x.W, _npad(hid)), hid, x.H, x.W)
        ops.conv3x3(a, x.H, x.W, _cpad(C), self.w[f"{p}.2.weight"], self.w[f"{p}.2.bias"], h.t)
        a = self._norm_act(h, f"{p}.3", hid_norm, act)
        ops.conv3x3(a, x.H, x.W, _cpad(hid), self.w[f"{p}.5.weight"], self.w[f"{p}.5.bias"], x.t, resid=x.t)

    def _resample(self, p: str, kind: str, x: _Act, Cn: int) -> _Act:
        """Resampler (modules.py:139-182), scale factor 2."""
        H, W, C = x.H, x.W, x.C
        if kind == "conv_transpose":          # ConvTranspose2d(k = s = 2) as a GEMM + scatter, then Conv2d 3x3
            g = self._new(H * W, 4 * Cn)
            ops.gemm(self._to_bf16(x), self.w[p + ".0.weight"], g, M=H * W, bias=self.w[p + ".0.bias"])
            up = self._new(4 * H * W, _cpad(Cn), self.dt16)
            ops.convt_scatter(g, H, W, Cn, Cn, _cpad(Cn), up)
            out = _Act(self._new(4 * H * W, _npad(Cn)), Cn, 2 * H, 2 * W)
            ops.conv3x3(up, 2 * H, 2 * W, _cpad(Cn), self.w[p + ".1.weight"], self.w[p + ".1.bias"], out.t)
            return out
        if kind == "pixel_shuffle":           # Conv2d(C, 4 Cn, 3) with (dy, dx, co)-major rows + the same scatter
            g = self._new(H * W, _npad(4 * Cn))
            ops.conv3x3(self._to_bf16(x, conv=True), H, W, _cpad(C), self.w[p + ".0.weight"], self.w[p + ".0.bias"], g)
            up = self._new(4 * H * W, _cpad(Cn), self.dt16)
            ops.convt_scatter(g, H, W, Cn, Cn, _cpad(Cn), up)
            out = _Act(self._new(4 * H * W, _npad(Cn)), Cn, 2 * H, 2 * W)
            ops.conv3x3(up, 2 * H, 2 * W, _cpad(Cn), self.w[p + ".2.weight"], self.w[p + ".2.bias"], out.t)
            return out
        # nn.Upsample(scale_factor=2, mode=nearest | bilinear(align_corners=False)) then Conv2d(C, Cn, 3)
        key = ("up2", kind, H, W)
        if key not in self._consts:
            def taps(n_in):
                if kind == "bilinear":
                    sc, wt = linear_taps(n_in, 2 * n_in, False)
                else:   # nearest: src = floor(dst * in / out) = dst // 2
                    sc = np.stack([np.arange(2 * n_in) // 2, np.ones(2 * n_in, dtype=np.int64)], 1).astype(np.int32)
                    wt = np.zeros((2 * n_in, 8), dtype=np.float32)
                    wt[:, 0] = 1.0
                return torch.from_numpy(sc).to(self.device), torch.from_numpy(wt).to(self.device)
            self._consts[key] = taps(H) + taps(W)
        ys, yw, xs, xw = self._consts[key]
        ld = x.t.shape[1]
        big = self._new(4 * H * W, ld)
        ops.resize_taps(x.t, (1, W * ld, ld), C, ys, yw, xs, xw, 2 * H, 2 * W, big, (1, 2 * W * ld, ld))
        a = self._new(4 * H * W, _cpad(C), self.dt16)
        ops.cast_rows(big, a, rows=4 * H * W, cols=_cpad(C), in_cols=C if C % 4 == 0 else _up(C, 4))
        out = _Act(self._new(4 * H * W, _npad(Cn)), Cn, 2 * H, 2 * W)
        ops.conv3x3(a, 2 * H, 2 * W, _cpad(C), self.w[p + ".1.weight"], self.w[p + ".1.bias"], out.t)
        return out

    def _conv_stack(self, name: str, cfg: Dict, feats: List[Optional[_Act]], uv_levels: bool, base_h: int, base_w: int,
                    ar: float, all_outputs: bool = False) -> List[Optional[_Act]]:
        """all_outputs: every level's output is consumed (the neck, v2.py:148); a head is read at its finest level only
        (v2.py:150-157)."""
        dims = cfg["dim_res_blocks"]
        n = len(dims)
        dim_in, dim_out = _lst(cfg["dim_in"], n), _lst(cfg["dim_out"], n)
        res = _lst(cfg["resamplers"], n - 1)
        nres = cfg.get("num_res_blocks", 1)
        act_name, in_norm, hid_norm, mult = _stack_options(cfg)
        act = ACTS[act_name]
        outs: List[_Act] = []
        x: Optional[_Act] = None
        for i in range(n):
            H, W = base_h * 2 ** i, base_w * 2 ** i
            C = dims[i]
            if i == 0:
                x = _Act(self._new(H * W, _npad(C)), C, H, W)
            f = feats[i]
            wk, bk = f"{name}.input_blocks.{i}.weight", f"{name}.input_blocks.{i}.bias"
            have_feat = False
            if dim_in[i] is not None and f is not None:
                a = self._to_bf16(f)
                ops.gemm(a, self.w[wk], x.t, M=H * W, K=a.shape[1], bias=self.w[bk], resid=x.t if i > 0 else None)
                have_feat = True
            elif dim_in[i] is None and f is not None:        # nn.Identity input block: x = feature / x = x + feature
                assert f.C == C and not uv_levels, "an identity input block needs a feature of the block's width"
                if i == 0:
                    x.t[:, :C].copy_(f.t[:, :C])
                    if x.t.shape[1] > C:
                        x.t[:, C:].zero_()
                else:
                    ops.add_rows(x.t, f.t, H * W, C)
                have_feat = True
            if uv_levels:
                u, v = self._uv(H, W, ar)
                wf = self.w[wk + "#f32"]
                ops.uv_affine(x.t, H, W, C, wf, wf.shape[1] - 2, None if have_feat else self.w[bk][:C].contiguous(),
                              u, v, accumulate=(i > 0 or have_feat))
            for j in range(nres[i] if isinstance(nres, list) else nres):
                self._res_block(f"{name}.res_blocks.{i}.{j}.layers", x, in_norm, hid_norm, act, mult)
            if not (all_outputs or i == n - 1):
                outs.append(None)
            elif dim_out[i] is not None:
                o = _Act(self._new(H * W, _npad(dim_out[i])), dim_out[i], H, W)
                ops.gemm(self._to_bf16(x), self.w[f"{name}.output_blocks.{i}.weight"], o.t, M=H * W,
                         bias=self.w[f"{name}.output_blocks.{i}.bias"])
                outs.append(o)
            else:                                             # nn.Identity output block: the map itself (a copy where
                outs.append(x if i == n - 1 else _Act(x.t.clone(), x.C, x.H, x.W))   # the next level replaces x)
            if i < n - 1:
                x = self._resample(f"{name}.resamplers.{i}", res[i], x, dims[i + 1])
        return outs

    # ------------------------------------------------------------------ forward / infer
    @torch.no_grad()
    def forward(self, image: torch.Tensor, num_tokens: int) -> Dict[str, torch.Tensor]:
        """MoGeModel.forward (v2.py:128-179) for one image (3, H, W) fp32 on the device."""
        cfg, bb, dev, w = self.cfg, self.bb, self.device, self.w
        _, H, W = image.shape
        ar = W / H
        bh, bw = int((num_tokens / ar) ** 0.5), int((num_tokens * ar) ** 0.5)
        D, heads, nreg = bb["dim"], bb["heads"], bb["n_reg"]
        P, T = bh * bw, 1 + nreg + bh * bw
        # --- DINOv2Encoder.forward (modules.py:120-136): antialiased bilinear resize to (14 bh, 14 bw)
        ys, yw = self._taps(H, 14 * bh, True)
        xs, xw = self._taps(W, 14 * bw, True)
        img14 = torch.empty(1, 3, 14 * bh, 14 * bw, device=dev)
        ops.resize_taps(image.contiguous(), (H * W, W, 1), 3, ys, yw, xs, xw, 14 * bh, 14 * bw, img14,
                        (14 * bh * 14 * bw, 14 * bw, 1))
        patches = self._new(P, 640, self.dt16)
        ops.patch_gather(img14, patches, IMAGE_MEAN, IMAGE_STD)
        key = ("pos", bh, bw)
        if key not in self._consts:
            pe = w["encoder.backbone.pos_embed"][0]
            M = int(math.sqrt(pe.shape[0] - 1))
            if bh == M and bw == M:
                pos_patch = pe[1:].contiguous()
            elif bb["offset"] > 0:      # scale_factor path (vision_transformer.py:205-211)
                wy = torch.from_numpy(bicubic_taps_dense(M, bh, 1.0 / (float(bh + bb["offset"]) / M))).to(dev)
                wx = torch.from_numpy(bicubic_taps_dense(M, bw, 1.0 / (float(bw + bb["offset"]) / M))).to(dev)
                pos_patch = ops.resample_grid(pe[1:].reshape(M, M, D).contiguous(), wy, wx).reshape(P, D)
            else:
                from .engine import bicubic_aa_taps
                wy, wx = (torch.from_numpy(bicubic_aa_taps(M, n)).to(dev) for n in (bh, bw))
                pos_patch = ops.resample_grid(pe[1:].reshape(M, M, D).contiguous(), wy, wx).reshape(P, D)
            special = [w["encoder.backbone.cls_token"][0] + pe[0:1]]
            if nreg:
                special.append(w["encoder.backbone.register_tokens"][0])
            self._consts[key] = (pos_patch.contiguous(), torch.cat(special, 0).contiguous())
        pos_patch, special = self._consts[key]
        x = self._new(T, D)
        ops.gemm(patches, w["encoder.backbone.patch_embed.proj.weight"], x, M=P,
                 bias=w["encoder.backbone.patch_embed.proj.bias"], rpg=P, gstride=T, goff=1 + nreg, addtab=pos_patch)
        ops.fill_tokens(x, 1, T, 0, special)
        bufs = (self._new(T, D, self.dt16), self._new(T, 3 * D, self.dt16), self._new(T, D, self.dt16),
                self._new(T, 4 * D, self.dt16), self._new(1, heads, torch.float32).view(-1))
        n_int = cfg["encoder"]["intermediate_layers"]
        take = list(range(bb["depth"] - n_int, bb["depth"])) if isinstance(n_int, int) else list(n_int)
        Cenc = cfg["encoder"]["dim_out"]
        feat = _Act(self._new(P, _npad(Cenc)), Cenc, bh, bw)
        cls = self._new(1, D)
        k = 0
        for i in range(bb["depth"]):
            run_block(w, f"encoder.backbone.blocks.{i}", x, T, 1, T, T, heads, bufs, ls=True)
            if i in take:      # get_intermediate_layers(norm=True) + 1x1 output projection, summed (modules.py:127-131)
                ops.layernorm(x, w["encoder.backbone.norm.weight"], w["encoder.backbone.norm.bias"], bufs[0], 1e-6,
                              rows=T)
                ops.gemm(bufs[0][1 + nreg:], w[f"encoder.output_projections.{k}.weight"], feat.t, M=P,
                         bias=w[f"encoder.output_projections.{k}.bias"], resid=feat.t if k > 0 else None)
                if i == take[-1]:
                    ops.layernorm(x, w["encoder.backbone.norm.weight"], w["encoder.backbone.norm.bias"], cls, 1e-6,
                                  rows=1)
                k += 1
        # --- neck + heads
        feats = self._conv_stack("neck", cfg["neck"], [feat, None, None, None, None], True, bh, bw, ar, all_outputs=True)
        out: Dict[str, torch.Tensor] = {"_base": (bh, bw)}
        h4, w4 = bh * 16, bw * 16
        ys, yw = self._taps(h4, H, False)
        xs, xw = self._taps(w4, W, False)
        if cfg.get("points_head"):
            p = self._conv_stack("points_head", cfg["points_head"], feats, False, bh, bw, ar)[-1]
            pts = torch.empty(H, W, 3, device=dev)
            ld = p.t.shape[1]
            ops.resize_taps(p.t, (1, w4 * ld, ld), 3, ys, yw, xs, xw, H, W, pts, (1, 3 * W, 3))
            out["points_raw"] = pts
        if cfg.get("mask_head"):
            m = self._conv_stack("mask_head", cfg["mask_head"], feats, False, bh, bw, ar)[-1]
            ml = torch.empty(H, W, device=dev)
            ld = m.t.shape[1]
            ops.resize_taps(m.t, (1, w4 * ld, ld), 1, ys, yw, xs, xw, H, W, ml, (H * W, W, 1))
            out["mask_logit"] = ml
        if cfg.get("scale_head"):
            dims = cfg["scale_head"]["dims"]
            hcur = cls.reshape(-1)
            for li in range(len(dims) - 1):
                y = torch.empty(dims[li + 1], device=dev)
                ops.dense_vec(hcur, w[f"scale_head.{2 * li}.weight"], w[f"scale_head.{2 * li}.bias"],
                              ops.ACT_RELU if li < len(dims) - 2 else ops.ACT_NONE, y)
                hcur = y
            out["log_metric_scale"] = hcur
        return out

    @torch.no_grad()
    def infer_graphed(self, image: torch.Tensor, resolution_level: int = 9) -> Dict:
        """infer() replayed as one captured hipGraph per image shape.  MoGe is ~400 small single-image kernels that run
        while the GPU has nothing else to do (after the pi3 forward of the chunk): launched one by one they are
        launch-latency bound; as a graph they run back to back.  Same results bit for bit; the returned tensors are the
        graph's static outputs (valid until the next call)."""
        if image.dim() == 4:
            image = image[0]
        key = (tuple(image.shape), resolution_level)
        graphs = self.__dict__.setdefault("_graphs", {})
        if key not in graphs:
            static_in = image.to(self.device, torch.float32).contiguous().clone()
            self.infer(static_in, resolution_level=resolution_level)        # allocates the cached tables
            torch.cuda.synchronize(self.device)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, capture_error_mode="thread_local"):   # other threads may allocate meanwhile
                static_out = self.infer(static_in, resolution_level=resolution_level)
            graphs[key] = (graph, static_in, static_out)
        graph, static_in, static_out = graphs[key]
        static_in.copy_(image, non_blocking=True)
        graph.replay()
        return static_out

    @torch.no_grad()
    def infer(self, image: torch.Tensor, num_tokens: Optional[int] = None, resolution_level: int = 9) -> Dict:
        """MoGeModel.infer (v2.py:181-290) with the pipeline's defaults.  image (3, H, W) or (1, 3, H, W) fp32 in
        [0, 1].  Returns device tensors: depth (H, W) with +inf outside the mask, mask (H, W) bool, intrinsics (3, 3)."""
        if image.dim() == 4:
            assert image.shape[0] == 1, "one image per call"
            image = image[0]
        image = image.to(self.device, torch.float32).contiguous()
        _, H, W = image.shape
        ar = W / H
        if num_tokens is None:
            lo, hi = self.cfg.get("num_tokens_range", [1200, 3600])
            num_tokens = int(lo + (resolution_level / 9) * (hi - lo))
        out = self.forward(image, num_tokens)
        pts = out["points_raw"]
        mask = torch.empty(H, W, device=self.device, dtype=torch.uint8)
        ops.moge_remap(pts, out.get("mask_logit"), H * W, REMAP[self.cfg.get("remap_output", "linear")], mask)
        u, v = self._uv(H, W, ar)
        fs = ops.focal_shift(pts.view(1, H, W, 3), None, u, v, mask=mask.view(1, H, W))
        depth = torch.empty(H, W, device=self.device)
        mask_network = mask.clone()      # sigmoid(mask logit) > 0.5 alone; moge_depth also clears pixels with depth <= 0
        ops.moge_depth(pts, fs["shift"], out.get("log_metric_scale"), mask, H * W, depth)
        focal = fs["focal"]
        fx = focal / 2 * (1 + ar ** 2) ** 0.5 / ar
        fy = focal / 2 * (1 + ar ** 2) ** 0.5
        if "K_base" not in self._consts:   # the constant entries live on the device (no host scalar in a graph capture)
            self._consts["K_base"] = torch.tensor([[0.0, 0.0, 0.5], [0.0, 0.0, 0.5], [0.0, 0.0, 1.0]], device=self.device)
        K = self._consts["K_base"].clone()
        K[0, 0], K[1, 1] = fx[0], fy[0]
        return {"depth": depth, "mask": mask.bool(), "intrinsics": K, "points_affine": pts, "focal": focal[0],
                "shift": fs["shift"][0], "mask_network": mask_network.bool()}
