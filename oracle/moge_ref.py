"""ORACLE — test infrastructure only.  CPU fp32 restatement of the MoGe-2 forward/infer path that the pipeline uses for
metric scale (slam/offline_chunk_creator.py:182-187 reads only infer(...)['depth']).

Restates moge/model/v2.py:128-290 (MoGeModel.forward / infer), moge/model/modules.py:18-254 (ResidualConvBlock,
DINOv2Encoder, Resampler, MLP, ConvStack) and moge/model/dinov2/models/vision_transformer.py:187-245, 309-333 as plain
torch functional code driven by the checkpoint's `model_config` dict and state_dict.  Two utils3d functions that the
reference imports from an un-vendored third-party package are restated from their call sites (v2.py:253, 263).

Pinning: oracle/gen_golden_moge.py runs the REAL MoGeModel class (imported from /root/reference) with a synthetic
model_config + recipe weights; tests/test_oracle_golden.py checks this restatement against those vectors.  The real
checkpoint (weights AND model_config) of "Ruicheng/moge-2-vits-normal" is unavailable offline, so end-to-end parity with
the released model is UNPINNED; operator-level parity is pinned.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import torch
import torch.nn.functional as F

from . import post_ref

BACKBONES = {  # moge/model/dinov2/hub/backbones.py + models/vision_transformer.py:340-407
    "dinov2_vits14": dict(dim=384, depth=12, heads=6, n_reg=0, antialias=False, offset=0.1),
    "dinov2_vitb14": dict(dim=768, depth=12, heads=12, n_reg=0, antialias=False, offset=0.1),
    "dinov2_vitl14": dict(dim=1024, depth=24, heads=16, n_reg=0, antialias=False, offset=0.1),
    "dinov2_vits14_reg": dict(dim=384, depth=12, heads=6, n_reg=4, antialias=True, offset=0.0),
    "dinov2_vitb14_reg": dict(dim=768, depth=12, heads=12, n_reg=4, antialias=True, offset=0.0),
    "dinov2_vitl14_reg": dict(dim=1024, depth=24, heads=16, n_reg=4, antialias=True, offset=0.0),
}


def vit_block(sd, p, x, heads):
    """NestedTensorBlock.forward for a plain tensor (moge/model/dinov2/layers/block.py), pre-LN + LayerScale."""
    h = F.layer_norm(x, (x.shape[-1],), sd[f"{p}.norm1.weight"], sd[f"{p}.norm1.bias"], 1e-6)
    B, S, C = h.shape
    qkv = F.linear(h, sd[f"{p}.attn.qkv.weight"], sd[f"{p}.attn.qkv.bias"]).reshape(B, S, 3, heads, C // heads)
    q, k, v = qkv.permute(2, 0, 3, 1, 4)
    att = ((q @ k.transpose(-1, -2)) * (C // heads) ** -0.5).softmax(-1)
    h = F.linear((att @ v).transpose(1, 2).reshape(B, S, C), sd[f"{p}.attn.proj.weight"], sd[f"{p}.attn.proj.bias"])
    x = x + h * sd[f"{p}.ls1.gamma"]
    h = F.layer_norm(x, (C,), sd[f"{p}.norm2.weight"], sd[f"{p}.norm2.bias"], 1e-6)
    h = F.linear(F.gelu(F.linear(h, sd[f"{p}.mlp.fc1.weight"], sd[f"{p}.mlp.fc1.bias"])),
                 sd[f"{p}.mlp.fc2.weight"], sd[f"{p}.mlp.fc2.bias"])
    return x + h * sd[f"{p}.ls2.gamma"]


def interpolate_pos_encoding(pos_embed, h0, w0, antialias, offset):
    """moge/model/dinov2/models/vision_transformer.py:187-221."""
    N = pos_embed.shape[1] - 1
    M = int(math.sqrt(N))
    if h0 * w0 == N and h0 == w0:
        return pos_embed
    dim = pos_embed.shape[-1]
    kwargs = {}
    if offset > 0:
        kwargs["scale_factor"] = (float(h0 + offset) / M, float(w0 + offset) / M)
    else:
        kwargs["size"] = (h0, w0)
    patch = F.interpolate(pos_embed[:, 1:].reshape(1, M, M, dim).permute(0, 3, 1, 2), mode="bicubic",
                          antialias=antialias, **kwargs)
    assert (h0, w0) == tuple(patch.shape[-2:])
    patch = patch.permute(0, 2, 3, 1).flatten(1, 2)
    return torch.cat((pos_embed[:, :1], patch), dim=1)


def encoder_forward(sd, cfg_enc, image, rows, cols):
    """DINOv2Encoder.forward (modules.py:120-136) + get_intermediate_layers (vision_transformer.py:309-333)."""
    bb = BACKBONES[cfg_enc["backbone"]]
    pre = "encoder.backbone"
    img14 = F.interpolate(image, (rows * 14, cols * 14), mode="bilinear", align_corners=False, antialias=True)
    img14 = (img14 - sd["encoder.image_mean"]) / sd["encoder.image_std"]
    x = F.conv2d(img14, sd[f"{pre}.patch_embed.proj.weight"], sd[f"{pre}.patch_embed.proj.bias"], stride=14)
    x = x.flatten(2).transpose(1, 2)
    x = torch.cat((sd[f"{pre}.cls_token"].expand(x.shape[0], -1, -1), x), dim=1)
    x = x + interpolate_pos_encoding(sd[f"{pre}.pos_embed"], rows, cols, bb["antialias"], bb["offset"])
    if bb["n_reg"]:
        x = torch.cat((x[:, :1], sd[f"{pre}.register_tokens"].expand(x.shape[0], -1, -1), x[:, 1:]), dim=1)
    n = cfg_enc["intermediate_layers"]
    take = list(range(bb["depth"] - n, bb["depth"])) if isinstance(n, int) else list(n)
    outs = []
    for i in range(bb["depth"]):
        x = vit_block(sd, f"{pre}.blocks.{i}", x, bb["heads"])
        if i in take:
            outs.append(F.layer_norm(x, (x.shape[-1],), sd[f"{pre}.norm.weight"], sd[f"{pre}.norm.bias"], 1e-6))
    feats = [o[:, 1 + bb["n_reg"]:] for o in outs]
    cls = outs[-1][:, 0]
    acc = 0
    for i, f in enumerate(feats):
        fm = f.permute(0, 2, 1).unflatten(2, (rows, cols))
        acc = acc + F.conv2d(fm, sd[f"encoder.output_projections.{i}.weight"], sd[f"encoder.output_projections.{i}.bias"])
    return acc, cls


def _norm(sd, key, x, kind):
    C = x.shape[1]
    if kind == "group_norm":
        return F.group_norm(x, C // 32, sd[key + ".weight"], sd[key + ".bias"], 1e-5)
    if kind == "layer_norm":
        return F.group_norm(x, 1, sd[key + ".weight"], sd[key + ".bias"], 1e-5)
    if kind == "instance_norm":      # nn.InstanceNorm2d(C): no affine parameters, eps 1e-5
        return F.instance_norm(x, eps=1e-5)
    if kind == "none":
        return x
    raise NotImplementedError(kind)


def _act(x, kind):
    if kind == "relu":
        return F.relu(x)
    if kind == "leaky_relu":
        return F.leaky_relu(x, 0.2)
    if kind == "silu":
        return F.silu(x)
    if kind == "elu":
        return F.elu(x)
    raise NotImplementedError(kind)


def conv3(sd, key, x):
    return F.conv2d(F.pad(x, (1, 1, 1, 1), mode="replicate"), sd[key + ".weight"], sd[key + ".bias"])


def resample(sd, p, kind, x):
    """Resampler (modules.py:139-182), scale factor 2, the four up-sampling kinds."""
    if kind == "conv_transpose":
        return conv3(sd, p + ".1", F.conv_transpose2d(x, sd[p + ".0.weight"], sd[p + ".0.bias"], stride=2))
    if kind == "pixel_shuffle":
        return conv3(sd, p + ".2", F.pixel_shuffle(conv3(sd, p + ".0", x), 2))
    if kind in ("nearest", "bilinear"):
        up = F.interpolate(x, scale_factor=2, mode=kind, align_corners=False if kind == "bilinear" else None)
        return conv3(sd, p + ".1", up)
    raise NotImplementedError(kind)


def conv_stack(sd, name, cfg, in_features: List[Optional[torch.Tensor]]):
    """ConvStack.forward (modules.py:242-254) with ResidualConvBlock (:18-68) and Resampler (:139-182)."""
    dims = cfg["dim_res_blocks"]
    nres = cfg.get("num_res_blocks", 1)
    in_norm, hid_norm = cfg.get("res_block_in_norm", "layer_norm"), cfg.get("res_block_hidden_norm", "group_norm")
    act = cfg.get("activation", "relu")
    dim_in = cfg["dim_in"] if isinstance(cfg["dim_in"], (list, tuple)) else [cfg["dim_in"]] * len(dims)
    dim_out = cfg["dim_out"] if isinstance(cfg["dim_out"], (list, tuple)) else [cfg["dim_out"]] * len(dims)
    res = cfg["resamplers"] if isinstance(cfg["resamplers"], (list, tuple)) else [cfg["resamplers"]] * (len(dims) - 1)
    outs = []
    x = None
    for i in range(len(dims)):
        feat = in_features[i]
        if dim_in[i] is not None:
            feat = F.conv2d(feat, sd[f"{name}.input_blocks.{i}.weight"], sd[f"{name}.input_blocks.{i}.bias"])
        x = feat if i == 0 else (x + feat if feat is not None else x)
        for j in range(nres[i] if isinstance(nres, list) else nres):
            p = f"{name}.res_blocks.{i}.{j}.layers"
            h = conv3(sd, p + ".2", _act(_norm(sd, p + ".0", x, in_norm), act))
            h = conv3(sd, p + ".5", _act(_norm(sd, p + ".3", h, hid_norm), act))
            x = x + h
        o = x
        if dim_out[i] is not None:
            o = F.conv2d(x, sd[f"{name}.output_blocks.{i}.weight"], sd[f"{name}.output_blocks.{i}.bias"])
        outs.append(o)
        if i < len(dims) - 1:
            x = resample(sd, f"{name}.resamplers.{i}", res[i], x)
    return outs


def remap_points(points, mode):
    if mode == "linear":
        return points
    if mode == "sinh":
        return torch.sinh(points)
    xy, z = points.split([2, 1], dim=-1)
    if mode == "exp":
        z = torch.exp(z)
        return torch.cat([xy * z, z], dim=-1)
    if mode == "sinh_exp":
        return torch.cat([torch.sinh(xy), torch.exp(z)], dim=-1)
    raise ValueError(mode)


def uv_map(width, height, aspect_ratio):
    """utils/geometry_torch.py:39-51 with an explicit aspect ratio (v2.py:142)."""
    sx = aspect_ratio / (1 + aspect_ratio ** 2) ** 0.5
    sy = 1 / (1 + aspect_ratio ** 2) ** 0.5
    u = torch.linspace(-sx * (width - 1) / width, sx * (width - 1) / width, width, dtype=torch.float32)
    v = torch.linspace(-sy * (height - 1) / height, sy * (height - 1) / height, height, dtype=torch.float32)
    u, v = torch.meshgrid(u, v, indexing="xy")
    return torch.stack([u, v], dim=-1)


@torch.no_grad()
def moge_forward(sd, cfg, image: torch.Tensor, num_tokens: int) -> Dict[str, torch.Tensor]:
    """MoGeModel.forward (v2.py:128-179).  image (1, 3, H, W) fp32 in [0, 1]."""
    B, _, H, W = image.shape
    ar = W / H
    base_h, base_w = int((num_tokens / ar) ** 0.5), int((num_tokens * ar) ** 0.5)
    feat, cls = encoder_forward(sd, cfg["encoder"], image, base_h, base_w)
    feats = [feat, None, None, None, None]
    for lvl in range(5):
        uv = uv_map(base_w * 2 ** lvl, base_h * 2 ** lvl, ar).permute(2, 0, 1).unsqueeze(0).expand(B, -1, -1, -1)
        feats[lvl] = uv if feats[lvl] is None else torch.cat([feats[lvl], uv], dim=1)
    feats = conv_stack(sd, "neck", cfg["neck"], feats)
    out = {}
    if cfg.get("points_head"):
        p = conv_stack(sd, "points_head", cfg["points_head"], feats)[-1]
        p = F.interpolate(p, (H, W), mode="bilinear", align_corners=False, antialias=False)
        out["points"] = remap_points(p.permute(0, 2, 3, 1), cfg.get("remap_output", "linear"))
    if cfg.get("mask_head"):
        m = conv_stack(sd, "mask_head", cfg["mask_head"], feats)[-1]
        m = F.interpolate(m, (H, W), mode="bilinear", align_corners=False, antialias=False)
        out["mask"] = m.squeeze(1).sigmoid()
    if cfg.get("scale_head"):
        dims = cfg["scale_head"]["dims"]
        h = cls
        for li in range(len(dims) - 1):
            h = F.linear(h, sd[f"scale_head.{2 * li}.weight"], sd[f"scale_head.{2 * li}.bias"])
            if li < len(dims) - 2:
                h = F.relu(h)
        out["metric_scale"] = h.squeeze(1).exp()
    out["_base"] = (base_h, base_w)
    return out


@torch.no_grad()
def moge_infer(sd, cfg, image: torch.Tensor, resolution_level: int = 9) -> Dict[str, torch.Tensor]:
    """MoGeModel.infer (v2.py:181-290) with the defaults the pipeline uses (num_tokens=None, force_projection,
    apply_mask=True, fov_x=None), fp32.  image (3, H, W).  Returns depth (H, W), mask (H, W), intrinsics (3, 3)."""
    image = image.unsqueeze(0).float()
    H, W = image.shape[-2:]
    ar = W / H
    lo, hi = cfg.get("num_tokens_range", [1200, 3600])
    num_tokens = int(lo + (resolution_level / 9) * (hi - lo))
    out = moge_forward(sd, cfg, image, num_tokens)
    return infer_tail(out, ar)


def infer_tail(out: Dict[str, torch.Tensor], ar: float) -> Dict[str, torch.Tensor]:
    """The post-network part of MoGeModel.infer (v2.py:238-274): focal / shift recovery, intrinsics, depth = z + shift,
    x metric scale, mask -> inf.  `out`: {'points' (1,H,W,3), 'mask' (1,H,W) probabilities, 'metric_scale' (1,)}."""
    points, mask = out["points"], out.get("mask")
    mask_binary = mask > 0.5 if mask is not None else None
    focal, shift = post_ref.recover_focal_shift(points, mask_binary if mask_binary is not None
                                                else torch.ones(points.shape[:3], dtype=torch.bool))
    fx = focal / 2 * (1 + ar ** 2) ** 0.5 / ar
    fy = focal / 2 * (1 + ar ** 2) ** 0.5
    K = torch.zeros(1, 3, 3)
    K[:, 0, 0], K[:, 1, 1], K[:, 0, 2], K[:, 1, 2], K[:, 2, 2] = fx, fy, 0.5, 0.5, 1.0   # utils3d intrinsics_from_focal_center
    z = points[..., 2] + shift[..., None, None]
    if mask_binary is not None:
        mask_binary = mask_binary & (z > 0)
    depth = z.clone()
    if "metric_scale" in out:
        depth = depth * out["metric_scale"][:, None, None]
    if mask_binary is not None:
        depth = torch.where(mask_binary, depth, torch.tensor(float("inf")))
    return dict(depth=depth[0], mask=mask_binary[0] if mask_binary is not None else None, intrinsics=K[0],
                focal=focal[0], shift=shift[0], points_affine=points[0],
                metric_scale=out.get("metric_scale", torch.ones(1))[0])
