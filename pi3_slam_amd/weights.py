"""Parameter inventory of the pi3 network (names == the reference state_dict keys) and the ways to
obtain values for them: a checkpoint (`model.safetensors`, pi3/models/pi3.py:14-16 PyTorchModelHubMixin layout) or the
deterministic recipe (recipe.py) generated directly on the device.
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Callable, Dict, Iterator, Tuple

import numpy as np
import torch

from .recipe import fnv1a64, recipe_params, recipe_tensor


@dataclass(frozen=True)
class Pi3Config:
    """Architecture hyper-parameters.  Defaults = the released pi3 model (pi3/models/pi3.py:17-129: DINOv2 ViT-L/14
    with 4 registers, 36-block 'large' decoder, three 5-block heads).  Smaller values exist only so that the parity
    tests can run the same code path quickly."""
    dim: int = 1024
    enc_depth: int = 24
    dec_depth: int = 36
    head_depth: int = 5
    cam_dim: int = 512
    pos_grid: int = 37          # 518 / 14 (dinov2_vitl14_reg: img_size 518)
    n_enc_reg: int = 4
    n_dec_reg: int = 5
    rope_base: float = 100.0
    eps: float = 1e-6

    @property
    def heads(self) -> int:
        return self.dim // 64


def _block_shapes(prefix: str, D: int, ls: bool, qk_norm: bool) -> Iterator[Tuple[str, tuple]]:
    yield f"{prefix}.norm1.weight", (D,)
    yield f"{prefix}.norm1.bias", (D,)
    yield f"{prefix}.attn.qkv.weight", (3 * D, D)
    yield f"{prefix}.attn.qkv.bias", (3 * D,)
    yield f"{prefix}.attn.proj.weight", (D, D)
    yield f"{prefix}.attn.proj.bias", (D,)
    if qk_norm:
        for n in ("q_norm", "k_norm"):
            yield f"{prefix}.attn.{n}.weight", (64,)
            yield f"{prefix}.attn.{n}.bias", (64,)
    if ls:
        yield f"{prefix}.ls1.gamma", (D,)
    yield f"{prefix}.norm2.weight", (D,)
    yield f"{prefix}.norm2.bias", (D,)
    yield f"{prefix}.mlp.fc1.weight", (4 * D, D)
    yield f"{prefix}.mlp.fc1.bias", (4 * D,)
    yield f"{prefix}.mlp.fc2.weight", (D, 4 * D)
    yield f"{prefix}.mlp.fc2.bias", (D,)
    if ls:
        yield f"{prefix}.ls2.gamma", (D,)


def param_shapes(cfg: Pi3Config) -> Dict[str, tuple]:
    """Ordered name -> shape map of every learnable tensor (1210 entries for the default config)."""
    D, C = cfg.dim, cfg.cam_dim
    out: Dict[str, tuple] = {}
    out["encoder.cls_token"] = (1, 1, D)
    out["encoder.pos_embed"] = (1, cfg.pos_grid * cfg.pos_grid + 1, D)
    out["encoder.register_tokens"] = (1, cfg.n_enc_reg, D)
    out["encoder.patch_embed.proj.weight"] = (D, 3, 14, 14)
    out["encoder.patch_embed.proj.bias"] = (D,)
    for i in range(cfg.enc_depth):
        out.update(_block_shapes(f"encoder.blocks.{i}", D, True, False))
    out["encoder.norm.weight"] = (D,)
    out["encoder.norm.bias"] = (D,)
    for i in range(cfg.dec_depth):
        out.update(_block_shapes(f"decoder.{i}", D, True, True))
    out["register_token"] = (1, 1, cfg.n_dec_reg, D)
    for head, od in (("point_decoder", D), ("conf_decoder", D), ("camera_decoder", C)):
        out[f"{head}.projects.weight"] = (D, 2 * D)
        out[f"{head}.projects.bias"] = (D,)
        for i in range(cfg.head_depth):
            out.update(_block_shapes(f"{head}.blocks.{i}", D, False, False))
        out[f"{head}.linear_out.weight"] = (od, D)
        out[f"{head}.linear_out.bias"] = (od,)
    out["point_head.proj.weight"] = (3 * 196, D)
    out["point_head.proj.bias"] = (3 * 196,)
    out["conf_head.proj.weight"] = (196, D)
    out["conf_head.proj.bias"] = (196,)
    for r in range(2):
        for l in (1, 2, 3):
            out[f"camera_head.res_conv.{r}.res_conv{l}.weight"] = (C, C)
            out[f"camera_head.res_conv.{r}.res_conv{l}.bias"] = (C,)
    for l in (0, 2):
        out[f"camera_head.more_mlps.{l}.weight"] = (C, C)
        out[f"camera_head.more_mlps.{l}.bias"] = (C,)
    out["camera_head.fc_t.weight"] = (3, C)
    out["camera_head.fc_t.bias"] = (3,)
    out["camera_head.fc_rot.weight"] = (9, C)
    out["camera_head.fc_rot.bias"] = (9,)
    return out


IMAGE_MEAN = (0.485, 0.456, 0.406)   # pi3/models/pi3.py:124-125
IMAGE_STD = (0.229, 0.224, 0.225)


def recipe_state_dict_cpu(cfg: Pi3Config, names=None) -> Dict[str, torch.Tensor]:
    """numpy recipe -> fp32 CPU tensors (used by the oracle and the fixture generator; 3.8 GB for the default config)."""
    shapes = param_shapes(cfg)
    out = {}
    for name, shape in shapes.items():
        if names is not None and name not in names:
            continue
        off, sc = recipe_params(name, shape)
        out[name] = torch.from_numpy(recipe_tensor(name, shape, off, sc))
    return out


def recipe_fill_device(name: str, shape, device, dtype=torch.float32) -> torch.Tensor:
    """Same values as recipe_tensor(name, ...) generated on the GPU by pi3_recipe_fill (csrc/recipe.hip)."""
    from . import ops
    off, sc = recipe_params(name, shape)
    t = torch.empty(shape, device=device, dtype=dtype)
    ops.recipe_fill(t, fnv1a64(name), off, sc)
    return t


def config_from_checkpoint_dir(path: str) -> "Pi3Config":
    """Pi3.from_pretrained on a local directory (huggingface_hub.PyTorchModelHubMixin) passes the init kwargs stored in
    `<dir>/config.json` to Pi3.__init__ (pi3/models/pi3.py:17-21): `pos_type` ('rope<freq>', default 'rope100') and
    `decoder_size` (default 'large').  The RoPE base follows pos_type; 'small' / 'base' decoders cannot even be run by
    the reference (the 384 / 768-wide decoder is fed the 1024-wide encoder tokens without a projection, pi3.py:140-160),
    so anything but 'large' is refused."""
    import json
    from dataclasses import replace
    cfg = Pi3Config()
    cj = os.path.join(path, "config.json") if os.path.isdir(path) else None
    if cj and os.path.exists(cj):
        with open(cj) as f:
            kw = json.load(f)
        pos_type = kw.get("pos_type", "rope100") or "none"
        if not str(pos_type).startswith("rope"):
            raise NotImplementedError(f"pos_type={pos_type!r}: the reference itself only implements 'rope<freq>' (pi3.py:36-43)")
        cfg = replace(cfg, rope_base=float(str(pos_type)[len("rope"):]))
        size = kw.get("decoder_size", "large")
        if size != "large":
            raise NotImplementedError(f"decoder_size={size!r}: only 'large' is a runnable reference model")
    return cfg


def load_checkpoint(path: str) -> Dict[str, torch.Tensor]:
    """`path` is a directory holding model.safetensors (the PyTorchModelHubMixin layout) or a .safetensors/.pt file."""
    if os.path.isdir(path):
        path = os.path.join(path, "model.safetensors")
    if path.endswith(".safetensors"):
        from safetensors.torch import load_file
        return load_file(path, device="cpu")
    sd = torch.load(path, map_location="cpu", weights_only=True)
    return sd.get("model", sd) if isinstance(sd, dict) else sd
