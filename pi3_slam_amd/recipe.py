"""Recipe weights: deterministic, name-keyed parameter values shared by the oracle, the fixtures and the GPU engine.

There are no pretrained checkpoints offline, so every parity fixture and the bench use weights that
are regenerated from a recipe instead of stored: value[i] = offset + scale * u(seed(name), i), where u is a
counter-based splitmix64 stream mapped to [-1, 1).  `recipe_tensor` (numpy, integer ops + one fp64 multiply-add) and
`pi3_recipe_fill` (csrc/recipe.hip) produce identical bits; tests/test_recipe.py checks that.

(offset, scale) depend only on the parameter's role, chosen so that activations stay O(1) through 75 blocks while
every parameter (biases, LayerScale gammas, norms, tokens) is non-trivial.
"""
from __future__ import annotations

import math
from typing import Tuple

import numpy as np

_M64 = (1 << 64) - 1


def fnv1a64(name: str) -> int:
    h = 0xCBF29CE484222325
    for b in name.encode("utf-8"):
        h ^= b
        h = (h * 0x100000001B3) & _M64
    return h


def recipe_unit(seed: int, n: int, start: int = 0) -> np.ndarray:
    """u in [-1, 1): float32 array of n values for counters start .. start+n-1."""
    with np.errstate(over="ignore"):
        i = np.arange(start + 1, start + n + 1, dtype=np.uint64)
        z = np.uint64(seed) + i * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    u = (z >> np.uint64(40)).astype(np.float32)
    return u * np.float32(2.0 ** -23) - np.float32(1.0)


def recipe_tensor(name: str, shape, offset: float, scale: float) -> np.ndarray:
    n = int(np.prod(shape))
    out = np.empty(n, dtype=np.float32)
    seed = fnv1a64(name)
    step = 1 << 24
    for s in range(0, n, step):
        m = min(step, n - s)
        u = recipe_unit(seed, m, s)
        # fp32 x fp32 is exact in fp64: no dependence on FMA contraction (matches csrc/recipe.hip bit for bit)
        out[s:s + m] = (np.float64(np.float32(offset)) + np.float64(np.float32(scale)) * u.astype(np.float64)
                        ).astype(np.float32)
    return out.reshape(shape)


def recipe_params(name: str, shape) -> Tuple[float, float]:
    """(offset, scale) for a parameter, by role.  `name` is the reference state_dict key (SURVEY.md §8c list)."""
    leaf = name.split(".")[-1]
    parent = name.split(".")[-2] if "." in name else ""
    if name in ("image_mean", "image_std"):
        raise ValueError("buffers are constants, not recipe tensors")
    if leaf == "gamma":  # LayerScale: encoder init 1.0 / decoder init 0.01 in the reference; trained values are O(0.1)
        return 0.15, 0.05
    if parent.startswith("norm") or parent in ("q_norm", "k_norm", "norm"):
        return (1.0, 0.1) if leaf == "weight" else (0.0, 0.05)
    if leaf in ("cls_token", "register_tokens", "register_token", "pos_embed"):
        return 0.0, 0.05
    if leaf == "bias":
        return 0.0, 0.02
    if leaf == "weight":
        fan_in = int(np.prod(shape[1:]))
        gain = 1.0
        if name.startswith("point_head") or name.startswith("conf_head"):
            gain = 0.15  # keeps exp(z) in a sane range (|z| <~ 2)
        return 0.0, gain * math.sqrt(3.0 / fan_in)
    raise ValueError(f"no recipe for parameter {name}")
