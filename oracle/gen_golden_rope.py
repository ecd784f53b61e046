"""ORACLE tooling — generates tests/golden/rope2d.npz with the REAL reference `RoPE2D` (the torch class
pi3/models/layers/pos_embed.py:112-159 that the reference runs whenever its `curope` extension is not importable - as
here: the checked-in curope .so is a CUDA / CPython-3.11 binary), imported from /root/reference (build container only).

    python oracle/gen_golden_rope.py

Inputs: tokens (B, heads, N, D) fp32 from the recipe stream, positions (B, N, 2) int64 in [0, 30) (pi3 positions are
patch coordinates + 1, specials 0: pi3.py:146-152).  Output: RoPE2D(freq=100, F0=1)(tokens, positions).  Two head dims:
64 (pi3) and 32.  `oracle.pi3_ref.rope_2d_cpu` (the restatement of curope.cpp:11-47) and the device entry pi3_rope_2d
are checked against these vectors (tests/test_oracle_golden.py, tests/test_kernels_gpu.py)."""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path.insert(0, REPO)

from pi3_slam_amd.recipe import fnv1a64, recipe_unit  # noqa: E402


def main() -> None:
    sys.path.insert(0, REF)
    from pi3.models.layers.pos_embed import RoPE2D      # prints its "using a slow pytorch version" warning
    assert RoPE2D.__name__ == "RoPE2D" and hasattr(RoPE2D, "apply_rope1d"), "expected the torch class, not cuRoPE2D"
    save = {}
    for tag, (B, H, N, D) in {"d64": (2, 4, 50, 64), "d32": (1, 3, 37, 32)}.items():
        tok = torch.from_numpy(recipe_unit(fnv1a64("golden.rope." + tag), B * H * N * D).reshape(B, H, N, D)
                               .astype(np.float32) * 2.0)
        pos = torch.from_numpy((np.abs(recipe_unit(fnv1a64("golden.rope.pos." + tag), B * N * 2)) * 29.99)
                               .astype(np.int64).reshape(B, N, 2))
        pos[0, 0] = 0                                      # a special token (position 0: identity rotation)
        out = RoPE2D(freq=100.0, F0=1.0)(tok.clone(), pos)
        save[tag + "_tokens"], save[tag + "_positions"], save[tag + "_out"] = tok.numpy(), pos.numpy(), out.numpy()
        print(tag, tuple(tok.shape), "pos range", int(pos.min()), int(pos.max()), "max|out|", float(out.abs().max()))
    path = os.path.join(REPO, "tests", "golden", "rope2d.npz")
    np.savez_compressed(path, **save)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
