import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pi3_slam_amd import ops
from pi3_slam_amd.recipe import fnv1a64, recipe_tensor
name = "decoder.3.attn.qkv.weight"
for (off, sc) in [(0.01, 0.3), (0.0, 1.0), (0.0, 0.5), (1.0, 0.0)]:
    out = torch.empty(100003, device="cuda:0")
    ops.recipe_fill(out, fnv1a64(name), off, sc)
    ref = recipe_tensor(name, (100003,), off, sc)
    o = out.cpu().numpy()
    bad = np.nonzero(o != ref)[0]
    print(off, sc, "mismatches", len(bad), bad[:8], o[bad[:4]].view(np.uint32), ref[bad[:4]].view(np.uint32))
