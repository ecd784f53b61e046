"""One chunk-sized bundle adjustment (100 cameras x 200 keypoints, the bench's synthetic problem) run N times: the
command to put under `rocprofv3 --kernel-trace --stats` to see where an LM iteration's time goes.

    python tools/dev_ba_profile.py [reps] [form]      form: euclidean | homogeneous | inverse_depth
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
from pi3_slam_amd import ops

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
form = sys.argv[2] if len(sys.argv) > 2 else "homogeneous"
kw = {"euclidean": {}, "homogeneous": dict(homogeneous=True), "inverse_depth": dict(inverse_depth=True)}[form]
dev = torch.device("cuda:0")
CL, KP = 100, 200
pb = bench.synthetic_ba_problem(CL, KP, seed=3, noise_px=0.5, perturb=1.0)
to = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)   # noqa: E731
uv_d, valid_d, intr_d = to(pb["uv"]), to(pb["valid"]), to(pb["intr"])
ts = []
for rep in range(reps):
    pts = to(pb["X"])
    rc = to(np.concatenate([pb["R"].reshape(CL, 9), pb["C"]], 1))
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    s = ops.bundle_adjust(pts, rc, intr_d, uv_d, valid_d, 2.0, 10, **kw).cpu().numpy()
    ts.append(time.perf_counter() - t0)
print(f"{form}: {1e3 * min(ts):.2f} ms for {int(s[5])} LM iterations ({int(s[6])} accepted), cost {s[8]:.1f} -> {s[0]:.1f}")
