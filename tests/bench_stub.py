"""Test-only stand-ins for bench.py's GPU objects (PI3_BENCH_STUB=1 with PI3_DIST_BACKEND=gloo): the N > 1 CONTROL FLOW of
`bench.py main()` - rendezvous, warm-up, timed loop, boundary all-gather, the 136-byte record all-gather, prefix
composition, barrier + max-over-ranks timing, the per-rank gather and the JSON line of rank 0 - on a box without a GPU,
so that the first real 8-GPU run cannot fail in Python that never ran.  Nothing here is measured or
shipped: the line a stub run prints says `"stub": true` and carries no roofline.  The solver is the CPU oracle (tests may
use oracle/); the chunks are cut from one synthetic world, so every rank's composed transforms can be checked."""
from __future__ import annotations

import time

import numpy as np
import torch


class StubEngine:
    """Carries the only thing main() asks of the engine outside the timed loop."""

    def flops(self, B, N, H, W):
        return {"total": 0.0}


class StubCreator:
    """`process_chunks(items)` of the product's creator: yields (meta, chunk dict) per item, chunk-file layout."""

    def __init__(self, cl: int, ov: int, kp: int, rank: int, world: int):
        self.cl, self.ov, self.kp, self.rank, self.world = cl, ov, kp, rank, world
        self.model = None
        self.target_size = None
        rng = np.random.default_rng(0)          # the same world on every rank
        self._kp = (rng.random((kp, 2)) * 300).astype(np.float16)
        self._seed_pts = rng.standard_normal((4096, kp, 3)) + np.array([0, 0, 4.0])

    def _chunk(self, c: int):
        cl, ov = self.cl, self.ov
        start = c * (cl - ov)
        ang, s = 0.05 * (c % 7), 1.0 + 0.02 * (c % 5)
        R = np.array([[np.cos(ang), -np.sin(ang), 0], [np.sin(ang), np.cos(ang), 0], [0, 0, 1.0]])
        t = np.array([0.1 * (c % 3), -0.05 * (c % 4), 0.02 * c])
        world_pts = self._seed_pts[(start + np.arange(cl)) % 4096]
        pts = (((world_pts - t) @ R) / s).astype(np.float16)
        poses = np.tile(np.eye(4, dtype=np.float32), (cl, 1, 1))
        poses[:, :3, 3] = ((np.stack([[0.1 * (start + j), 0, 0] for j in range(cl)]) - t) @ R) / s
        return dict(points=torch.from_numpy(pts), keypoints=torch.from_numpy(np.tile(self._kp, (cl, 1, 1))),
                    masks=torch.ones(cl, self.kp, 1, dtype=torch.bool), camera_poses=torch.from_numpy(poses),
                    _metrics={"infer_s": 0.002, "num_frames": cl, "fps": cl / 0.002, "post_s": 0.0, "stage_in_s": 0.0})

    def process_chunks(self, items):
        for i, item in enumerate(items):
            time.sleep(0.002)      # a 'forward'
            # step i of rank r is global chunk i * world + r (bench.py: one chunk per rank per step, wave by wave)
            yield dict(item.get("meta") or {}), self._chunk(i * self.world + self.rank)


def stub_solver(ov: int, cl: int):
    """[accepted, T(16)] with the CPU oracle's closed form (oracle/post_ref.py) on two boundary blocks."""
    from oracle import post_ref

    def solve(prev, cur):
        res = torch.zeros(17, dtype=torch.float64)
        res[1:] = torch.eye(4, dtype=torch.float64).reshape(16)
        ref, qry = prev["tail"], cur["head"]
        out = post_ref.align_chunks(ref["points"].numpy(), qry["points"].numpy(), ref["keypoints"].numpy(),
                                    qry["keypoints"].numpy(), prev["last_pose"].numpy(), True)
        if out["n_used"] >= 3 and np.isfinite(out["M"]).all():
            res[0] = 1.0
            res[1:] = torch.from_numpy(out["M"].reshape(16))
        return res
    return solve


def transform_chunk_cpu(chunk, G: torch.Tensor) -> None:
    """alignment.transform_chunk(absolute=True) without the device: points / poses of the chunk frame times G (f64)."""
    G = G.to(torch.float64).reshape(4, 4)
    src = chunk.setdefault("_chunk_frame", {"points": chunk["points"], "camera_poses": chunk["camera_poses"]})
    p = src["points"].to(torch.float64)
    chunk["points"] = (p @ G[:3, :3].T + G[:3, 3]).to(torch.float32)
    s = torch.linalg.det(G[:3, :3]).abs() ** (1.0 / 3.0)
    P = src["camera_poses"].to(torch.float64)
    out = P.clone()
    out[:, :3, :3] = (G[:3, :3] / s) @ P[:, :3, :3]
    out[:, :3, 3] = P[:, :3, 3] @ G[:3, :3].T + G[:3, 3]
    chunk["camera_poses"] = out.to(torch.float32)
    chunk["_sim3_global"] = G.clone()
