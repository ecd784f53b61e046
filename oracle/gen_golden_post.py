"""ORACLE tooling — generates tests/golden/post_*.npz by running the reference's OWN glue functions (imported from
/root/reference; build container only).  Third-party modules the reference imports at module level but that are
absent here (cv2, torchvision, torchcodec, pytheia, natsort, plyfile) are replaced by EMPTY placeholder modules: none
of their attributes is ever called by the functions exercised below (SURVEY.md §8c did the same to measure them).

    python oracle/gen_golden_post.py
"""
from __future__ import annotations

import os
import sys
import types

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path.insert(0, REPO)

from pi3_slam_amd.recipe import fnv1a64, recipe_unit  # noqa: E402


def synthetic_chunk(name: str, N: int, H: int, W: int):
    """Seeded dense maps that look like a pi3 output: pinhole-consistent local points (so focal/shift are
    recoverable), a depth step (so depth_edge fires), random confidence logits around the 0.1-sigmoid threshold."""
    def u(tag, n):
        return recipe_unit(fnv1a64(f"golden.post.{name}.{tag}"), n)
    ys, xs = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing="ij")
    f_true = 0.9 * max(H, W)
    lp = np.zeros((N, H, W, 3), dtype=np.float32)
    for n in range(N):
        z = 2.0 + 0.1 * np.sin(xs / 9.0 + n) + 0.06 * np.cos(ys / 7.0) + 0.004 * u(f"z{n}", H * W).reshape(H, W)
        z[:, W // 2:] += 0.8 + 0.1 * n          # depth discontinuity
        z = z.astype(np.float32)
        shift = 0.15 * (n + 1)
        lp[n, ..., 0] = (xs - W / 2) / f_true * (z + shift)
        lp[n, ..., 1] = (ys - H / 2) / f_true * (z + shift)
        lp[n, ..., 2] = z
    conf = (-2.2 + 1.5 * u("conf", N * H * W)).reshape(N, H, W, 1).astype(np.float32)
    poses = np.tile(np.eye(4, dtype=np.float32), (N, 1, 1))
    for n in range(N):
        a = 0.1 * n
        poses[n, :3, :3] = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]], np.float32)
        poses[n, :3, 3] = [0.2 * n, 0.01 * n, 0.05 * n]
    hom = np.concatenate([lp, np.ones_like(lp[..., :1])], -1)
    pts = np.einsum("nij,nhwj->nhwi", poses, hom)[..., :3].astype(np.float32)
    images = (0.5 + 0.5 * u("img", N * 3 * H * W)).reshape(N, 3, H, W).astype(np.float32)
    moge = (lp[0, ..., 2] * 1.37 * (1.0 + 0.05 * u("moge", H * W).reshape(H, W))).astype(np.float32)
    return dict(local_points=torch.from_numpy(lp), conf=torch.from_numpy(conf), camera_poses=torch.from_numpy(poses),
                points=torch.from_numpy(pts), images=torch.from_numpy(images), moge_depth=torch.from_numpy(moge))


class _Placeholder(types.ModuleType):
    """Empty stand-in for an absent third-party module: any attribute is again an inert placeholder that is never called."""
    __path__ = []

    def __getattr__(self, item):
        if item.startswith("__"):
            raise AttributeError(item)
        return _Placeholder(f"{self.__name__}.{item}")


CASES = {"post_a": (4, 56, 70), "post_b": (3, 84, 112)}


def main() -> None:
    for name in ("cv2", "natsort", "plyfile", "torchvision", "torchvision.transforms", "torchcodec",
                 "torchcodec.decoders", "pytheia"):
        if name not in sys.modules:
            sys.modules[name] = _Placeholder(name)
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    sys.path.insert(0, REF)
    os.chdir(REF)
    from slam.offline_chunk_creator import OfflineChunkCreator
    from utils.keypoint_extraction import GridKeypointExtractor
    from utils.geometry_torch import recover_focal_shift
    from utils.reconstruction_alignment import create_view_graph_matches
    from datasets.image_datasets import ChunkImageDataset

    out_dir = os.path.join(REPO, "tests", "golden")
    for name, (N, H, W) in CASES.items():
        d = synthetic_chunk(name, N, H, W)
        res = {"points": d["points"][None], "local_points": d["local_points"][None], "conf": d["conf"][None]}
        masks = OfflineChunkCreator._compute_masks(res)[0]
        scale = OfflineChunkCreator._get_scale_factor_for_pi3(d["moge_depth"], d["local_points"][0][..., 2], masks[0])
        save = {"shape": np.array([N, H, W]), "masks": masks.numpy(), "scale": np.array(scale.item(), np.float32)}
        for tag, max_kp, seed in (("full", 4096, None), ("sub", 12, 1234)):
            ext = GridKeypointExtractor(max_num_keypoints=max_kp, device="cpu")
            if seed is not None:
                torch.manual_seed(seed)
            kp = ext.extract_with_colors(d["images"][None])
            dense = dict(points=d["points"], local_points=d["local_points"], conf=d["conf"], masks=masks,
                         images=d["images"])
            interp = OfflineChunkCreator._interpolate_world_points_for_keypoints(dense, kp["keypoints"])
            save[f"kp_{tag}"] = kp["keypoints"].numpy()
            save[f"colors_{tag}"] = kp["colors"].numpy()
            save[f"ipoints_{tag}"] = interp["points"].to(torch.float16).numpy()
            save[f"ilocal_{tag}"] = interp["local_points"].to(torch.float16).numpy()
            save[f"iconf_{tag}"] = interp["conf"].to(torch.float16).numpy()
            save[f"imasks_{tag}"] = interp["masks"].numpy()
            print(name, tag, "keypoints", tuple(kp["keypoints"].shape), "spacing",
                  ext._calculate_grid_spacing(H, W))
        conf_masks = torch.sigmoid(d["conf"][..., 0]) > 0.1
        focal, shift = recover_focal_shift(d["local_points"][None], conf_masks[None])
        save["focal"], save["shift"] = focal[0].numpy(), shift[0].numpy()
        print(name, "focal", focal[0].tolist(), "shift", shift[0].tolist(), "scale", scale.item(),
              "mask frac", masks.float().mean().item())
        np.savez_compressed(os.path.join(out_dir, name + ".npz"), **save)

    layout = {}
    for (n, cl, ov) in [(32, 32, 8), (1000, 100, 20), (4000, 100, 20), (10, 4, 1), (7, 100, 20), (101, 100, 99)]:
        ds = ChunkImageDataset([f"f{i}.png" for i in range(n)], cl, ov, (28, 42))
        layout[f"chunks_{n}_{cl}_{ov}"] = np.array(ds.chunk_indices, dtype=np.int64).reshape(-1, 2)
    for (cl, ov) in [(100, 20), (32, 8), (5, 0)]:
        layout[f"vgm_{cl}_{ov}"] = np.array(create_view_graph_matches(cl, ov), dtype=np.int64).reshape(-1, 2)
    np.savez_compressed(os.path.join(out_dir, "post_layout.npz"), **layout)

    # next tier: observation projection of ChunkPTRecon, run through the reference's own methods
    for name in ("matplotlib", "matplotlib.pyplot", "matplotlib.animation"):
        if name not in sys.modules:
            try:
                __import__(name)
            except Exception:
                sys.modules[name] = _Placeholder(name)
    from utils.chunk_reconstruction import ChunkPTRecon
    rec = object.__new__(ChunkPTRecon)  # the projection methods use no constructor state (pytheia is absent)
    N, K, W, H, max_obs = 7, 40, 112, 84, 5
    rec.set_target_size(W, H)
    g = torch.Generator().manual_seed(77)
    ang = 0.25 * torch.randn(N, 3, generator=g, dtype=torch.float64)
    poses = torch.eye(4, dtype=torch.float64).repeat(N, 1, 1)
    for i in range(N):
        th = ang[i].norm()
        k = ang[i] / th
        Kx = torch.tensor([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]], dtype=torch.float64)
        poses[i, :3, :3] = torch.eye(3, dtype=torch.float64) + torch.sin(th) * Kx + (1 - torch.cos(th)) * Kx @ Kx
        poses[i, :3, 3] = 0.3 * torch.randn(3, generator=g, dtype=torch.float64)
    poses = poses.float()
    intr = torch.tensor([[90.0, 0, W / 2], [0, 92.0, H / 2], [0, 0, 1]]).repeat(N, 1, 1)
    intr[:, 0, 0] += torch.arange(N) * 0.5
    pts = torch.randn(N, K, 3, generator=g) * torch.tensor([1.0, 0.8, 1.0]) + torch.tensor([0.0, 0.0, 3.0])
    pts = pts.to(torch.float16)
    chunk = {"camera_poses": poses, "intrinsics": intr, "points": pts, "keypoints": torch.zeros(N, K, 2)}
    uv = np.zeros((N, N, K, 2))
    valid = np.zeros((N, N, K), dtype=bool)
    for src in range(N):
        frames = list(range(N))
        targets = frames[:src] + frames[src + 1: src + max_obs // 2 + 1]
        for tgt, pr in zip(targets, rec._project_points_to_other_cams(chunk, src, targets)):
            uv[src, tgt] = pr
            valid[src, tgt] = [(0 <= q[0] < rec.original_width and 0 <= q[1] < rec.original_height) for q in pr]
    np.savez_compressed(os.path.join(out_dir, "post_proj.npz"), poses=poses.numpy(), intrinsics=intr.numpy(),
                        points=pts.numpy(), uv=uv, valid=valid, shape=np.array([N, K, W, H, max_obs // 2]))
    print("projection golden: valid frac", valid.mean())
    print("wrote post goldens to", out_dir)


if __name__ == "__main__":
    main()
