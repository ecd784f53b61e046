"""Track observations of one chunk, on the device.

Mirrors the projection half of ``ChunkPTRecon.create_recon_from_chunk`` (utils/chunk_reconstruction.py:162-185): every
frame's keypoint world points are projected into all earlier frames and the next ``max_observations_per_track // 2``
frames, and kept when they fall inside the original image.  The pytheia ``Reconstruction`` the reference feeds these
into (AddObservation) is out of scope; this returns the same observations as flat arrays a caller can hand to any
bundle adjuster.
"""
from __future__ import annotations

from typing import Dict

import torch

from . import ops


def project_chunk_observations(chunk: Dict[str, torch.Tensor], original_width: int, original_height: int,
                               max_observations_per_track: int = 5) -> Dict[str, torch.Tensor]:
    """chunk: the dict a chunk file holds ('points' f16 [N,K,3], 'camera_poses' f32 [N,4,4], 'intrinsics' f32 [N,3,3]).

    Returns source_frame / target_frame / keypoint index (int64) and the projected pixel (f32 [M,2]) of every kept
    observation, ordered like the reference's loops (source, then target, then keypoint)."""
    pts = chunk["points"]
    if pts.dtype != torch.float16:
        pts = pts.to(torch.float16)
    uv, valid = ops.project_observations(pts.contiguous(), chunk["camera_poses"], chunk["intrinsics"],
                                         int(original_width), int(original_height), max_observations_per_track // 2)
    src, tgt, kp = torch.nonzero(valid, as_tuple=True)
    return {"source_frame": src, "target_frame": tgt, "keypoint": kp, "uv": uv[src, tgt, kp]}
