"""Overlap-based Sim(3) chunk alignment: mirror of utils/reconstruction_alignment.py (create_view_graph_matches :16-37,
align_and_refine_reconstructions :40-198, steps 1-3 + the transform), operating on chunk-file dictionaries instead of
pytheia Reconstructions.  The bundle adjustment of steps 4-5 (:107-171) is third-party Ceres code behind pytheia and is
out of scope; the returned info dict keeps the reference's keys for the parts that exist.
All arithmetic runs in csrc/sim3.hip."""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import torch

from . import ops


def create_view_graph_matches(chunk_size: int, overlap_size: int) -> List[Tuple[int, int]]:
    """(ref_view_idx, qry_view_idx) pairs of the overlapping region — utils/reconstruction_alignment.py:16-37."""
    return [(chunk_size - overlap_size + i, i) for i in range(overlap_size)]


def _overlap_block(chunk: Dict[str, torch.Tensor], frames: List[int], device) -> Dict[str, torch.Tensor]:
    idx = torch.tensor(frames, dtype=torch.long)
    out = {}
    for k in ("points", "keypoints", "masks"):
        t = chunk[k]
        out[k] = t[idx.to(t.device)].to(device).contiguous()
    return out


def estimate_sim3(chunk_ref: Dict, chunk_qry: Dict, view_graph_matches: List[Tuple[int, int]], device="cuda:0",
                  use_masks: bool = False, use_filter: bool = True) -> torch.Tensor:
    """Relative similarity qry -> ref from the overlap views (steps 1-3).  Returns the f64 device vector of
    pi3_sim3_umeyama: s, R(9), t(3), M(16), n_used, n_common, median, rms.
    use_masks=True weights the pairs by both chunks' validity masks (the 'weighted' variant; the reference passes all
    common points, reconstruction_alignment.py:97)."""
    n_ref = int(chunk_ref["points"].shape[0])
    n_qry = int(chunk_qry["points"].shape[0])
    pairs = [(r, q) for (r, q) in view_graph_matches if r < n_ref and q < n_qry]
    if not pairs:
        raise ValueError("no overlapping views between the two chunks")
    ref = _overlap_block(chunk_ref, [r for r, _ in pairs], device)
    qry = _overlap_block(chunk_qry, [q for _, q in pairs], device)
    idx = ops.sim3_match_keypoints(ref["keypoints"].to(torch.float16), qry["keypoints"].to(torch.float16))
    # "last camera" of the reference reconstruction = its last view (reconstruction_alignment.py:79)
    last_pose = chunk_ref["camera_poses"][n_ref - 1].to(device, torch.float32).contiguous()
    w_ref = ref["masks"].reshape(len(pairs), -1).to(torch.uint8).contiguous() if use_masks else None
    w_qry = qry["masks"].reshape(len(pairs), -1).to(torch.uint8).contiguous() if use_masks else None
    return ops.sim3_umeyama(ref["points"].to(torch.float16), qry["points"].to(torch.float16), idx, last_pose,
                            w_ref, w_qry, use_filter)


def transform_chunk(chunk: Dict, M4: torch.Tensor, device="cuda:0") -> None:
    """TransformReconstruction4 (reconstruction_alignment.py:105) on a chunk dict, in place: world points and
    cam->world poses.  Points keep their storage dtype (fp16 in chunk files)."""
    pts = chunk["points"].to(device, torch.float32).contiguous()
    poses = chunk["camera_poses"].to(device, torch.float32).contiguous()
    ops.sim3_apply(M4.to(device), pts, poses)
    chunk["points"] = pts.to(chunk["points"].dtype).to(chunk["points"].device)
    chunk["camera_poses"] = poses.to(chunk["camera_poses"].device)


def align_and_refine_reconstructions(chunk_ref: Dict, chunk_qry: Dict, view_graph_matches: List[Tuple[int, int]],
                                     use_inverse_depth: bool = False, device="cuda:0",
                                     use_masks: bool = False) -> Tuple[bool, Dict]:
    """Same contract as the reference (returns (False, {"error": ...}) instead of raising): chunk_qry is transformed
    in place into chunk_ref's frame."""
    print("🔄 Starting reconstruction alignment (closed-form Sim(3) over the overlap views)...")
    try:
        out = estimate_sim3(chunk_ref, chunk_qry, view_graph_matches, device, use_masks)
        o = out.cpu()
        n_used = int(o[29].item())
        if n_used < 3 or not torch.isfinite(o[:29]).all():
            print("❌ Sim3 alignment failed")
            return False, {"error": "sim3_failed", "num_common_tracks": n_used}
        transform_chunk(chunk_qry, out[13:29].contiguous(), device)
        info = {"success": True, "num_common_tracks": n_used,
                "sim3_summary": {"success": True, "alignment_error": float(o[32].item()), "scale": float(o[0].item()),
                                 "matrix": o[13:29].reshape(4, 4).clone()},
                "priors_set": 0, "bundle_adjustment": None}
        return True, info
    except Exception as e:  # noqa: BLE001 - the reference swallows and reports (reconstruction_alignment.py:194-198)
        print(f"❌ Complete reconstruction alignment failed: {e}")
        return False, {"error": "exception", "message": str(e)}
