"""Bundle adjustment of chunk reconstructions: the host side of csrc/ba.hip.

Mirror of what the reference does through pytheia around its BundleAdjustReconstruction calls:
  * ChunkPTRecon.create_recon_from_chunk (utils/chunk_reconstruction.py:36-222): views with the chunk's poses and
    intrinsics priors (default fx = fy = max(W, H), principal point at the centre when a chunk has no intrinsics,
    :96-107), one track per (frame, keypoint) with its own keypoint observation (:128-160) plus the projections into all
    earlier frames and the next max_observations_per_track // 2 frames that land inside the image (:162-185), then
    10 LM iterations with Huber width 2.0 (:188-209) and SetOutlierTracksToUnestimated(tracks, 2, 0.25) (:218);
  * align_and_refine_reconstructions steps 4-5 (utils/reconstruction_alignment.py:107-171): orientation / position
    priors (covariance 2 I / 25 I) on the query chunk's overlap views taken from the reference chunk's views, 50 LM
    iterations with Huber width 3.0, outlier tracks (3, 0.25).
The arithmetic of those pytheia calls is third-party C++ that is not available offline: parity UNPINNED, see
csrc/ba.hip and oracle/ba_ref.py.  Everything below is bookkeeping; the solver runs in the HIP library."""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import torch

from . import ops

# The last three keys configure the sanity gate below, which has NO counterpart in the reference (Theia applies whatever
# Ceres returns): sanity_gate=False switches it off for reference-parity runs (OfflineReconstructor(ba_sanity_gate=False),
# `cli reconstruct --no-ba-sanity-gate`); max_camera_move_extents = how many scene extents a camera centre may move;
# min_surviving_tracks = how many of the tracks that took part must still be estimated.
# inverse_depth (default False): the reference's --use-inverse-depth - use_inverse_depth_parametrization = True and
# use_homogeneous_point_parametrization = False (pi3_bundle_adjust_inverse_depth).
# homogeneous_points: Theia's use_homogeneous_point_parametrization, which the reference sets to True unless
# --use-inverse-depth is given (utils/chunk_reconstruction.py:199-204, utils/reconstruction_alignment.py:147-152): tracks step in the tangent space of their 4-vector (pi3_bundle_adjust_homogeneous).  False = Euclidean steps.
PER_CHUNK = dict(max_iters=10, huber_width=2.0, max_reprojection_px=2.0, min_triangulation_angle_deg=0.25,
                 sanity_gate=True, max_camera_move_extents=1.0, min_surviving_tracks=3, homogeneous_points=True)
AFTER_ALIGNMENT = dict(max_iters=50, huber_width=3.0, max_reprojection_px=3.0, min_triangulation_angle_deg=0.25,
                       sanity_gate=True, max_camera_move_extents=1.0, min_surviving_tracks=3, homogeneous_points=True)
PRIOR_SQRT_INFO_ROT = (1.0 / 2.0) ** 0.5        # orientation prior covariance 2 I (reconstruction_alignment.py:123)
PRIOR_SQRT_INFO_POS = (1.0 / 25.0) ** 0.5       # position prior covariance 25 I (:127)


def chunk_intrinsics(chunk: Dict, W: int, H: int, device) -> torch.Tensor:
    """(N, 4) f64 fx, fy, cx, cy: the chunk's estimated intrinsics, else the reference's default (:96-107)."""
    N = int(chunk["camera_poses"].shape[0])
    K3 = chunk.get("intrinsics")
    if K3 is None:
        f = float(max(W, H))
        return torch.tensor([[f, f, W / 2.0, H / 2.0]], dtype=torch.float64, device=device).repeat(N, 1)
    K3 = K3.to(device, torch.float64)
    return torch.stack([K3[:, 0, 0], K3[:, 1, 1], K3[:, 0, 2], K3[:, 1, 2]], dim=1).contiguous()


def poses_to_rc(cam_to_world: torch.Tensor) -> torch.Tensor:
    """cam->world (N,4,4) -> (N,12) f64 [R world->camera | centre] (SetOrientationFromRotationMatrix(pose[:3,:3].T),
    SetPosition(pose[:3,3]), chunk_reconstruction.py:118-119)."""
    P = cam_to_world.to(torch.float64)
    return torch.cat([P[:, :3, :3].transpose(1, 2).reshape(-1, 9), P[:, :3, 3]], dim=1).contiguous()


def rc_to_poses(rc: torch.Tensor) -> torch.Tensor:
    N = rc.shape[0]
    P = torch.zeros(N, 4, 4, dtype=torch.float64, device=rc.device)
    P[:, :3, :3] = rc[:, :9].reshape(N, 3, 3).transpose(1, 2)
    P[:, :3, 3] = rc[:, 9:]
    P[:, 3, 3] = 1.0
    return P


def build_observations(chunk: Dict, W: int, H: int, max_observations_per_track: int, device):
    """Dense observation arrays of a chunk: uv f32 [N,N,K,2] / valid u8 [N,N,K] from pi3_project_observations, with
    the diagonal (a track in its own frame) holding the keypoint pixel."""
    pts16 = chunk["points"].to(device, torch.float16).contiguous()
    poses = chunk["camera_poses"].to(device, torch.float32).contiguous()
    N, K = pts16.shape[:2]
    intr = chunk_intrinsics(chunk, W, H, device)
    K3 = torch.zeros(N, 3, 3, device=device, dtype=torch.float32)
    K3[:, 0, 0], K3[:, 1, 1], K3[:, 0, 2], K3[:, 1, 2], K3[:, 2, 2] = intr[:, 0], intr[:, 1], intr[:, 2], intr[:, 3], 1.0
    uv, valid = ops.project_observations(pts16, poses, K3, int(W), int(H), max_observations_per_track // 2)
    idx = torch.arange(N, device=device)
    uv[idx, idx] = chunk["keypoints"].to(device, torch.float32)
    valid = valid.to(torch.uint8)
    valid[idx, idx] = 1
    return uv.contiguous(), valid.contiguous(), intr


def ba_summary(infos: List[Optional[Dict]]) -> Dict[str, int]:
    """Counts over a run's adjustments: ran / applied / rejected by the sanity gate / failed numerically."""
    infos = [i for i in infos if i]
    rejected = sum(1 for i in infos if i.get("rejected"))
    applied = sum(1 for i in infos if i.get("success"))
    return {"ran": len(infos), "applied": applied, "rejected_by_sanity_gate": rejected,
            "failed": len(infos) - applied - rejected}


def sanity_gate(pts0: torch.Tensor, rc0: torch.Tensor, pts1: torch.Tensor, rc1: torch.Tensor,
                est: torch.Tensor, est_before: Optional[torch.Tensor], max_camera_move_extents: float = 1.0,
                min_surviving_tracks: int = 3) -> Optional[str]:
    """Why a finished adjustment must NOT be taken over, or None.  Ceres reports success for any run that ends
    without a numerical failure, and so does ba.hip; on geometry the chunk's own projections contradict (e.g. recipe
    weights) LM walks the cameras away by many scene extents and every track ends up an outlier.  Two plain facts are
    checked: at least three of the tracks that took part are still estimated (fewer cannot even fix a pose), and no
    camera centre moved further than the extent of the scene (bounding-box diagonal of the input points and centres)."""
    took_part = est_before if est_before is not None else torch.ones_like(est)
    left = int((est & took_part).sum())
    if int(took_part.sum()) >= min_surviving_tracks and left < min_surviving_tracks:
        return f"{left} of {int(took_part.sum())} tracks survived the outlier test"
    finite = torch.isfinite(pts0).all(dim=1)
    cloud = torch.cat([pts0[finite], rc0[:, 9:]], dim=0)
    extent = float((cloud.max(dim=0).values - cloud.min(dim=0).values).norm())
    moved = float((rc1[:, 9:] - rc0[:, 9:]).norm(dim=1).max())
    if not moved <= max_camera_move_extents * extent:
        return f"a camera moved {moved:.3g} units, the scene extent is {extent:.3g}"
    return None


def bundle_adjust_chunk(chunk: Dict, W: int, H: int, max_observations_per_track: int = 5, device="cuda:0",
                        settings: Dict = PER_CHUNK, priors: Optional[Dict[int, torch.Tensor]] = None,
                        release_observations: bool = False) -> Dict:
    """Refine chunk['points'] (-> fp32) and chunk['camera_poses'] in place; chunk['track_estimated'] (N, K) bool marks
    the tracks SetOutlierTracksToUnestimated keeps (cumulative: a track set to unestimated by an earlier adjustment
    takes no part in a later one, as Theia's BundleAdjustReconstruction only adds IsEstimated() tracks).
    priors: {view index: cam->world 4x4 of the reference view}.  release_observations: drop the cached observation
    arrays (device memory, ~18 MB at N=100, K=200) once this adjustment is done - no later one will follow.
    Returns {'success', 'initial_cost', 'final_cost', 'iterations', 'accepted_steps', 'removed_tracks'[, 'rejected']};
    a result the sanity gate rejects leaves the chunk untouched and reports success=False with the reason."""
    if "keypoints" not in chunk or chunk.get("keypoints") is None:
        return {"success": False, "reason": "no keypoints"}
    N, K = chunk["points"].shape[:2]
    if N > 128:
        return {"success": False, "reason": "more than 128 views in a chunk"}
    # the reference adds the observations once, when the reconstruction is built from the chunk file values
    # (chunk_reconstruction.py:128-185), and every later adjustment reuses them (pixels do not change under Sim(3))
    if "_observations" not in chunk:
        chunk["_observations"] = build_observations(chunk, W, H, max_observations_per_track, device)
    uv, valid, intr = chunk["_observations"]
    est_before = chunk.get("track_estimated")
    if est_before is not None:       # tracks an earlier adjustment set to unestimated stay out (chunk_reconstruction.py:218
        est_before = est_before.to(device).bool()                      # then reconstruction_alignment.py:159)
        valid = valid & est_before[:, None, :].to(valid.dtype)
    pts = chunk["points"].to(device, torch.float64).reshape(N * K, 3).contiguous()
    rc = poses_to_rc(chunk["camera_poses"].to(device))
    pts0, rc0 = pts.clone(), rc.clone()
    pr = pc = pf = None
    if priors:
        pr = torch.zeros(N, 9, dtype=torch.float64, device=device)
        pc = torch.zeros(N, 3, dtype=torch.float64, device=device)
        pf = torch.zeros(N, dtype=torch.uint8, device=device)
        for v, P in priors.items():
            if 0 <= v < N:
                r = poses_to_rc(P.to(device).reshape(1, 4, 4))[0]
                pr[v], pc[v], pf[v] = r[:9], r[9:], 1
    summary = ops.bundle_adjust(pts, rc, intr, uv, valid, settings["huber_width"], settings["max_iters"], pr, pc, pf,
                                PRIOR_SQRT_INFO_ROT, PRIOR_SQRT_INFO_POS,
                                homogeneous=bool(settings.get("homogeneous_points", True)) and not settings.get("inverse_depth"),
                                inverse_depth=bool(settings.get("inverse_depth", False)))
    est = ops.ba_outlier_tracks(pts, rc, intr, uv, valid, settings["max_reprojection_px"],
                                settings["min_triangulation_angle_deg"])
    s = summary.cpu()
    if est_before is not None:
        est = est & est_before
    if release_observations:
        chunk.pop("_observations", None)
    ok = bool(torch.isfinite(s[0])) and bool(torch.isfinite(pts).all()) and bool(torch.isfinite(rc).all())
    info = {"success": ok, "initial_cost": float(s[8]), "final_cost": float(s[0]), "iterations": int(s[5]),
            "accepted_steps": int(s[6]), "removed_tracks": int((~est).sum().item())}
    why = (sanity_gate(pts0, rc0, pts, rc, est, est_before, settings.get("max_camera_move_extents", 1.0),
                       settings.get("min_surviving_tracks", 3))
           if ok and settings.get("sanity_gate", True) else None)
    if why is not None:
        print(f"   ⚠️  bundle adjustment not applied: {why}")
        info.update(success=False, rejected=why)
        return info
    if ok:
        dst = chunk["points"].device
        chunk["points"] = pts.reshape(N, K, 3).to(torch.float32).to(dst)
        chunk["camera_poses"] = rc_to_poses(rc).to(torch.float32).to(chunk["camera_poses"].device)
        chunk["track_estimated"] = est.to(dst)
    return info


def overlap_priors(chunk_ref: Dict, view_graph_matches: List[Tuple[int, int]]) -> Dict[int, torch.Tensor]:
    """Pose priors for the query chunk's overlap views = the reference chunk's poses of the same images
    (reconstruction_alignment.py:110-132)."""
    n_ref = int(chunk_ref["camera_poses"].shape[0])
    return {q: chunk_ref["camera_poses"][r] for r, q in view_graph_matches if r < n_ref}
