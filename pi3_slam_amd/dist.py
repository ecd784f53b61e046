"""Chunk-parallel execution over the GPUs of one node (one process per GPU, torch.distributed: backend "nccl" is RCCL
over xGMI on ROCm, "gloo" on CPU for tests).

The reference has no distributed code.  The path shards at chunk granularity: chunk creation has no
cross-chunk dependency (each chunk's metric scale comes from its own first frame, offline_chunk_creator.py:184-187).
The only exchange is for the progressive alignment: rank r needs the overlap block of chunk c-1 (the last `ov` views'
keypoints / world points / validity + the last camera pose, ~50 KB) to compute the relative similarity T_{c-1<-c}, and
every rank needs all T's to compose the global transforms.  Both are ONE all-gather per wave of G chunks (latency-bound
message, no ring reduction), followed by a local prefix product G_c = G_{c-1} . T_c (associative 4x4 products).
This equals the reference's sequential "align to the already transformed previous chunk" for the closed-form step
because the near-half filter and the Umeyama solve are similarity-equivariant.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import torch
import torch.distributed as dist


def ensure_process_group() -> Tuple[int, int]:
    """(rank, world).  Under torch.distributed.run (WORLD_SIZE > 1 in the environment) the process group is created on
    first use - backend "nccl" (= RCCL over xGMI) unless PI3_DIST_BACKEND says otherwise (the CPU / one-GPU tests use
    gloo) - so the reference's unmodified CLIs shard across GPUs once launched with torchrun."""
    import os
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        return 0, 1
    backend = os.environ.get("PI3_DIST_BACKEND", "nccl" if torch.cuda.is_available() else "gloo")
    if backend == "nccl":
        dev = torch.device("cuda", local_device_index())
        torch.cuda.set_device(dev)
        dist.init_process_group("nccl", device_id=dev)
    else:
        dist.init_process_group(backend)
    return dist.get_rank(), dist.get_world_size()


def local_device_index() -> int:
    """cuda index of this rank: LOCAL_RANK, folded onto the visible devices (several ranks may share a card in tests)."""
    import os
    n = max(1, torch.cuda.device_count())
    return int(os.environ.get("LOCAL_RANK", "0")) % n


def gather_objects(obj, dst: int = 0):
    """All ranks' python objects on rank dst (None elsewhere); works on gloo and nccl process groups."""
    world = dist.get_world_size()
    out = [None] * world
    dist.all_gather_object(out, obj)
    return out if dist.get_rank() == dst else None


def shard_chunks(n_chunks: int, rank: int, world: int) -> List[int]:
    """Chunk c runs on rank c % world: neighbouring chunks sit on neighbouring ranks, every wave of `world` chunks is
    complete before its all-gather."""
    return list(range(rank, n_chunks, world))


def pack_boundary(chunk: Dict[str, torch.Tensor], overlap: int, K: int) -> torch.Tensor:
    """Flat fp32 boundary block of one chunk: head (first ov views) and tail (last ov views) of keypoints (as fp16 bit
    patterns widened to fp32-exact integers), world points, validity, and the last camera pose.
    Layout: [n_frames, head_kp(ov*K*2), head_pts(ov*K*3), head_mask(ov*K), tail_kp, tail_pts, tail_mask, pose(16)]."""
    n = int(chunk["points"].shape[0])
    ov = min(overlap, n)

    def blk(sl):
        kp = chunk["keypoints"][sl].to(torch.float16).contiguous().view(torch.int16).to(torch.float32)
        pt = chunk["points"][sl].to(torch.float16).contiguous().view(torch.int16).to(torch.float32)
        mk = chunk["masks"][sl].reshape(-1).to(torch.float32)
        out = torch.zeros(overlap * K * 6)
        out[: ov * K * 2] = kp.reshape(-1)
        out[overlap * K * 2: overlap * K * 2 + ov * K * 3] = pt.reshape(-1)
        out[overlap * K * 5: overlap * K * 5 + ov * K] = mk
        return out

    head, tail = blk(slice(0, ov)), blk(slice(n - ov, n))
    pose = chunk["camera_poses"][n - 1].reshape(-1).to(torch.float32)
    return torch.cat([torch.tensor([float(n)]), head, tail, pose])


def unpack_boundary(flat: torch.Tensor, overlap: int, K: int) -> Dict[str, torch.Tensor]:
    n = int(flat[0].item())
    sz = overlap * K * 6

    def blk(t):
        kp = t[: overlap * K * 2].to(torch.int16).view(torch.float16).reshape(overlap, K, 2)
        pt = t[overlap * K * 2: overlap * K * 5].to(torch.int16).view(torch.float16).reshape(overlap, K, 3)
        mk = t[overlap * K * 5:].reshape(overlap, K, 1) > 0.5
        return dict(keypoints=kp, points=pt, masks=mk)

    return dict(n_frames=n, head=blk(flat[1: 1 + sz]), tail=blk(flat[1 + sz: 1 + 2 * sz]),
                last_pose=flat[1 + 2 * sz: 1 + 2 * sz + 16].reshape(4, 4))


def allgather_boundaries(local: torch.Tensor, device) -> List[torch.Tensor]:
    """One all-gather of the per-rank boundary blocks (RCCL on GPUs; gloo in the CPU tests)."""
    world = dist.get_world_size()
    buf = local.to(device).contiguous()
    out = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(out, buf)
    return out


def relative_sim3_from_boundaries(prev: Dict, cur: Dict, overlap: int, device, use_filter: bool = True,
                                  chunk_length: Optional[int] = None):
    """T_{c-1<-c} from the previous chunk's TAIL block and the current chunk's HEAD block (device kernels).
    View pairs are the nominal (chunk_length - overlap + i, i) of create_view_graph_matches restricted to views that
    exist (like alignment.estimate_sim3): the tail block holds the previous chunk's last min(overlap, n) views, so for a
    short previous chunk nominal view chunk_length - overlap + i sits at tail position i + d."""
    from . import ops
    n_prev, n_cur = int(prev["n_frames"]), int(cur["n_frames"])
    ov_p = min(overlap, n_prev)
    d = 0 if chunk_length is None else (chunk_length - overlap) - (n_prev - ov_p)
    pairs = [(i + d, i) for i in range(overlap) if 0 <= i + d < ov_p and i < min(overlap, n_cur)]
    if not pairs:
        raise ValueError("no overlapping views between the two chunks")
    ri = torch.tensor([r for r, _ in pairs], dtype=torch.long)
    qi = torch.tensor([q for _, q in pairs], dtype=torch.long)
    ref, qry = prev["tail"], cur["head"]
    kp_r = ref["keypoints"][ri].to(device).contiguous()
    kp_q = qry["keypoints"][qi].to(device).contiguous()
    idx = ops.sim3_match_keypoints(kp_r, kp_q)
    return ops.sim3_umeyama(ref["points"][ri].to(device).contiguous(), qry["points"][qi].to(device).contiguous(),
                            idx, prev["last_pose"].to(device, torch.float32).contiguous(), None, None, use_filter)


def compose_global(rel: torch.Tensor) -> torch.Tensor:
    """rel: [n, 16] f64 relative similarities (rel[0] = identity for the first chunk) -> global G [n, 16]."""
    from . import ops
    return ops.sim3_compose_prefix(rel.contiguous())
