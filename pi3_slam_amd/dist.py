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
    if world <= 1 and os.environ.get("PI3_DIST_FORCE") != "1":
        return 0, 1          # PI3_DIST_FORCE=1: build the group anyway (a 1-rank RCCL group exercises the nccl branch)
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


def resolve_device(device: Optional[str]) -> str:
    """'cuda' / None -> this rank's card (cuda:LOCAL_RANK): the card the nccl process group is bound to.  An explicit
    'cuda:N' is honoured (single-process use, or tests that share one card between ranks)."""
    if device is None or str(device) == "cuda":
        return f"cuda:{local_device_index()}"
    return str(device)


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


PAD_KEYPOINT_TAIL, PAD_KEYPOINT_HEAD = -1.0, -2.0    # never equal to a pixel coordinate, nor to each other


def pack_boundary(chunk: Dict[str, torch.Tensor], overlap: int, K: int, device="cpu") -> torch.Tensor:
    """Flat fp32 boundary block of one chunk: head (first ov views) and tail (last ov views) of keypoints (as fp16 bit
    patterns widened to fp32-exact integers), world points (fp32 values), validity, and the last camera pose.
    Layout: [n_frames, head_kp(ov*K*2), head_pts(ov*K*3), head_mask(ov*K), tail_kp, tail_pts, tail_mask, pose(16)].
    K is the wave-wide keypoint count (the block size every rank must agree on before the all-gather); a chunk with
    fewer keypoints fills its first K_local slots per view and pads the rest with keypoints that cannot match
    (-1 in a tail block, -2 in a head block: a tail is only ever matched against a head), zero points, mask 0."""
    n = int(chunk["points"].shape[0])
    ov = min(overlap, n)
    Kl = int(chunk["keypoints"].shape[1])
    if Kl > K:
        raise ValueError(f"chunk has {Kl} keypoints per view, the wave-wide block holds {K}")

    def blk(sl, pad_value):
        kp = torch.full((overlap, K, 2), pad_value, dtype=torch.float16)
        kp[:ov, :Kl] = chunk["keypoints"][sl].to(torch.float16)
        pt = torch.zeros(overlap, K, 3)
        pt[:ov, :Kl] = chunk["points"][sl].to(torch.float32)      # fp16 file values are exact in fp32; refined stay fp32
        mk = torch.zeros(overlap, K, 1)
        mk[:ov, :Kl] = chunk["masks"][sl].reshape(ov, Kl, 1).to(torch.float32)
        return torch.cat([kp.view(torch.int16).to(torch.float32).reshape(-1), pt.reshape(-1), mk.reshape(-1)])

    # assembled on the host, one pinned asynchronous upload (alignment.upload): nothing here waits for the device
    flat = torch.cat([torch.tensor([float(n)]), blk(slice(0, ov), PAD_KEYPOINT_HEAD),
                      blk(slice(n - ov, n), PAD_KEYPOINT_TAIL),
                      chunk["camera_poses"][n - 1].reshape(-1).to(torch.float32).cpu()])
    from .alignment import upload
    return upload(flat, device)


def boundary_numel(overlap: int, K: int) -> int:
    return 1 + 2 * overlap * K * 6 + 16


def unpack_boundary(flat: torch.Tensor, overlap: int, K: int, n_frames: Optional[int] = None) -> Dict[str, torch.Tensor]:
    """Views into the gathered block, on whatever device it sits (no copy).  Pass n_frames when it is already known on
    the host (it travels with the per-wave size exchange) to avoid a device sync."""
    n = int(flat[0].item()) if n_frames is None else int(n_frames)
    sz = overlap * K * 6

    def blk(t):
        kp = t[: overlap * K * 2].to(torch.int16).view(torch.float16).reshape(overlap, K, 2)
        pt = t[overlap * K * 2: overlap * K * 5].reshape(overlap, K, 3)
        mk = t[overlap * K * 5:].reshape(overlap, K, 1) > 0.5
        return dict(keypoints=kp, points=pt, masks=mk)

    return dict(n_frames=n, head=blk(flat[1: 1 + sz]), tail=blk(flat[1 + sz: 1 + 2 * sz]),
                last_pose=flat[1 + 2 * sz: 1 + 2 * sz + 16].reshape(4, 4))


def repad_tail(block: Dict, K: int) -> Dict:
    """The carried block of the previous wave re-padded to a wider keypoint count K (only its TAIL is ever read again):
    keypoints that cannot match, zero points, mask 0 - the same padding pack_boundary() writes."""
    tail = block["tail"]
    ov, K0 = tail["keypoints"].shape[:2]
    if K0 == K:
        return block
    if K0 > K:
        raise ValueError(f"carried block holds {K0} keypoints per view, cannot shrink to {K}")
    kp = torch.full((ov, K, 2), PAD_KEYPOINT_TAIL, dtype=tail["keypoints"].dtype, device=tail["keypoints"].device)
    kp[:, :K0] = tail["keypoints"]
    pt = torch.zeros((ov, K, 3), dtype=tail["points"].dtype, device=tail["points"].device)
    pt[:, :K0] = tail["points"]
    mk = torch.zeros((ov, K, 1), dtype=tail["masks"].dtype, device=tail["masks"].device)
    mk[:, :K0] = tail["masks"]
    return dict(block, tail=dict(keypoints=kp, points=pt, masks=mk), head=None)


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
    # the pairs are one run of consecutive views on both sides: slices, no index tensor to upload
    ri = slice(pairs[0][0], pairs[0][0] + len(pairs))
    qi = slice(pairs[0][1], pairs[0][1] + len(pairs))
    from .alignment import upload
    ref, qry = prev["tail"], cur["head"]
    kp_r = upload(ref["keypoints"][ri], device).contiguous()
    kp_q = upload(qry["keypoints"][qi], device).contiguous()
    idx = ops.sim3_match_keypoints(kp_r, kp_q)
    return ops.sim3_umeyama(upload(ref["points"][ri], device).contiguous(), upload(qry["points"][qi], device).contiguous(),
                            idx, upload(prev["last_pose"], device, torch.float32).contiguous(), None, None, use_filter)


def compose_global(rel: torch.Tensor) -> torch.Tensor:
    """rel: [n, 16] f64 relative similarities (rel[0] = identity for the first chunk) -> global G [n, 16]."""
    from . import ops
    return ops.sim3_compose_prefix(rel.contiguous())


# ---------------------------------------------------------------------------------------------------- wave alignment
def default_solver(overlap: int, device, chunk_length: Optional[int]):
    """solve(prev_block, cur_block) -> f64 [17] = [accepted, T(16)] on `device`, with the device kernels."""
    from .alignment import sim3_accepted

    eye = {}

    def solve(prev: Dict, cur: Dict) -> torch.Tensor:
        out = relative_sim3_from_boundaries(prev, cur, overlap, device, chunk_length=chunk_length)
        if out.device.type == "cpu":
            ok = sim3_accepted(out)
            return torch.cat([torch.tensor([1.0 if ok else 0.0], dtype=torch.float64),
                              out[13:29] if ok else torch.eye(4, dtype=torch.float64).reshape(16)])
        # the acceptance rule of alignment.sim3_accepted evaluated on the device: the record goes straight into the
        # all-gather, the host reads it once for the whole wave
        if "I" not in eye:
            eye["I"] = torch.eye(4, dtype=torch.float64, device=out.device).reshape(16)
        ok = (out[29] >= 3) & torch.isfinite(out[:29]).all()
        return torch.cat([ok.to(torch.float64).reshape(1), torch.where(ok, out[13:29], eye["I"])])
    return solve


def align_wave(rank: int, world: int, w0: int, n_chunks: int, blocks: List[Dict], prev_tail: Optional[Dict],
               G_last: torch.Tensor, solve, comm_device="cpu", compose=None) -> Tuple[List[torch.Tensor], List[bool]]:
    """One wave of the chunk-parallel progressive alignment.  blocks[r] = unpacked boundary block of chunk w0 + r
    (every rank holds all of them after the boundary all-gather).  Rank r solves ONLY its own T_{c-1<-c}
    (c = w0 + r; the predecessor's block is blocks[r-1], or prev_tail = the last block of the previous wave for r = 0);
    a second all-gather of 17 doubles per rank (136 B) distributes [accepted, T]; every rank then forms the global
    transforms of the wave by the prefix product G_c = G_{c-1} . T_c.

    Failure policy (same as the sequential path and the reference, offline_reconstructor.py:100-102): a chunk whose
    solve is rejected stays in its own frame, G_c = I, and the chunks after it chain onto it.
    Returns ([G_c for the chunks of this wave] as f64 4x4 CPU tensors, [accepted flags])."""
    from .alignment import upload
    c = w0 + rank
    pred = blocks[rank - 1] if rank > 0 else prev_tail
    eye = torch.eye(4, dtype=torch.float64).reshape(16)
    if c < n_chunks and pred is not None:
        try:
            mine = upload(solve(pred, blocks[rank]), comm_device)
        except Exception as e:  # noqa: BLE001 - a rank that raised here would skip the collective below and every
            # other rank would block in it until the RCCL timeout: the failure becomes a 'rejected' record instead
            # (the reference reports and carries on, offline_reconstructor.py:100-102)
            print(f"❌ rank {rank}: Sim(3) solve of chunk {c} failed ({type(e).__name__}: {e}); chunk left unaligned")
            mine = upload(torch.cat([torch.zeros(1, dtype=torch.float64), eye]), comm_device)
    else:
        mine = upload(torch.cat([torch.ones(1, dtype=torch.float64), eye]), comm_device)
    rel = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(rel, mine.contiguous())
    n_wave = min(world, n_chunks - w0)
    rel_cpu = torch.stack(rel[:n_wave]).cpu()
    Gs: List[torch.Tensor] = []
    oks: List[bool] = []
    base = G_last.to("cpu", torch.float64).reshape(4, 4).clone()
    pending: List[torch.Tensor] = []

    def flush():
        # prefix product of the pending relative transforms onto `base`; `compose` = the device kernel
        # (pi3_sim3_compose_prefix) when a GPU is present, else plain f64 4x4 products (CPU tests under gloo)
        nonlocal base, pending
        if pending:
            stack = torch.stack([base.reshape(16)] + pending).contiguous()
            out = (compose(stack) if compose is not None else _prefix_cpu(stack)).cpu()
            for g in out[1:]:
                Gs.append(g.reshape(4, 4).clone())
            base, pending = Gs[-1], []

    for r in range(n_wave):
        first_chunk = (w0 + r == 0)
        ok = bool(rel_cpu[r, 0] > 0.5)
        oks.append(ok or first_chunk)
        if first_chunk or not ok:      # restart the chain: this chunk keeps its own frame
            flush()
            Gs.append(torch.eye(4, dtype=torch.float64))
            base = Gs[-1]
        else:
            pending.append(rel_cpu[r, 1:])
    flush()
    return Gs, oks


def _prefix_cpu(T: torch.Tensor) -> torch.Tensor:
    out = [T[0].reshape(4, 4)]
    for i in range(1, T.shape[0]):
        out.append(out[-1] @ T[i].reshape(4, 4))
    return torch.stack([o.reshape(16) for o in out])


class WaveAligner:
    """Progressive alignment of chunk-parallel ranks, one wave (= `world` consecutive chunks) at a time; shared by the
    offline reconstructor and the online pipeline.  Every rank calls step() once per wave with its own chunk of the
    wave (or None past the end of the sequence) and gets back the global transforms of ALL chunks of the wave."""

    def __init__(self, rank: int, world: int, overlap: int, chunk_length: Optional[int], device, solve=None,
                 compose=None):
        self.rank, self.world, self.overlap, self.chunk_length = rank, world, overlap, chunk_length
        self.on_gpu = dist.get_backend() == "nccl"
        self.comm_dev = device if self.on_gpu else "cpu"
        if solve is None:
            from . import ops
            solve = default_solver(overlap, device, chunk_length)
            from .alignment import upload
            compose = lambda T: ops.sim3_compose_prefix(upload(T, device))   # noqa: E731
        self.solve, self.compose = solve, compose
        self.G_last = torch.eye(4, dtype=torch.float64)
        self.prev_tail: Optional[Dict] = None
        self.K_run = 0

    def step(self, chunk: Optional[Dict], w0: int, n_chunks: int) -> Tuple[List[torch.Tensor], List[bool]]:
        """chunk: this rank's chunk dict (chunk index w0 + rank) or None.  Collectives: sizes (2 ints per rank), boundary
        blocks (~50 KB per rank), [accepted, T] records (136 B per rank)."""
        from .alignment import upload
        sz = upload(torch.tensor([int(chunk["keypoints"].shape[1]), int(chunk["points"].shape[0])] if chunk is not None
                                 else [0, 0]), self.comm_dev)
        szs = [torch.zeros_like(sz) for _ in range(self.world)]
        dist.all_gather(szs, sz)
        szs = torch.stack(szs).cpu().tolist()          # one device -> host copy for the whole wave
        # K is run-wide: the widest keypoint count seen so far (every rank computes the same value from the gathered
        # sizes).  The block carried over from the previous wave was packed with THAT wave's K; with ragged keypoint
        # counts (low-texture frames, a short last wave) the two differ and the solver's shape check would raise on
        # rank 0 only - so the wave never shrinks below the carried block and the carried block is widened to the wave
        self.K_run = K = max(max(k for k, _ in szs), self.K_run)
        if self.prev_tail is not None:
            self.prev_tail = repad_tail(self.prev_tail, K)
        if chunk is not None:
            local = pack_boundary(chunk, self.overlap, K, device=self.comm_dev)
        else:   # ragged last wave: an empty block (n_frames = 0)
            local = torch.zeros(boundary_numel(self.overlap, K), device=self.comm_dev)   # a fill kernel, no copy
        blocks = [unpack_boundary(b, self.overlap, K, n_frames=szs[r][1])
                  for r, b in enumerate(allgather_boundaries(local, self.comm_dev))]
        Gs, oks = align_wave(self.rank, self.world, w0, n_chunks, blocks, self.prev_tail, self.G_last, self.solve,
                             self.comm_dev, self.compose)
        self.G_last, self.prev_tail = Gs[-1], blocks[len(Gs) - 1]
        return Gs, oks


# ------------------------------------------------------------------------------------------- sequential refinement chain
CHAIN_KEYS = ("points", "keypoints", "masks", "camera_poses", "_chunk_frame", "_sim3_global", "track_estimated")


def chain_payload(chunk: Optional[Dict]) -> Optional[Dict]:
    """What the alignment of chunk c + 1 reads from chunk c (alignment.align_and_refine_reconstructions,
    bundle_adjust.overlap_priors): host tensors only, ~0.5 MB at 100 views x 200 keypoints."""
    if chunk is None:
        return None
    out = {}
    for k in CHAIN_KEYS:
        if k in chunk and chunk[k] is not None:
            v = chunk[k]
            out[k] = {kk: vv.cpu() for kk, vv in v.items()} if isinstance(v, dict) else (v.cpu() if torch.is_tensor(v) else v)
    return out


def chain_step(payload: Optional[Dict], src: int) -> Optional[Dict]:
    """Hand the refined chunk c from its owner to every rank (the owner of chunk c + 1 needs it; a broadcast keeps the
    collective order identical on all ranks).  Used when bundle adjustment is on: the reference's refinement is a strictly
    sequential chain (align chunk c + 1 to the ALREADY REFINED chunk c, then adjust it with pose priors from c:
    slam/offline_reconstructor.py:130-133, utils/reconstruction_alignment.py:107-171), so the ranks take turns for it -
    stage 2 is 2-3 orders of magnitude cheaper than chunk creation, which stays chunk-parallel."""
    box = [payload]
    dist.broadcast_object_list(box, src=src)
    return box[0]
