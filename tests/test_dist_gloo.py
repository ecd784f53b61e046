"""CPU, world_size 2, gloo: the chunk-parallel plumbing (sharding, boundary all-gather, who-needs-what)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_chunks, ov, K, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pi3_slam_amd.dist import allgather_boundaries, pack_boundary, shard_chunks, unpack_boundary
    mine = shard_chunks(n_chunks, rank, world)
    N = 6
    got = {}
    for wave in range((n_chunks + world - 1) // world):
        c = wave * world + rank
        g = torch.Generator().manual_seed(100 + c)          # chunk content is a function of the chunk id only
        ch = dict(points=torch.randn(N, K, 3, generator=g).half(), keypoints=(torch.rand(N, K, 2, generator=g) * 300).half(),
                  masks=torch.rand(N, K, 1, generator=g) > 0.5, camera_poses=torch.randn(N, 4, 4, generator=g))
        blocks = allgather_boundaries(pack_boundary(ch, ov, K), "cpu")
        for r, b in enumerate(blocks):
            cc = wave * world + r
            if cc < n_chunks:
                got[cc] = unpack_boundary(b, ov, K)["tail"]["points"].clone()
    dist.barrier()
    q.put((rank, mine, {k: v.float().sum().item() for k, v in got.items()}))
    dist.destroy_process_group()


def test_allgather_boundaries_world2():
    world, n_chunks, ov, K = 2, 4, 2, 5
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_chunks, ov, K, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    res.sort()
    assert res[0][1] == [0, 2] and res[1][1] == [1, 3]                    # chunk c on rank c % world
    assert res[0][2] == res[1][2] and sorted(res[0][2]) == [0, 1, 2, 3]   # every rank sees every chunk's boundary
    # and it is the right data: recompute chunk 3's tail checksum
    g = torch.Generator().manual_seed(103)
    pts = torch.randn(6, K, 3, generator=g).half()
    assert abs(res[0][2][3] - pts[-ov:].float().sum().item()) < 1e-6
