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


# ------------------------------------------------------------------------------------------------ wave alignment
def _synthetic_chunks(n_chunks, cl, ov, K, bad_chunk=None, last_n=None, few_kp_chunk=None, kp_map=None):
    """Chunk dicts (chunk-file layout) cut from one world: chunk c = S_c^-1 (world).  bad_chunk's keypoints are shifted
    so that it shares no track with its predecessor (its alignment must fail and restart the chain)."""
    import numpy as np
    rng = np.random.default_rng(0)
    n_frames = cl + (n_chunks - 1) * (cl - ov)
    world_pts = rng.standard_normal((n_frames, K, 3)) + np.array([0, 0, 4.0])
    kp = (rng.random((K, 2)) * 300).astype(np.float16)
    chunks, sims = [], []
    for c in range(n_chunks):
        start = c * (cl - ov)
        n = min(cl, n_frames - start) if c < n_chunks - 1 else (last_n or cl - 2)      # short last chunk
        ang, s = 0.3 * c, 1.0 + 0.1 * c
        R = np.array([[np.cos(ang), -np.sin(ang), 0], [np.sin(ang), np.cos(ang), 0], [0, 0, 1.0]])
        t = np.array([0.5 * c, -0.3 * c, 0.1 * c])
        pts = (((world_pts[start:start + n] - t) @ R) / s).astype(np.float16)
        poses = np.tile(np.eye(4, dtype=np.float32), (n, 1, 1))
        poses[:, :3, 3] = ((np.stack([[0.1 * (start + j), 0, 0] for j in range(n)]) - t) @ R) / s
        k = np.tile(kp, (n, 1, 1))
        if c == bad_chunk:
            k[:ov] = (k[:ov].astype(np.float32) + 1000).astype(np.float16)     # only its head: the tail still pairs
        if c == few_kp_chunk:       # a chunk with fewer keypoints per view than the rest of its wave
            pts, k = pts[:, : K - 7], k[:, : K - 7]
        if kp_map and c in kp_map:  # ragged keypoint counts per chunk (ALIKED on low-texture frames)
            pts, k = pts[:, : kp_map[c]], k[:, : kp_map[c]]
        chunks.append(dict(points=torch.from_numpy(pts), keypoints=torch.from_numpy(k),
                           masks=torch.ones(n, pts.shape[1], 1, dtype=torch.bool), camera_poses=torch.from_numpy(poses)))
        M = np.eye(4); M[:3, :3] = s * R; M[:3, 3] = t
        sims.append(M)
    return chunks, sims


def _oracle_solver(ov, cl):
    """[accepted, T] with the CPU oracle (tests may use oracle/; the product's default solver is the HIP kernel)."""
    import numpy as np
    from oracle import post_ref

    def solve(prev, cur):
        n_prev, n_cur = prev["n_frames"], cur["n_frames"]
        ov_p = min(ov, n_prev)
        d = (cl - ov) - (n_prev - ov_p)
        pairs = [(i + d, i) for i in range(ov) if 0 <= i + d < ov_p and i < min(ov, n_cur)]
        ri, qi = [r for r, _ in pairs], [q for _, q in pairs]
        ref, qry = prev["tail"], cur["head"]
        res = torch.zeros(17, dtype=torch.float64)
        res[1:] = torch.eye(4, dtype=torch.float64).reshape(16)
        try:
            out = post_ref.align_chunks(ref["points"][ri].numpy(), qry["points"][qi].numpy(), ref["keypoints"][ri].numpy(),
                                        qry["keypoints"][qi].numpy(), prev["last_pose"].numpy(), True)
        except Exception:  # noqa: BLE001 - empty pair set
            return res
        if out["n_used"] >= 3 and np.isfinite(out["M"]).all():
            res[0] = 1.0
            res[1:] = torch.from_numpy(out["M"].reshape(16))
        return res
    return solve


def _shape_checking(solve, raise_on_chunk_with_kp=None):
    """The device solver's preconditions (ops.sim3_match_keypoints asserts kp_ref.shape == kp_qry.shape) on top of the
    oracle solver; optionally a solver that raises for one chunk (recognised by its keypoint count)."""
    def checked(prev, cur):
        assert prev["tail"]["keypoints"].shape == cur["head"]["keypoints"].shape, \
            (prev["tail"]["keypoints"].shape, cur["head"]["keypoints"].shape)
        if raise_on_chunk_with_kp is not None and int((cur["head"]["keypoints"][0, :, 0] > -0.5).sum()) == raise_on_chunk_with_kp:
            raise RuntimeError("solver failure injected by the test")
        return solve(prev, cur)
    return checked


def _wave_worker(rank, world, port, n_chunks, cl, ov, K, bad, q, last_n=None, few_kp=None, kp_map=None, raise_kp=None):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pi3_slam_amd.dist import WaveAligner
    chunks, _ = _synthetic_chunks(n_chunks, cl, ov, K, bad, last_n, few_kp, kp_map)
    # the product's wave driver (sizes all-gather, boundary all-gather, own solve, 136-byte all-gather, prefix product)
    # with the CPU oracle standing in for the HIP solver
    solve = _oracle_solver(ov, cl)
    if kp_map is not None:
        solve = _shape_checking(solve, raise_kp)
    aligner = WaveAligner(rank, world, ov, cl, "cpu", solve=solve)
    Gall, okall = [], []
    for w0 in range(0, n_chunks, world):
        c = w0 + rank
        Gs, oks = aligner.step(chunks[c] if c < n_chunks else None, w0, n_chunks)
        Gall += Gs
        okall += oks
    dist.barrier()
    q.put((rank, torch.stack(Gall).numpy(), okall))
    dist.destroy_process_group()


@pytest.mark.parametrize("n_chunks,bad", [(5, None), (4, 2)])
def test_align_wave_world2_equals_sequential_composition(n_chunks, bad):
    """World-size-2 chunk-parallel alignment (each rank solves its own T, 136-byte all-gather, prefix product with a
    restart at a rejected chunk, ragged last wave) == the sequential chain computed in one process."""
    import numpy as np
    world, cl, ov, K = 2, 8, 3, 30
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_wave_worker, args=(r, world, port, n_chunks, cl, ov, K, bad, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert np.array_equal(res[0][1], res[1][1]) and res[0][2] == res[1][2]       # every rank holds the same transforms
    G, oks = res[0][1], res[0][2]
    assert len(G) == n_chunks
    # sequential reference in this process
    from pi3_slam_amd.dist import pack_boundary, unpack_boundary
    chunks, sims = _synthetic_chunks(n_chunks, cl, ov, K, bad)
    solve = _oracle_solver(ov, cl)
    blocks = [unpack_boundary(pack_boundary(ch, ov, K), ov, K) for ch in chunks]
    Gseq = [np.eye(4)]
    for c in range(1, n_chunks):
        r = solve(blocks[c - 1], blocks[c])
        Gseq.append(Gseq[-1] @ r[1:].numpy().reshape(4, 4) if r[0] > 0.5 else np.eye(4))
    np.testing.assert_allclose(G, np.stack(Gseq), rtol=1e-12, atol=1e-12)
    assert oks == [c == 0 or c != bad for c in range(n_chunks)]
    # and the chain recovers the ground-truth similarities (chunk 0 = world frame up to S_0 = identity)
    for c in range(n_chunks):
        base = c if bad is None or c < bad else None
        if base is not None:
            np.testing.assert_allclose(G[c], sims[c], atol=2e-2)
        else:     # chunks from the rejected one on live in the rejected chunk's frame
            np.testing.assert_allclose(G[c], np.linalg.inv(sims[bad]) @ sims[c], atol=2e-2)


def test_align_wave_world8_two_waves_restart_on_wave_boundary_and_ragged_keypoints():
    """north_star's 8 ranks (gloo on the CPU): 13 chunks = a full wave of 8 + a ragged wave of 5; chunk 8 - the FIRST
    chunk of the second wave, whose predecessor block is rank 0's `prev_tail` carried over from the previous wave - is
    rejected, so the chain restarts there; the last chunk is short (4 of 10 frames, the 40-of-100 case of a
    1000-frame sequence); chunk 10 has fewer keypoints per view than its wave (pack_boundary pads to
    the wave-wide block).  Every rank must hold the same transforms, equal to the sequential composition."""
    import numpy as np
    world, n_chunks, cl, ov, K, bad, last_n, few = 8, 13, 10, 3, 30, 8, 4, 10
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_wave_worker, args=(r, world, port, n_chunks, cl, ov, K, bad, q, last_n, few))
             for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in range(1, world):
        assert np.array_equal(res[0][1], res[r][1]) and res[0][2] == res[r][2]
    G, oks = res[0][1], res[0][2]
    assert len(G) == n_chunks
    from pi3_slam_amd.dist import pack_boundary, unpack_boundary
    chunks, sims = _synthetic_chunks(n_chunks, cl, ov, K, bad, last_n, few)
    solve = _oracle_solver(ov, cl)
    blocks = [unpack_boundary(pack_boundary(ch, ov, K), ov, K) for ch in chunks]
    Gseq = [np.eye(4)]
    for c in range(1, n_chunks):
        r = solve(blocks[c - 1], blocks[c])
        Gseq.append(Gseq[-1] @ r[1:].numpy().reshape(4, 4) if r[0] > 0.5 else np.eye(4))
    np.testing.assert_allclose(G, np.stack(Gseq), rtol=1e-12, atol=1e-12)
    assert oks == [c != bad for c in range(n_chunks)]
    for c in range(n_chunks):
        want = sims[c] if c < bad else np.linalg.inv(sims[bad]) @ sims[c]
        np.testing.assert_allclose(G[c], want, atol=3e-2)


@pytest.mark.parametrize("kp_map,raise_kp", [({2: 20, 3: 20, 4: 26}, None), ({2: 40, 3: 36}, None), ({2: 20, 3: 21, 4: 22}, 21)])
def test_wave_keypoint_count_changes_between_waves(kp_map, raise_kp):
    """ADVICE r3 (dist.py:268): the block carried from wave w into wave w + 1 was packed with wave w's K.  With ragged
    keypoint counts the next wave's K differs - smaller ({2: 20, 3: 20}: the wave must not shrink below the carried
    block) or larger ({2: 40, ...}: the carried block must be widened) - and the device solver's shape check raised on
    rank 0 only, BEFORE the 136-byte all-gather, leaving every other rank blocked in it.  K is now run-wide and the
    carried block is re-padded; a solver that still raises (third case: injected for chunk 3) yields a 'rejected'
    record instead of a missing collective.  Results equal the sequential composition."""
    import numpy as np
    world, n_chunks, cl, ov, K = 2, 5, 8, 3, 30
    kmax = max(K, max(kp_map.values()))
    full = {c: K for c in range(n_chunks)}
    full.update(kp_map)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_wave_worker, args=(r, world, port, n_chunks, cl, ov, kmax, None, q, None, None, full, raise_kp))
             for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert np.array_equal(res[0][1], res[1][1]) and res[0][2] == res[1][2]
    G, oks = res[0][1], res[0][2]
    from pi3_slam_amd.dist import pack_boundary, unpack_boundary
    chunks, sims = _synthetic_chunks(n_chunks, cl, ov, kmax, kp_map=full)
    assert [int(ch["keypoints"].shape[1]) for ch in chunks] == [full[c] for c in range(n_chunks)]
    solve = _oracle_solver(ov, cl)
    blocks = [unpack_boundary(pack_boundary(ch, ov, kmax), ov, kmax) for ch in chunks]
    bad = [c for c in range(n_chunks) if raise_kp is not None and full[c] == raise_kp]
    Gseq = [np.eye(4)]
    for c in range(1, n_chunks):
        r = solve(blocks[c - 1], blocks[c])
        Gseq.append(Gseq[-1] @ r[1:].numpy().reshape(4, 4) if (r[0] > 0.5 and c not in bad) else np.eye(4))
    np.testing.assert_allclose(G, np.stack(Gseq), rtol=1e-12, atol=1e-12)
    assert oks == [c not in bad for c in range(n_chunks)]


def test_repad_tail():
    from pi3_slam_amd.dist import pack_boundary, repad_tail, unpack_boundary
    g = torch.Generator().manual_seed(1)
    n, K, ov = 6, 5, 3
    ch = dict(points=torch.randn(n, K, 3, generator=g).half(), keypoints=(torch.rand(n, K, 2, generator=g) * 300).half(),
              masks=torch.ones(n, K, 1, dtype=torch.bool), camera_poses=torch.randn(n, 4, 4, generator=g))
    b = unpack_boundary(pack_boundary(ch, ov, K), ov, K)
    wide = repad_tail(b, 9)
    want = unpack_boundary(pack_boundary(ch, ov, 9), ov, 9)
    for k in ("keypoints", "points", "masks"):
        assert torch.equal(wide["tail"][k], want["tail"][k]), k
    assert wide["n_frames"] == b["n_frames"] and torch.equal(wide["last_pose"], b["last_pose"])
    assert repad_tail(b, K) is b
    with pytest.raises(ValueError):
        repad_tail(b, K - 1)


def test_pack_boundary_pads_a_chunk_with_fewer_keypoints():
    """dist.py pack_boundary: the block is sized by the wave-wide K; a chunk with fewer keypoints must fill its own
    slots and pad the rest with keypoints that can never pair (it used to raise inside a collective wave)."""
    from pi3_slam_amd.dist import boundary_numel, pack_boundary, unpack_boundary
    g = torch.Generator().manual_seed(0)
    n, Kl, K, ov = 6, 5, 9, 3
    ch = dict(points=torch.randn(n, Kl, 3, generator=g).half(), keypoints=(torch.rand(n, Kl, 2, generator=g) * 300).half(),
              masks=torch.ones(n, Kl, 1, dtype=torch.bool), camera_poses=torch.randn(n, 4, 4, generator=g))
    flat = pack_boundary(ch, ov, K)
    assert flat.numel() == boundary_numel(ov, K)
    b = unpack_boundary(flat, ov, K)
    assert torch.equal(b["tail"]["keypoints"][:, :Kl], ch["keypoints"][-ov:])
    assert torch.equal(b["head"]["points"][:, :Kl], ch["points"][:ov].float())
    assert not b["tail"]["masks"][:, Kl:].any() and not b["head"]["masks"][:, Kl:].any()
    pad_t, pad_h = b["tail"]["keypoints"][:, Kl:], b["head"]["keypoints"][:, Kl:]
    assert (pad_t < 0).all() and (pad_h < 0).all() and not (pad_t == pad_h).any()
    with pytest.raises(ValueError):
        pack_boundary(ch, ov, Kl - 1)
