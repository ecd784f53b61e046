"""bench.py's N > 1 start-up, on the CPU (gloo): started as a plain process it launches its own ranks before anything
touches the GPU and relays rank 0's line; started under torch.distributed.run it uses the ranks it is given; a failing
rank makes the whole command fail.  `--launch-check` does the rendezvous + one all-gather and no chunk work."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env(**kw):
    env = dict(os.environ, **dict({"PI3_DIST_BACKEND": "gloo"}, **kw))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "PI3_BENCH_LAUNCHER"):
        env.pop(k, None)
    return env


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world", [2, 8])
def test_plain_process_launches_its_own_ranks(world):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--launch-check"],
                       env=_env(), capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and r.stdout.strip() == lines[0]          # ONE line on stdout, nothing else
    rec = json.loads(lines[0])
    comm = rec["comm"]
    assert rec["n_gpus"] == world and comm["world_size"] == world and comm["launcher"] == "self"
    assert [x["rank"] for x in comm["ranks"]] == list(range(world))
    assert [x["local_rank"] for x in comm["ranks"]] == list(range(world))
    assert len({x["pid"] for x in comm["ranks"]}) == world            # one process per rank


def test_under_torch_distributed_run_uses_the_given_ranks():
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"),
                        "--gpus", "2", "--launch-check"], env=_env(), capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    comm = json.loads(lines[0])["comm"]
    assert comm["world_size"] == 2 and comm["launcher"] == "torch.distributed.run"


def test_a_failing_rank_fails_the_command():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-check"],
                       env=_env(PI3_DIST_BACKEND="no_such_backend"), capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode != 0
    assert "[bench launcher] rank" in r.stderr and not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_world_size_mismatch_is_refused():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--launch-check"],
                       env=dict(_env(), RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port())),
                       capture_output=True, text=True, timeout=120, cwd=ROOT)
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr


@pytest.mark.parametrize("world,how", [(8, "plain"), (2, "torchrun")])
def test_full_main_control_flow_with_stub_engine(world, how):
    """`bench.py --gpus N` through ALL of main() - not only the rendezvous of --launch-check: warm-up, the timed loop with
    the boundary all-gather, every rank's own Sim(3) solve, the 136-byte record all-gather, the prefix composition, the
    barrier + max-over-ranks timing, the per-rank gather, rank 0's JSON line, the process-group teardown - with the GPU
    objects replaced by tests/bench_stub.py (PI3_BENCH_STUB=1, gloo).  An 8-GPU node has never been available to the
    builder: this is the only place where the Python of the first real 8-rank run executes beforehand.  The stub's
    chunks are cut from one synthetic world, so the composed global transforms of the last wave are checked as well."""
    import numpy as np
    args = ["--gpus", str(world), "--steps", "2", "--warmup", "1"]
    if how == "plain":
        cmd = [sys.executable, os.path.join(ROOT, "bench.py")] + args
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr",
               "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py")] + args
    r = subprocess.run(cmd, env=_env(PI3_BENCH_STUB="1"), capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["stub"] is True and "roofline" not in rec and rec["data"].startswith("STUB")      # never a measurement
    assert rec["n_gpus"] == world and rec["steps"] == 2 and rec["scaling"] == "weak" and rec["unit"] == "frames/s"
    comm = rec["comm"]
    assert comm["world_size"] == world and comm["backend"] == "gloo" and len(comm["per_rank"]) == world
    assert sorted(x["rank"] for x in comm["per_rank"]) == list(range(world))
    assert comm["launcher"] == ("self" if how == "plain" else "torch.distributed.run")
    assert rec["n1_reference_ms_per_step"]["ms_per_step"] > 0
    assert rec["value"] == pytest.approx(world * 100 * 2 / (rec["ms_per_step"] * 2e-3), rel=1e-6)     # whole-job frames / max time
    # the last timed wave: chunks world .. 2 world - 1; G_c = S_0^-1 S_c for the stub's known chunk frames
    sys.path.insert(0, os.path.join(ROOT, "tests"))

    def S(c):
        ang, s = 0.05 * (c % 7), 1.0 + 0.02 * (c % 5)
        M = np.eye(4)
        M[:3, :3] = s * np.array([[np.cos(ang), -np.sin(ang), 0], [np.sin(ang), np.cos(ang), 0], [0, 0, 1.0]])
        M[:3, 3] = [0.1 * (c % 3), -0.05 * (c % 4), 0.02 * c]
        return M
    got = np.array(rec["stub_check"]["global_transforms_last_wave"]).reshape(world, 4, 4)
    for r_, c in enumerate(range(world, 2 * world)):
        want = np.linalg.inv(S(0)) @ S(c)
        assert np.abs(got[r_] - want).max() < 2e-2, (c, got[r_], want)      # fp16 chunk-file points: ~1e-3 relative
