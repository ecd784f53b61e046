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
