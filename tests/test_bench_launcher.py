"""`python bench.py --gpus N` must start N ranks itself when no launcher set
WORLD_SIZE (the way the round-end driver may call it for N > 1 is
torch.distributed.run, which sets it; a bare call must not silently run one
rank).  Exercised on CPU with the stub workload over gloo."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, timeout=240):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True,
                       text=True, timeout=timeout)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    return lines


@pytest.mark.parametrize("n", [2, 3])
def test_bare_call_spawns_n_ranks(n):
    lines = _run(["--gpus", str(n), "--stub", "--backend", "gloo", "--steps", "3", "--warmup", "1", "--quick"])
    assert len(lines) == 1, lines  # rank 0 alone prints
    res = json.loads(lines[0])
    assert res["n_gpus"] == n and res["ranks_seen"] == n and res["steps"] == 3 and res["warmup"] == 1


def test_single_rank_unchanged():
    lines = _run(["--gpus", "1", "--stub", "--steps", "2", "--warmup", "0"])
    assert len(lines) == 1 and json.loads(lines[0])["n_gpus"] == 1


def test_under_torch_distributed_run():
    """The driver's own launch line for N > 1: the ranks come from the launcher's environment."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
                        "--gpus", "2", "--stub", "--backend", "gloo", "--steps", "2", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0])["ranks_seen"] == 2
