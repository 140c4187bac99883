"""world_size-2 and -4 gloo tests of the per-round neighbour-set exchange (the only collective of the path)."""
import json
import os
import subprocess
import sys

from tests.util import ROOT


import pytest


def _launch(mode, extra=(), world=2):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29653 + world))
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_worker.py"), mode, str(r), str(world), *extra],
                              env=env) for r in range(world)]
    return [p.wait(timeout=900) for p in procs]


@pytest.mark.parametrize("world", [2, 4])
def test_allgather_callback_gloo(world):
    assert _launch("callback", world=world) == [0] * world


def test_bench_launcher_spawns_n_ranks():
    """`python bench.py --gpus N` starts N ranks itself (--gpus was parsed and never read).  --spawn-check
    stops after the process group is up: every rank contributes a 1 to an all-reduce and rank 0 prints the sum."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    for n in (2, 3):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--spawn-check", "--backend", "gloo"],
                           env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1, r.stdout  # ONE JSON line, from rank 0
        js = json.loads(lines[0])
        assert js["n_ranks"] == n and js["world_size"] == n


def test_bench_refuses_world_size_mismatch():
    """Under torch.distributed.run with WORLD_SIZE != --gpus the script refuses instead of reporting the wrong n_gpus."""
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--spawn-check"], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 2 and "refusing" in r.stderr


def test_bench_refuses_more_ranks_than_gpus():
    """No GPU here: --gpus 2 must fail loudly, not run fewer ranks."""
    import torch
    if torch.cuda.device_count() >= 2:
        return
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 2 and "refusing" in r.stderr
