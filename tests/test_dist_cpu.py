"""world_size-2 gloo test of the per-round neighbour-set exchange (the only collective of the path)."""
import os
import subprocess
import sys

from tests.util import ROOT


def _launch(mode, extra=()):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29653")
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_worker.py"), mode, str(r), "2", *extra],
                              env=env) for r in range(2)]
    return [p.wait(timeout=600) for p in procs]


def test_allgather_callback_gloo_world2():
    assert _launch("callback") == [0, 0]
