"""The discrete HITON-PC driver is deterministic (hiton.jl:109-149, interleaved.jl:124-183 `single_il`; test/learning.jl:176-237 asserts
exact edge lists): whatever the persistent kernel's helpers, boards, team targets and look-ahead did in a pass, the learned network is
THE network.  Round 5 shipped a race in dh_mi_team (about one cfg2 pass in 1 000, 0.3 % with two processes on the GPU, once a hang);
a single pass per configuration -- what every other test runs -- cannot see such a defect.  This test repeats cfg2's pass in two
concurrent processes (contention is what varies the schedule) for a fixed time and demands that every pass of both processes is the
same bytes, and that those bytes are the oracle's network."""
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np
import pytest

from flashweave_jl_amd import preprocess as pre
from flashweave_jl_amd import synth
from oracle import oracle as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "determinism_worker.py")


def _run_pair(extra, seconds, tmp):
    start = os.path.join(tmp, "go")
    procs = []
    for w in range(2):
        cmd = [sys.executable, WORKER, "--passes", "1000000", "--seconds", str(seconds), "--start-file", start,
               "--save", os.path.join(tmp, "net%d.npz" % w)] + extra
        procs.append(subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=ROOT))
    t0 = time.time()
    while time.time() - t0 < 100:  # both workers have built their engine and warmed up: start together
        if len([f for f in os.listdir(tmp) if f.startswith("go.")]) >= 2:
            break
        time.sleep(0.05)
    open(start, "w").close()
    out = []
    for pr in procs:
        try:
            so, se = pr.communicate(timeout=seconds + 150)  # (a hang -- the r05 defect could end in one -- fails the test instead of the suite)
        except subprocess.TimeoutExpired:
            pr.kill()
            so, se = pr.communicate()
            raise AssertionError("determinism worker made no progress (hang): " + se[-2000:])
        assert pr.returncode == 0, se[-2000:]
        out.append(json.loads(so.strip().splitlines()[-1]))
    for f in os.listdir(tmp):
        if f.startswith("go"):
            os.remove(os.path.join(tmp, f))
    return out


@pytest.mark.parametrize("round_size", [-1, 256])
def test_discrete_network_is_deterministic_under_contention(round_size):
    # cfg2 (1 000 OTUs x 500 samples, mi, max_k 3): round_size -1 = bench.py's headline schedule (one feed-forward round: no whitelist yet),
    # 256 = four rounds with device-built whitelists
    with tempfile.TemporaryDirectory() as tmp:
        res = _run_pair(["--p", "1000", "--n", "500", "--seed", "20260930", "--kind", "mi", "--feed-forward", "1", "--round-size", str(round_size)],
                        20 if round_size < 0 else 12, tmp)
        for r in res:
            assert r["differing"] == 0, r
            assert r["passes"] >= 200, r  # (a pass is ~6 ms alone: far more in practice)
        assert res[0]["sha256"] == res[1]["sha256"]
        net = np.load(os.path.join(tmp, "net0.npz"))
        counts = synth.generate(1000, 500, 20260930, mode="F")
        data, _, _ = pre.normalize(counts, "mi")
        orc = O.Oracle("mi", np.ascontiguousarray(data), sparse=True, max_k=3)
        exp = orc.learn(max_k=3, feed_forward=True, round_size=1024 if round_size < 0 else round_size, threads=8)
        got = {(int(s), int(d)): float(w) for s, d, w in zip(net["edge_src"], net["edge_dst"], net["edge_weight"])}
        assert set(got) == set(exp["edges"])
        assert int(net["n_cond_tests"]) == exp["n_cond_tests"]
        for e, w in exp["edges"].items():
            assert abs(got[e] - w) <= 1e-10 * max(1.0, abs(w))


def test_discrete_network_independent_of_the_persistent_kernel_s_cuts():
    # The queue words next_target / targets_done switch how a job's ranks are cut into prefix / window / board records (tail mode), and the
    # team / look-ahead / four-per-step machinery decides which wavefront and which step evaluates a rank: the merge is in rank order, so
    # none of it may show in the network.  One process per setting (the knobs are read once), 30 passes each, all the same bytes.
    base = ["--p", "1000", "--n", "500", "--seed", "20260930", "--kind", "mi", "--feed-forward", "1", "--round-size", "256", "--passes", "30"]
    settings = [{}, {"FW_MI_TEAM_TAIL": "1"}, {"FW_MI_AHEAD": "0"}, {"FW_MI_ROW4": "0"}, {"FW_MI_TEAM_MAX": "0"}, {"FW_MI_HELP_JOBS": "0"},
                {"FW_MI_SEQ_TAIL": "1", "FW_MI_TEAM_TAIL": "1"}]
    procs = [subprocess.Popen([sys.executable, WORKER] + base, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=ROOT,
                              env=dict(os.environ, FW_KNOBS="1", **s)) for s in settings]
    seen = {}
    for s, pr in zip(settings, procs):
        so, se = pr.communicate(timeout=300)
        assert pr.returncode == 0, se[-2000:]
        r = json.loads(so.strip().splitlines()[-1])
        assert r["differing"] == 0 and r["passes"] == 30, (s, r)
        seen[json.dumps(s)] = (r["sha256"], r["sha256_integers"], r["edges"], r["ref_tests"])
    # edge list, directed lists and the reference-order test count: the same bytes under every setting
    assert len({v[1:] for v in seen.values()}) == 1, seen
    # ... and the statistics too, except where the setting swaps the ARITHMETIC FORM of the test (FW_MI_ROW4 = 0: one test per wavefront step
    # instead of four per step -- another summation order of the MI terms, equal to 1e-12: DESIGN.md section 2; the forms never mix within a launch)
    same_form = {k: v for k, v in seen.items() if "FW_MI_ROW4" not in k}
    assert len({v[0] for v in same_form.values()}) == 1, seen
