"""A fixed handful of the randomised end-to-end parity cases of tests/fuzz_gpu.py (kind, shape, learn_network keywords
and schedule drawn from the seed; HIP path through the C ABI against the CPU oracle).  `python -m tests.fuzz_gpu
--cases N` runs as many as wanted; DESIGN.md section 2 records the last long sweep."""
import pytest

from tests import fuzz_gpu

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("first", [0, 40, 80])
def test_random_cases_match_oracle(first):
    kinds = set()
    for seed in range(first, first + 40):
        case, msg = fuzz_gpu.run_case(seed)
        assert msg is None, (msg, case)
        kinds.add(case["kind"])
    assert kinds == {"fz", "fz_nz", "mi", "mi_nz"}


def test_random_max_k_4_5_networks_match_oracle(monkeypatch):
    # whole Fisher-z networks with max_k 4 / 5: level-2 / level-3 table kernels in the device rounds, on the targets' local matrices (r06)
    monkeypatch.setattr(fuzz_gpu, "HIGHK", True)
    ks = set()
    for seed in range(650000, 650012):
        case, msg = fuzz_gpu.run_case(seed)
        assert msg is None, (msg, case)
        ks.add(case["max_k"])
    assert ks == {4, 5}


def test_random_test_subsets_batches_match_oracle():
    for seed in range(60):
        case, msg = fuzz_gpu.run_subsets_case(seed)
        assert msg is None, (msg, case)


def test_random_cases_with_concurrent_chains():
    # FW_DH_CHAINS deals the targets of a round to concurrent chains of device rounds (own host thread, stream, arena); the
    # default only does so from 512 targets on, so force it for the small random cases (the knobs are read once per
    # process -> subprocess)
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for chains, first in (("2", 300), ("3", 400)):
        env = dict(os.environ, FW_DH_CHAINS=chains, FW_DH_CHAINS_DISC=chains, FW_DH_CHAIN_MIN="4")
        out = subprocess.run([sys.executable, "-m", "tests.fuzz_gpu", "--first", str(first), "--cases", "100"], env=env, cwd=root,
                             capture_output=True, text=True)
        assert out.returncode == 0 and "100 cases, 0 failures" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
