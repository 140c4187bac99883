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


def test_random_test_subsets_batches_match_oracle():
    for seed in range(60):
        case, msg = fuzz_gpu.run_subsets_case(seed)
        assert msg is None, (msg, case)
