import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
# the library reads its tuning / test knobs (FW_HOST_HITON, FW_DEV_MIN_TARGETS, FW_DH_*, ...) only when FW_KNOBS=1 is set
# (csrc/fw_internal.h: fw_knob); the tests switch code paths with them, also in the subprocesses they start
os.environ.setdefault("FW_KNOBS", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def _gpu_available():
    """True when the HIP path can run: the in-tree library loads and a gfx950 context can be created."""
    try:
        import flashweave_jl_amd as fw
        eng = fw.Engine("fz", 32, 4, max_k=0)
        eng.close()
        return True
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    """Plain `pytest tests` on a host without a GPU: gpu-marked tests are skipped instead of failing with FW_ERR_DEVICE
    (an explicit `-m gpu` run still fails loudly there -- a GPU box without a usable device must not look green)."""
    if "gpu" in (config.getoption("-m") or ""):
        return
    gpu_items = [it for it in items if it.get_closest_marker("gpu")]
    if gpu_items and not _gpu_available():
        skip = pytest.mark.skip(reason="no gfx950 device (the HIP path has no CPU fallback)")
        for it in gpu_items:
            it.add_marker(skip)
