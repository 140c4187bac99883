"""Host-side logic of the Python mirror that needs no GPU: argument validation of learn_network / normalize_counts (reference
behaviour: Julia exceptions, src/learning.jl:466-598), the default schedule, and that the product never falls back to a CPU path."""
import numpy as np
import pytest

import flashweave_jl_amd as fw
from flashweave_jl_amd import api


def test_default_round_size():
    # small problems: the reference's single_il schedule (reproduces its golden networks by default)
    assert api.default_round_size(50) == 1 and api.default_round_size(512) == 1
    # beyond: eight to ten rounds per pass at every size (one round would silently switch feed_forward off) ...
    for p in (513, 1000, 5000, 8000, 10_000, 30_000, 50_020, 100_000):
        R = api.default_round_size(p)
        assert R >= 64 and 8 <= -(-p // R) <= 10, (p, R)
    # ... and at the benchmark sizes R = 1024 * ceil(p / 10240), bench.py's headline schedule
    assert api.default_round_size(10_000) == 1024 and api.default_round_size(10_241) == 1281
    assert api.default_round_size(50_020) == 5120 and api.default_round_size(100_000) == 10_240


def test_learn_network_rejects_what_it_would_otherwise_ignore():
    x = np.ones((10, 4), dtype=np.int64)
    with pytest.raises(TypeError):
        fw.learn_network(x, time_limit=30.0)              # unsupported option: raise, never silently ignore
    with pytest.raises(ValueError):
        fw.learn_network(x.astype(np.float32), normalize=False, meta_data=np.ones((10, 1)))   # the mask would be lost


def test_normalize_counts_validates_before_touching_the_device():
    with pytest.raises(TypeError):
        fw.normalize_counts(np.array([[0.5, 1.0], [2.0, 3.0]]), "fz")          # relative abundances: a silent int cast would zero them
    with pytest.raises(ValueError):
        fw.normalize_counts(np.array([[-1, 1], [2, 3]]), "fz")
    with pytest.raises(ValueError):
        fw.normalize_counts(np.array([[2 ** 40, 1], [2, 3]]), "fz")
    with pytest.raises(ValueError):
        fw.normalize_counts(np.arange(6), "fz")


def test_integral_float_tables_take_the_device_front_end():
    assert api._integral(np.array([[1.0, 0.0], [3.0, 2.0]])) and api._integral(np.array([[1, 0]], dtype=np.int32))
    assert not api._integral(np.array([[0.5, 1.0]])) and not api._integral(np.array([[np.nan, 1.0]]))
    # integer tables outside Int32 / with negative entries go to the host front-end too (they used to reach the device and raise)
    assert not api._integral(np.array([[2 ** 40, 1]], dtype=np.int64)) and not api._integral(np.array([[-1, 1]], dtype=np.int64))
    assert not api._integral(np.array([[2 ** 31, 1]], dtype=np.uint32)) and api._integral(np.array([[2 ** 31 - 1, 0]], dtype=np.uint32))
