"""Device normalisation front-end (fw_normalize_counts, csrc/fw_norm.hip) against the host front-end (preprocess.py, itself
pinned on the reference's preprocessing fixtures in tests/test_preprocess_cpu.py).
Tolerance: both compute in Float64 and return Float32; device log vs numpy log can differ in the last Float64 bit, which
survives the rounding to Float32 in rare cases -> 1 ulp of Float32 (relative 1.2e-7) on the continuous modes; masks, shapes
and the discrete mode are exact."""
import numpy as np
import pytest

import flashweave_jl_amd as fw
from flashweave_jl_amd import preprocess as pre
from flashweave_jl_amd import synth
from tests.util import GOLDEN

pytestmark = pytest.mark.gpu


def _cases():
    raw = np.loadtxt(GOLDEN + "/HMP_SRA_gut_small.tsv", delimiter="\t", skiprows=1, usecols=range(1, 51)).astype(np.int64)
    yield "hmp", raw
    c = synth.generate(700, 333, 5, mode="S")
    c[:, 17] = 3          # a constant column
    c[40, :] = 0          # a sample without reads
    yield "synthS", c
    yield "synthF", synth.generate(400, 257, 6, mode="F", habitats=4)


@pytest.mark.parametrize("test_name", ["fz", "fz_nz", "mi"])
def test_device_normalisation_equals_host(test_name):
    for tag, counts in _cases():
        exp, erm, ecm = pre.normalize(counts, test_name, prec=32)
        got, rm, cm = fw.normalize_counts(counts, test_name)
        assert (rm == erm).all() and (cm == ecm).all(), tag
        assert got.shape == exp.shape, tag
        if test_name == "mi":
            assert np.array_equal(got, exp), tag
        else:
            assert np.allclose(got, exp, rtol=2.4e-7, atol=1e-7), (tag, np.abs(got - exp).max())
            assert (got == exp).mean() > 0.999, tag  # nearly every entry is the same Float32


def test_fixture_through_device_normalisation():
    # the reference's expected clr_adapt table (test/data/preprocessing_expected/clr_adapt.tsv, printed Float32 precision)
    raw = np.loadtxt(GOLDEN + "/HMP_SRA_gut_small.tsv", delimiter="\t", skiprows=1, usecols=range(1, 51)).astype(np.int64)
    got, _, _ = fw.normalize_counts(raw, "fz")
    exp = np.loadtxt(GOLDEN + "/clr_adapt.tsv")
    assert got.shape == exp.shape and np.allclose(got, exp, rtol=1e-5, atol=1e-5)
    got, _, _ = fw.normalize_counts(raw, "mi")
    assert np.array_equal(got, np.loadtxt(GOLDEN + "/pres_abs.tsv").astype(np.int64))
