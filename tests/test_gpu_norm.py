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


@pytest.mark.parametrize("test_name", ["fz", "fz_nz", "mi", "mi_nz"])
def test_device_normalisation_equals_host(test_name):
    for tag, counts in _cases():
        exp, erm, ecm = pre.normalize(counts, test_name, prec=32)
        got, rm, cm = fw.normalize_counts(counts, test_name)
        assert (rm == erm).all() and (cm == ecm).all(), tag
        assert got.shape == exp.shape, tag
        if test_name in ("mi", "mi_nz"):
            assert np.array_equal(got, exp), tag
            if test_name == "mi_nz":
                assert set(np.unique(got)) == {0, 1, 2}, tag
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
    # binned_nz_clr (test/data/preprocessing_expected/clr_nonzero_binned.tsv): per-column tied ranks of the non-zeros on the device
    got, _, _ = fw.normalize_counts(raw, "mi_nz")
    assert np.array_equal(got, np.loadtxt(GOLDEN + "/clr_nonzero_binned.tsv").astype(np.int64))


def test_binned_nz_clr_ties_and_duplicated_samples():
    # tied ranks: duplicated samples give every column pairs of equal clr values (average ranks), a few-valued table gives
    # long runs of ties; the device bins must equal the host front-end's (scipy rankdata, "average")
    rng = np.random.default_rng(3)
    base = rng.poisson(3.0, size=(150, 60)) * (rng.random((150, 60)) < 0.6)
    counts = np.concatenate([base, base[:50]], axis=0)   # 50 duplicated samples
    counts[:, 5] = (np.arange(200) % 2) * 7               # a two-valued column: all non-zeros tie
    exp, erm, ecm = pre.normalize(counts, "mi_nz")
    got, rm, cm = fw.normalize_counts(counts, "mi_nz")
    assert (rm == erm).all() and (cm == ecm).all() and np.array_equal(got, exp)


def test_binned_nz_clr_beyond_16384_samples():
    # the column sort of binned_nz_clr leaves LDS above 16 384 samples (keys in device memory, workgroups striding over the columns):
    # 20 000 samples incl. duplicated ones (ties) against the host front-end (preprocessing.jl:217-291 has no bound on n)
    rng = np.random.default_rng(11)
    base = rng.poisson(2.0, size=(12_000, 40)) * (rng.random((12_000, 40)) < 0.5)
    counts = np.concatenate([base, base[:8_000]], axis=0)
    exp, erm, ecm = pre.normalize(counts, "mi_nz")
    got, rm, cm = fw.normalize_counts(counts, "mi_nz")
    assert exp.shape[0] > 16384
    assert (rm == erm).all() and (cm == ecm).all() and got.shape == exp.shape
    # A tied rank hangs on the last bit of the row's geometric mean: two rows with the same non-zero counts in another order have
    # the same mean in exact arithmetic and means one ulp apart in any floating-point summation, so log(x / g) ties on one side and
    # not on the other (half a rank; the reference's own summation order is as arbitrary).  With 20 000 rows a handful of entries sit
    # on such a near-tie AND on the bin boundary: every difference must be one of them.
    bad = np.argwhere(got != exp)
    assert len(bad) <= 20
    x = counts[erm][:, ecm].astype(np.float64)
    g = np.array([np.exp(np.log(r[r != 0]).mean()) if (r != 0).any() else 1.0 for r in x])
    for r, q in bad:
        same = (x[:, q] == x[r, q])
        rel = np.abs(g[same] - g[r]) / g[r]
        assert ((rel > 0) & (rel < 1e-12)).any(), (r, q, got[r, q], exp[r, q])


def test_learn_network_uses_the_device_front_end():
    # learn_network(normalize=True) on a count table: the default path normalises on the device (no host pre.normalize) and
    # gives the network of the host front-end + engine composition
    import flashweave_jl_amd.api as api
    raw = np.loadtxt(GOLDEN + "/HMP_SRA_gut_small.tsv", delimiter="\t", skiprows=1, usecols=range(1, 51)).astype(np.int64)
    for kw in (dict(sensitive=True, heterogeneous=False), dict(sensitive=False, heterogeneous=True)):
        calls = []
        orig = api.pre.normalize
        api.pre.normalize = lambda *a, **k: calls.append(1) or orig(*a, **k)
        try:
            r1 = fw.learn_network(raw, max_k=3, **kw)
        finally:
            api.pre.normalize = orig
        assert not calls
        r2 = fw.learn_network(raw, max_k=3, device_normalize=False, **kw)
        assert set(r1["edges"]) == set(r2["edges"])
        assert len(r1["edges"]) > 0 or not kw["sensitive"]   # the bundled table gives FlashWeaveHE-F one edge at most
        for e, w in r2["edges"].items():
            assert abs(r1["edges"][e] - w) <= 5e-5   # Float32 last-bit differences of the clr values (module docstring)
        assert r1["counters"]["t_normalize_s"] > 0 and r1["counters"]["normalized_on_device"] and not r2["counters"]["normalized_on_device"]
