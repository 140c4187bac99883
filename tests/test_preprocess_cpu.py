"""Normalisation front-end against the reference's fixtures test/data/preprocessing_expected/*.tsv
(reference test/preprocessing.jl:48-85), and the synthetic generator's determinism."""
import numpy as np

from flashweave_jl_amd import preprocess as pre
from flashweave_jl_amd import synth
from tests.util import GOLDEN, load_norm


def _raw():
    return np.loadtxt(GOLDEN + "/HMP_SRA_gut_small.tsv", delimiter="\t", skiprows=1, usecols=range(1, 51))


def test_binary_and_binned_exact():
    raw = _raw()
    for tn, fx in (("mi", "pres_abs"), ("mi_nz", "clr_nonzero_binned")):
        d, rm, cm = pre.normalize(raw, tn)
        e = load_norm(fx, np.int64)
        assert d.shape == e.shape == (346, 50) and (d == e).all()


def test_clr_adapt_matches_fixture():
    d, rm, cm = pre.normalize(_raw(), "fz", prec=64)
    e = load_norm("clr_adapt", np.float64)  # printed as Float32
    assert d.shape == e.shape
    assert (np.abs(d - e) / np.maximum(np.abs(e), 1e-30)).max() < 3e-7


def test_synth_deterministic_and_shaped():
    a = synth.generate(300, 120, 7, mode="F")
    b = synth.generate(300, 120, 7, mode="F")
    assert a.shape == (120, 300) and (a == b).all() and synth.checksum(a) == synth.checksum(b)
    c, m = synth.generate(100, 80, 3, mode="F", habitats=4, n_meta=20)
    assert m.shape == (80, 20) and set(np.unique(m)) <= {0, 1}
    assert ((m[:, :4].sum(axis=1)) == 1).all()
