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


def test_meta_variables_onehot_golden():
    """reference test/preprocessing.jl:144-190: the 19-sample table + six meta variables (three numeric, two string factors
    with three categories each, one continuous) through the front-end for all four test kinds; the encoded meta block must
    equal the reference's fixture meta_tiny_oneHotTest.tsv (with the +1 shift undone for fz_nz)."""
    import os

    from flashweave_jl_amd import preprocess as pre
    from tests.util import GOLDEN
    rows = [ln.rstrip("\n").split("\t") for ln in open(os.path.join(GOLDEN, "HMP_SRA_gut_tiny.tsv"))]
    header, counts = rows[0], np.array(rows[1:], dtype=np.float64)
    mrows = [ln.rstrip("\n").split("\t") for ln in open(os.path.join(GOLDEN, "HMP_SRA_gut_tiny_meta_oneHotTest.tsv"))]
    mheader = mrows[0]
    meta = np.empty((len(mrows) - 1, len(mheader)), dtype=object)
    for i, r in enumerate(mrows[1:]):
        for j, v in enumerate(r):
            try:
                meta[i, j] = float(v)
            except ValueError:
                meta[i, j] = v
    erows = [ln.rstrip("\n").split("\t") for ln in open(os.path.join(GOLDEN, "meta_tiny_oneHotTest.tsv"))]
    eheader, exp = erows[0], np.array(erows[1:], dtype=np.float64)
    for test_name in ("fz", "mi", "fz_nz", "mi_nz"):
        r = pre.normalize_with_meta(counts, test_name, meta, header=header, meta_header=mheader)
        A = r["data"][:, r["meta_mask"]].astype(np.float64)[:, :-1]  # skip the continuous column for the identity test
        if test_name == "fz_nz":
            A = A - 1  # the +1 shift of one-hot variables in fz_nz
        assert np.array_equal(A, exp[r["row_mask"], :-1]), test_name
        assert r["meta_header"] == eheader
        if test_name.startswith("mi"):
            assert len(np.unique(r["data"][:, -1])) == 2  # the continuous meta variable was discretised into two bins
        assert r["data"].shape[1] == len(r["header"])
        r2 = pre.normalize_with_meta(counts, test_name, meta[:, :3], header=header, meta_header=mheader[:3], make_onehot=False)
        assert r2["meta_mask"].sum() == 3 and r2["data"].shape[1] == len(r2["header"])
