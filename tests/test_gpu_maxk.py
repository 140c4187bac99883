"""Conditioning sets of 6 and 7 variables (r05, ABI 6: FW_MAX_K = 7).  The reference has no cap on max_k (tests.jl:311-343 enumerates
subsets of max_k ... 1 accepted variables, statfuns.jl:23-75 recurses to any depth, types.jl:98-117 sizes the tables L x L x L^max_k); the
table kernels, the device rounds and the persistent discrete kernel serve max_k <= 5, beyond that the host job pool drives general-form
kernels (fz / fz_nz: fz_subsets_slow_kernel, fz_test_batch_kernel<true>; discrete: the segment / batch kernels with the large LDS table).
Same tolerances as the other parity tests: integers and conditioning sets exact, Fisher-z statistics to the bit, p 1e-12; MI 1e-12, p 1e-10."""
import numpy as np
import pytest

import flashweave_jl_amd as fw
from flashweave_jl_amd import preprocess as pre
from flashweave_jl_amd import synth
from oracle import oracle as O
from tests.util import rel

pytestmark = pytest.mark.gpu


def _fz_data(p, n, seed):
    counts = synth.generate(p, n, seed, mode="S")
    data, _, _ = pre.normalize(counts, "fz", prec=32)
    return np.asfortranarray(data)


@pytest.mark.parametrize("max_k", [6, 7])
def test_fz_single_tests_and_subsets(max_k):
    data = _fz_data(120, 400, 31)
    n, p = data.shape
    eng = fw.Engine("fz", n, p, max_k=max_k)
    eng.set_data(data)
    cm = eng.cor()
    orc = O.Oracle("fz", cor_mat=cm, n_obs=n)
    rng = np.random.default_rng(max_k)
    X, Y, Zs = [], [], []
    for _ in range(600):
        k = int(rng.integers(0, max_k + 1))
        v = rng.choice(p, size=k + 2, replace=False)
        X.append(int(v[0])); Y.append(int(v[1])); Zs.append(tuple(int(t) for t in v[2:]))
    got = eng.test_batch(X, Y, Zs)
    for x, y, z, g in zip(X, Y, Zs, got):
        s, pv, df, pw = orc.test(x, y, z, n_obs_min=20)
        assert (g.stat == s) or (np.isnan(g.stat) and np.isnan(s)), (x, y, z, g.stat, s)
        assert rel(g.pval, pv) < 1e-12 or (np.isnan(g.pval) and np.isnan(pv))
    # test_subsets: short lists (whole enumerations at alpha close to 1, incl. lists shorter than max_k), a max_tests stop
    for alpha, max_tests in ((0.01, 10_000_000), (0.9999, 10_000_000), (0.9999, 3000)):
        e2 = fw.Engine("fz", n, p, max_k=max_k, alpha=alpha, max_tests=max_tests)
        e2.set_data(data)
        e2.set_cor_mat(cm)
        T, C, A = [], [], []
        for _ in range(60):
            a = int(rng.integers(0, 14))
            v = rng.choice(p, size=a + 2, replace=False)
            T.append(int(v[0])); C.append(int(v[1])); A.append([int(t) for t in v[2:]])
        for t, c, a, g in zip(T, C, A, e2.test_subsets_batch(T, C, A)):
            e = orc.test_subsets(t, c, a, max_k=max_k, alpha=alpha, n_obs_min=20, max_tests=max_tests)
            assert g["status"] == e["status"] and g["num_tests"] == e["num_tests"], (t, c, a, g, e)
            if e["status"] == 0:
                continue
            assert g["Zs"] == e["Zs"] and g["stat"] == e["stat"], (g, e)
            assert rel(g["pval"], e["pval"]) < 1e-12
        e2.close()
    eng.close()


@pytest.mark.parametrize("kind,max_k", [("fz", 6), ("fz", 7), ("fz_nz", 6)])
def test_fz_network(kind, max_k):
    if kind == "fz":
        data = _fz_data(60, 500, 41)
    else:
        counts = synth.generate(60, 500, 43, mode="S", habitats=2)
        data, _, _ = pre.normalize(counts, "fz_nz", prec=32)
        data = np.asfortranarray(data)
    n, p = data.shape
    for ff, R in ((False, 0), (True, 16)):
        eng = fw.Engine(kind, n, p, max_k=max_k, max_tests=20_000)  # (the counters of a context add up over its passes)
        eng.set_data(data)
        orc = O.Oracle("fz", cor_mat=eng.cor(), n_obs=n) if kind == "fz" else O.Oracle("fz_nz", data=data.astype(np.float64))
        got = eng.lgl(feed_forward=ff, round_size=R)
        exp = orc.learn(max_k=max_k, feed_forward=ff, round_size=max(R, 1) if ff else 1, max_tests=20_000)
        assert set(got["edges"]) == set(exp["edges"]) and len(exp["edges"]) > 0
        for e, w in exp["edges"].items():
            assert got["edges"][e] == w, (e, got["edges"][e], w)
        assert eng.counters()["cond_tests_ref"] == exp["n_cond_tests"]
        eng.close()


@pytest.mark.parametrize("kind,max_k", [("mi", 6), ("mi", 7), ("mi_nz", 6), ("mi_nz", 7), ("mi3", 6)])
def test_discrete(kind, max_k):
    # hps = 0 / 1 and a few thousand samples: with the default hps = 5 no test on 3^6 strata has power and the tables would never be read
    hps = 1
    if kind == "mi3":  # three-valued data under the plain "mi" rules: 3 x 3 sub-tables (the form that fits 3^6 strata, not 3^7)
        rng0 = np.random.default_rng(5)
        base = rng0.integers(0, 3, size=(3000, 6))
        data = np.ascontiguousarray((base[:, rng0.integers(0, 6, size=40)] + (rng0.random((3000, 40)) < 0.25) * rng0.integers(0, 3, size=(3000, 40))) % 3)
        kind, hps = "mi", 0
    else:
        counts = synth.generate(70, 3000, 47, mode="F", habitats=2 if kind == "mi_nz" else 1)
        data, _, _ = pre.normalize(counts, kind)
        data = np.ascontiguousarray(data)
        hps = 0 if kind == "mi_nz" else 1  # (nz-adjusted sub-tables hold a few hundred rows)
    n, p = data.shape
    eng = fw.Engine(kind, n, p, max_k=max_k, max_tests=5_000, n_obs_min=0, hps=hps)
    eng.set_data(data)
    orc = O.Oracle(kind, data, sparse=True, max_k=max_k)
    rng = np.random.default_rng(100 + max_k)
    X, Y, Zs = [], [], []
    for _ in range(400):
        k = int(rng.integers(0, max_k + 1))
        v = rng.choice(p, size=k + 2, replace=False)
        X.append(int(v[0])); Y.append(int(v[1])); Zs.append(tuple(int(t) for t in v[2:]))
    n_pow = 0
    for x, y, z, g in zip(X, Y, Zs, eng.test_batch(X, Y, Zs)):
        s, pv, df, pw = orc.test(x, y, z, hps=hps, n_obs_min=0)
        assert (g.df, g.suff_power) == (df, pw), (x, y, z)
        assert (g.stat == s) or rel(g.stat, s) < 1e-12 or abs(g.stat - s) < 1e-15
        assert (g.pval == pv) or rel(g.pval, pv) < 1e-10
        n_pow += int(pw and len(z) >= 6)
    assert n_pow > 10  # the deep tables are really evaluated
    T, C, A = [], [], []
    for _ in range(40):
        a = int(rng.integers(0, 12))
        v = rng.choice(p, size=a + 2, replace=False)
        T.append(int(v[0])); C.append(int(v[1])); A.append([int(t) for t in v[2:]])
    for t, c, a, g in zip(T, C, A, eng.test_subsets_batch(T, C, A)):
        e = orc.test_subsets(t, c, a, max_k=max_k, alpha=0.01, hps=hps, n_obs_min=0, max_tests=5_000)
        assert g["status"] == e["status"] and g["num_tests"] == e["num_tests"], (t, c, a, g, e)
        if e["status"] == 0:
            continue
        assert g["Zs"] == e["Zs"] and g["df"] == e["df"], (g, e)
        assert (g["stat"] == e["stat"]) or rel(g["stat"], e["stat"]) < 1e-12 or abs(g["stat"] - e["stat"]) < 1e-15
    before = eng.counters()["cond_tests_ref"]  # (the test_subsets batches above are counted too)
    got = eng.lgl(feed_forward=False, round_size=0)
    exp = orc.learn(max_k=max_k, feed_forward=False, max_tests=5_000, n_obs_min=0, hps=hps)
    assert set(got["edges"]) == set(exp["edges"])
    assert eng.counters()["cond_tests_ref"] - before == exp["n_cond_tests"]
    eng.close()


def test_limits():
    # recursive_pcor = 0 conditions job-local Gram matrices of up to 5 + 2 variables; 3-valued "mi" data with a 3 x 3 sub-table fits 3^6 strata
    with pytest.raises(fw.FlashWeaveError):
        fw.Engine("fz", 50, 10, max_k=6, recursive_pcor=False)
    with pytest.raises(fw.FlashWeaveError):
        fw.Engine("fz", 50, 10, max_k=8)
    data = np.ascontiguousarray(np.random.default_rng(1).integers(0, 3, size=(80, 12)))
    eng = fw.Engine("mi", 80, 12, max_k=7)
    with pytest.raises(fw.FlashWeaveError):
        eng.set_data(data)
    eng.close()
    eng = fw.Engine("mi", 80, 12, max_k=6)
    eng.set_data(data)
    eng.close()
