"""Pins the CPU oracle (oracle/fw_oracle.c) against every golden vector / known-answer test the
reference holds for the hot path:
  test/data/tests_expected.tsv, test/contingency.jl:5-24,55-83, test/statfuns.jl:24-71,
  test/data/learning_expected/*.edgelist.
CPU only."""
import numpy as np
import pytest

from oracle import oracle as O
from tests.util import GOLDEN, load_norm, read_edgelist, read_tests_expected, rel

EXP = read_tests_expected()


@pytest.fixture(scope="module")
def mats():
    return dict(mi=load_norm("pres_abs", np.int64), mi_nz=load_norm("clr_nonzero_binned", np.int64),
                fz=load_norm("clr_adapt", np.float64))


# ---- test/tests.jl:41-74 ------------------------------------------------------------------
@pytest.mark.parametrize("kind", ["mi", "mi_nz"])
@pytest.mark.parametrize("sparse", [False, True])
def test_discrete_tests_expected(mats, kind, sparse):
    o = O.Oracle(kind, mats[kind], sparse=sparse, max_k=3)
    for Y in range(1, 50):  # FlashWeave.test(1, collect(2:50), data, test_name)
        s, p, df, pw = o.test(0, Y, (), hps=5, n_obs_min=0)
        es, ep, edf, epw = EXP["exp_uni_" + kind][Y - 1]
        assert (df, pw) == (edf, epw)
        assert rel(s, es) < 1e-12 and rel(p, ep) < 1e-12
    for key, Zs in (("condZ1", (6,)), ("condZ3", (6, 13, 17))):  # test(31, 21, (7,)/(7,14,18), ...)
        s, p, df, pw = o.test(30, 20, Zs, hps=5)
        es, ep, edf, epw = EXP["exp_%s_%s" % (key, kind)][0]
        assert (df, pw) == (edf, epw)
        assert rel(s, es) < 1e-12 and rel(p, ep) < 1e-12


def test_fz_tests_expected(mats):
    # inputs are the Float32-printed fixture clr_adapt.tsv -> tolerance is fixture precision; the
    # conditional rows only match with len_z = 0 and carry pcor_rec's 5-digit rounding.
    clr = mats["fz"]
    o = O.Oracle("fz", cor_mat=O.cor(clr, "f64"), n_obs=clr.shape[0])
    for Y in range(1, 50):
        s, p, df, pw = o.test(0, Y, (), n_obs_min=0)
        es, ep, edf, epw = EXP["exp_uni_fz"][Y - 1]
        assert (df, pw) == (edf, epw)
        assert rel(s, es) < 2e-5 and rel(p, ep) < 2e-5
    for key, Zs in (("condZ1", (6,)), ("condZ3", (6, 13, 17))):
        s, p, df, pw = o.test(30, 20, Zs)
        es, ep, edf, epw = EXP["exp_%s_fz" % key][0]
        assert (df, pw) == (edf, epw)
        assert abs(s - es) < 1e-4 and rel(p, ep) < 1e-3


# ---- test/contingency.jl ---------------------------------------------------------------------
VEC = np.array([[0, 0, 0, 0, 1, 1, 1, 1, 0, 1, 0, 1],
                [0, 0, 1, 1, 1, 1, 0, 0, 0, 1, 0, 1],
                [0, 0, 1, 1, 1, 1, 0, 0, 0, 1, 0, 2],
                [0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1]]).T


def _strata_multiset(t):
    return sorted(tuple(t[:2, :2, k].ravel()) for k in range(t.shape[2]) if t[:, :, k].sum() > 0)


@pytest.mark.parametrize("sparse", [False, True])
def test_contingency_known_answers(sparse):
    o = O.Oracle("mi", VEC, sparse=sparse, max_k=2)
    t12, _ = o.contingency_table(0, 1)
    assert (t12[:2, :2, 0] == np.array([[4, 2], [2, 4]])).all()
    t23, _ = o.contingency_table(1, 2)
    assert (t23[:2, :3, 0] == np.array([[6, 0, 0], [0, 5, 1]])).all()
    c12_3 = np.zeros((2, 2, 3), int)
    c12_3[0, 0, 0], c12_3[1, 0, 0], c12_3[0, 1, 1], c12_3[1, 1, 1], c12_3[1, 1, 2] = 4, 2, 2, 3, 1
    t, lz = o.contingency_table(0, 1, (2,), nslots=9)
    assert lz == 3 and _strata_multiset(t) == _strata_multiset(c12_3)
    c12_34 = np.zeros((2, 2, 6), int)
    c12_34[0, 0, 0] = 2
    c12_34[0, 1, 1], c12_34[1, 1, 1] = 2, 2
    c12_34[0, 0, 2], c12_34[1, 0, 2] = 2, 2
    c12_34[1, 1, 3] = 1
    c12_34[1, 1, 4] = 1
    t, lz = o.contingency_table(0, 1, (2, 3), nslots=9)
    assert lz == 5 and _strata_multiset(t) == _strata_multiset(c12_34)


def test_sparse_backend_all_zero_y_terminates():
    # test/contingency.jl:71-83: must not hang when Y is all-zero under Nz
    v = np.concatenate([np.ones(25, int), np.full(25, 2)])
    A = np.stack([v, np.zeros(50, int), v], axis=1)
    o = O.Oracle("mi_nz", A, sparse=True, max_k=1)
    o.test(0, 1, (2,), hps=5)


# ---- test/statfuns.jl ------------------------------------------------------------------------
def test_statfuns_known_answers(mats):
    assert rel(O.fz_pval(-0.16393307352649356, 351, 1), 0.0020593283914246987) < 1e-6
    assert rel(O.fz_pval(-0.07643814205965811, 351, 3), 0.1548665431407692) < 1e-6
    assert rel(abs(O.mutual_information([[4, 2], [2, 4]])), 0.05663301226513242) < 1e-12
    c12_3 = np.zeros((2, 2, 3), int)
    c12_3[0, 0, 0], c12_3[1, 0, 0], c12_3[0, 1, 1], c12_3[1, 1, 1], c12_3[1, 1, 2] = 4, 2, 2, 3, 1
    assert abs(O.mutual_information(c12_3)) < 1e-15
    assert rel(O.mi_pval(0.05663301226513242, 1, 351), 2.8770005665168745e-10) < 1e-6
    # pcor_rec vs the non-recursive values, atol 1e-4 (test/statfuns.jl:24-37), Float64 cor_mat
    clr = mats["fz"]
    o = O.Oracle("fz", cor_mat=O.cor(clr, "f64"), n_obs=clr.shape[0])
    assert abs(o.pcor_rec(0, 15, (40,)) - (-0.16393307352649356)) < 1e-4
    assert abs(o.pcor_rec(30, 20, (6, 13, 17)) - (-0.07643814205965811)) < 1e-4


def test_benjamini_hochberg_known_answer():
    pv = [0.0, 1.0, 0.973774, 0.722245, 0.805758, 0.713164, 0.314595, 0.947966, 0.001, 0.0339692]
    fdr = np.array([0.0, 1.0, 1.0, 1.0, 1.0, 1.0, 0.786488, 1.0, 0.005, 0.113231])
    got = O.benjamini_hochberg(pv)
    sig = got < 0.01
    assert (sig == (fdr < 0.01)).all()
    assert np.allclose(got[sig], fdr[sig], rtol=1e-6)


def test_chisq_ccdf_against_scipy():
    from scipy.stats import chi2
    rng = np.random.default_rng(1)
    for df in (1, 2, 3, 4, 5, 8, 13, 27, 54):
        for g in np.concatenate([rng.uniform(0, 5, 20), rng.uniform(5, 400, 20), [0.0, 1e-8, 1500.0]]):
            a, b = O.lib().fwo_chisq_ccdf(df, float(g)), chi2.sf(g, df)
            assert rel(a, b) < 5e-13, (df, g, a, b)


# ---- test/learning.jl:176-237 golden networks -------------------------------------------------
@pytest.mark.parametrize("kind,max_k,wtol", [("mi", 0, 1e-14), ("mi", 3, 1e-14), ("mi_nz", 0, 1e-14),
                                             ("mi_nz", 3, 1e-14), ("fz", 0, 1e-7), ("fz", 3, 2e-7)])
def test_golden_networks(mats, kind, max_k, wtol):
    exp = read_edgelist("%s/learning_expected/exp_%s_maxk%d.edgelist" % (GOLDEN, kind, max_k))
    if kind == "fz":
        o = O.Oracle("fz", cor_mat=O.cor(mats["fz"], "f32"), n_obs=mats["fz"].shape[0])
    else:
        o = O.Oracle(kind, mats[kind], sparse=True, max_k=max_k)
    r = o.learn(max_k=max_k, feed_forward=True, round_size=1)  # deterministic single_il schedule
    got = r["edges"]
    assert set(got) == set(exp)
    for e in exp:
        assert abs(got[e] - exp[e]) <= wtol
    assert r["n_level0_tests"] == 50 * 49 // 2


def test_mi_maxk3_single_mode_within_reference_tolerance(mats):
    # parallel="single" (no feed-forward) is granted approx_nbr_diff = 22 by test/learning.jl:210-212
    exp = read_edgelist("%s/learning_expected/exp_mi_maxk3.edgelist" % GOLDEN)
    o = O.Oracle("mi", mats["mi"], sparse=True, max_k=3)
    got = o.learn(max_k=3, feed_forward=False)["edges"]
    diff = set(got) ^ set(exp)
    assert 2 * len(diff) <= 22


def test_dense_equals_sparse_mi_nz_low_k(mats):
    # property borrowed from test/learning.jl:369-383 (dense == sparse for mi_nz at max_k 0/1 with some
    # binary variables); the oracle dense path omits the row views of hiton.jl:41-50, which the
    # nz-adjusted sub-table makes redundant for k <= 1
    A = mats["mi_nz"].copy()
    A[:, -6:] = (A[:, -6:] == 0).astype(A.dtype)
    for mk in (0, 1):
        od = O.Oracle("mi_nz", A, sparse=False, max_k=mk)
        os_ = O.Oracle("mi_nz", A, sparse=True, max_k=mk)
        a = od.learn(max_k=mk, feed_forward=True, round_size=1)["edges"]
        b = os_.learn(max_k=mk, feed_forward=True, round_size=1)["edges"]
        assert set(a) == set(b)
        for e in a:
            assert abs(a[e] - b[e]) < 1e-12


# ---- HE-S ("fz_nz") ---------------------------------------------------------------------
@pytest.fixture(scope="module")
def clr_nz64():
    from flashweave_jl_amd import preprocess as pre
    raw = np.loadtxt(GOLDEN + "/HMP_SRA_gut_small.tsv", delimiter="\t", skiprows=1, usecols=range(1, 51))
    data, _, _ = pre.normalize(raw, "fz_nz", prec=64)
    fx = load_norm("clr_nonzero", np.float64)  # printed as Float32
    assert data.shape == fx.shape and np.abs(data - fx).max() < 1e-6
    return data


def test_fz_nz_tests_expected(clr_nz64):
    o = O.Oracle("fz_nz", data=clr_nz64)
    for Y in range(1, 50):  # sub_data = rows with X != 0; the test removes the rows with Y == 0 (tests.jl:127-131)
        s, p, df, pw = o.test(0, Y, (), n_obs_min=0)
        es, ep, edf, epw = EXP["exp_uni_fz_nz"][Y - 1]
        assert (df, pw) == (edf, epw)
        assert rel(s, es) < 1e-12 and rel(p, ep) < 1e-12
    # conditional rows: the convenience wrapper uses a Float64 cor_mat of the row view (tests.jl:269-276)
    sub = clr_nz64[(clr_nz64[:, 30] != 0) & (clr_nz64[:, 20] != 0)]
    of = O.Oracle("fz", cor_mat=O.cor(sub, "f64"), n_obs=sub.shape[0])
    for key, Zs in (("condZ1", (6,)), ("condZ3", (6, 13, 17))):
        s, p, df, pw = of.test(30, 20, Zs)
        es, ep, edf, epw = EXP["exp_%s_fz_nz" % key][0]
        assert (df, pw) == (edf, epw) and abs(s - es) < 1e-4 and rel(p, ep) < 1e-3


def test_fz_nz_without_a_matrix(clr_nz64):
    """fz_nz with recursive_pcor = false (fwo_fz_nz_set_stream: pcor -> StatsBase.partialcor on the row view, tests.jl:253 with an empty
    cor_mat).  Anchors: (i) the reference's own conditional fz_nz rows of tests_expected.tsv -- computed there with pcor_rec, which
    differs from pcor only by its 5-digit rounding: 1e-4, as the reference's statfuns test states for the pair; (ii) the partial
    correlation from the inverse covariance of the view's columns (numpy), 1e-10; (iii) the view's own row count in the p-value."""
    o = O.Oracle("fz_nz", data=clr_nz64)
    o.set_fz_nz_stream(True)
    rows = (clr_nz64[:, 30] != 0) & (clr_nz64[:, 20] != 0)
    sub = clr_nz64[rows]
    for key, Zs in (("condZ1", (6,)), ("condZ3", (6, 13, 17))):
        s, p, df, pw = o.test(30, 20, Zs, n_obs_min=0)
        es, ep, edf, epw = EXP["exp_%s_fz_nz" % key][0]
        assert (df, pw) == (edf, epw) and abs(s - es) < 1e-4 and rel(p, ep) < 2e-3
        cols = [30, 20] + list(Zs)
        prec = np.linalg.inv(np.cov(sub[:, cols], rowvar=False))
        r_np = -prec[0, 1] / np.sqrt(prec[0, 0] * prec[1, 1])
        assert abs(s - r_np) < 1e-10
        assert rel(p, O.fz_pval(s, int(rows.sum()), 0)) < 1e-12
    # a whole job: same stopping behaviour as single tests in the reference's order (size 3 first, lexicographic)
    e = o.test_subsets(30, 20, [6, 13, 17, 2], max_k=3, alpha=0.01, n_obs_min=0)
    s3, p3, _, _ = o.test(30, 20, (6, 13, 17), n_obs_min=0)
    if not (p3 < 0.01):
        assert e["status"] == 1 and e["num_tests"] == 1 and abs(e["stat"] - s3) < 1e-15


@pytest.mark.parametrize("max_k", [0, 3])
def test_fz_nz_golden_networks(clr_nz64, max_k):
    exp = read_edgelist("%s/learning_expected/exp_fz_nz_maxk%d.edgelist" % (GOLDEN, max_k))
    o = O.Oracle("fz_nz", data=clr_nz64)
    got = o.learn(max_k=max_k, feed_forward=True, round_size=1)["edges"]
    assert set(got) == set(exp) and len(exp) in (9, 10)
    for e in exp:
        assert abs(got[e] - exp[e]) <= 1e-12


def test_fast_division_by_1e5_is_exact():
    # the device evaluates round(x, digits=5) with an FMA-based division by 1e5 (csrc/fw_fz.hip round5_f32/_f64);
    # exhaustive check of the identity over every integer the kernels can meet (|n| <= 400 000)
    import ctypes
    L = O.lib()
    L.fwo_check_fast_div1e5.restype = ctypes.c_int64
    L.fwo_check_fast_div1e5.argtypes = [ctypes.c_int64]
    assert L.fwo_check_fast_div1e5(400000) == 0


# ---- single_il master: the first TWO targets are enqueued with an empty whitelist (interleaved.jl:62,76-86) -----------
def single_il_first_two_matrix():
    """6 variables: {0, 1} a correlated pair (degree 1 each: the first two targets of the schedule, mutual neighbours),
    {2, 3, 4, 5} a clique (degree 3)."""
    cm = np.eye(6, dtype=np.float32)
    cm[0, 1] = cm[1, 0] = 0.6
    for a in range(2, 6):
        for b in range(a + 1, 6):
            cm[a, b] = cm[b, a] = 0.5
    return cm


def test_single_il_first_two_targets_have_no_whitelist():
    o = O.Oracle("fz", cor_mat=single_il_first_two_matrix(), n_obs=200)
    r = o.learn(max_k=3, feed_forward=True, round_size=1)
    assert (0, 1) in r["edges"]
    # target 1 is the second job of the schedule: it must TEST variable 0 (statistic and p present), not accept it from a
    # whitelist built from target 0's result (which would leave (NaN, NaN) in its directed result)
    off, idx, pv = r["pc_off"], r["pc_idx"], r["pc_pval"]
    for T in (0, 1):
        assert list(idx[off[T]:off[T + 1]]) == [1 - T]
        assert not np.isnan(pv[off[T]])
    # from the third target on the whitelist applies: target 3 accepts 2 untested
    sl = slice(off[3], off[4])
    assert np.isnan(pv[sl][list(idx[sl]).index(2)])


# ---- test/statfuns.jl:24-37: pcor (StatsBase.partialcor on the data, no cor_mat) -------------------------------------
def test_pcor_known_answers():
    from flashweave_jl_amd import preprocess as pre
    raw = np.loadtxt(GOLDEN + "/HMP_SRA_gut_small.tsv", delimiter="\t", skiprows=1, usecols=range(1, 51))
    clr, _, _ = pre.normalize(raw, "fz", prec=64)  # preprocess_data_default(data, "fz", prec=64), test/statfuns.jl:25
    o = O.Oracle("fz", cor_mat=O.cor(clr, "f64"), n_obs=clr.shape[0])
    o.set_fz_data(clr)
    assert rel(o.pcor(0, 15, (40,)), -0.16393307352649356) < 1e-6          # pcor(1, 16, (41,), data_clr)
    assert rel(o.pcor(30, 20, (6, 13, 17)), -0.07643814205965811) < 1e-6   # pcor(31, 21, (7, 14, 18), data_clr)
    # the same numbers through pcor_rec only to 1e-4 (5-digit rounding), as the reference's own test states
    assert abs(o.pcor_rec(0, 15, (40,)) - (-0.16393307352649356)) < 1e-4
    # conditional test through the stream path: p-value with len_z = 0 (tests.jl:256)
    s_, p_, df, pw = o.test(0, 15, (40,))
    assert rel(s_, -0.16393307352649356) < 1e-6 and pw
    assert rel(p_, O.fz_pval(s_, clr.shape[0], 0)) < 1e-12


# ---- test/learning.jl:369-383: dense == sparse networks for mi_nz at max_k 0 / 1 ("sparse special optim") -------------
def dense_sparse_property_matrix():
    """normalize_data(data, test_name="mi_nz", make_sparse=false) with the last six variables made binary, as the reference's
    test builds it (A[:, end-5:end] .= iszero.(A[:, end-5:end]))."""
    A = load_norm("clr_nonzero_binned", np.int64).copy()
    A[:, -6:] = (A[:, -6:] == 0).astype(np.int64)
    return A


@pytest.mark.parametrize("max_k", [0, 1])
def test_dense_equals_sparse_mi_nz(max_k):
    # the dense path tests on per-(target, candidate) row views (hiton.jl:41-50) with the Matrix table methods; the reference
    # asserts that its networks equal those of the SparseMatrixCSC path for max_k <= 1 -- this pins the oracle's views
    A = dense_sparse_property_matrix()
    nets = []
    for sparse in (False, True):
        o = O.Oracle("mi_nz", A, sparse=sparse, max_k=max_k)
        nets.append(o.learn(max_k=max_k, feed_forward=True, round_size=1))
    assert set(nets[0]["edges"]) == set(nets[1]["edges"]) and len(nets[0]["edges"]) >= 4
    for e, w in nets[0]["edges"].items():
        assert rel(w, nets[1]["edges"][e]) < 1e-9


def test_closed_form_fz_thresholds_sit_inside_the_guard_band():
    """csrc/fw_fz.hip fznz_thresholds: the |r| threshold of `p < alpha` for a job with n_R rows is taken in closed form,
    r* = tanh(erfc^-1(alpha) / (sqrt2 * zscale)), zscale = sqrt(n_R - 3) / 2, and the kernels decide by the exact p-value
    inside r* (1 +- 1e-9).  Here: the same formula against the oracle's p-value function (statfuns.jl:3-17) -- just outside
    the band the verdict of the threshold and of the p-value must agree, for small and large n_R and alpha from 1e-12 to
    0.9999 (the band is ~6 orders of magnitude wider than the rounding of tanh / erfc)."""
    import math
    from scipy.special import erfcinv

    for alpha in (1e-12, 1e-6, 0.001, 0.01, 0.05, 0.5, 0.9999):
        xcrit = float(erfcinv(alpha))
        for n_r in (4, 5, 10, 40, 200, 2000, 100000):
            zscale = math.sqrt(n_r - 3) / 2.0
            rs = math.tanh(xcrit * 0.7071067811865476 / zscale)
            for sgn in (1.0, -1.0):
                hi, lo = sgn * rs * (1.0 + 1e-9), sgn * rs * (1.0 - 1e-9)
                if abs(hi) < 1.0:
                    assert O.fz_pval(hi, n_r, 0) < alpha, (alpha, n_r, sgn)
                assert not (O.fz_pval(lo, n_r, 0) < alpha), (alpha, n_r, sgn)


def test_pcor_with_a_repeated_conditioning_variable():
    """Feed-forward whitelists can put a variable twice into the pool of an elimination job (hiton.jl:24-26 pushes a whitelisted
    member again).  StatsBase.partialcor then meets r(z, z) = S / sqrt(S S) = 1 exactly and divides 0 by 0: NaN, i.e. "not
    significant" (tests.jl:1-3) -- while pcor_rec on the matrix returns a finite value for the same call (its own guards).  The two
    variants of the reference therefore learn different feed-forward networks; the GPU paths follow each of them
    (tests/test_gpu_fzs.py::test_feed_forward_network_without_a_correlation_matrix, tests/test_gpu_fz.py)."""
    rng = np.random.default_rng(0)
    d = rng.standard_normal((200, 6))
    d[:, 2] += d[:, 0]
    d[:, 3] += d[:, 1] + d[:, 0]
    cm = O.cor(d, "f32")
    data_variant = O.Oracle("fz", cor_mat=cm, n_obs=200)
    data_variant.set_fz_data(d)
    matrix_variant = O.Oracle("fz", cor_mat=cm, n_obs=200)
    s1, p1, _, _ = data_variant.test(0, 1, (2,))
    s2, p2, _, _ = matrix_variant.test(0, 1, (2,))
    assert abs(s1 - s2) < 1e-5 and np.isfinite(p1)
    for zs in ((2, 2), (2, 3, 2), (3, 2, 2)):
        s, p, _, _ = data_variant.test(0, 1, zs)
        assert np.isnan(s) and np.isnan(p), (zs, s, p)
        s, p, _, _ = matrix_variant.test(0, 1, zs)
        assert np.isfinite(s) and np.isfinite(p), (zs, s, p)


def test_threaded_learn_equals_sequential_learn():
    """`fwo_learn_mt` (the oracle on a pool of threads, used by the full-size GPU tests): the targets between two whitelist
    snapshots are independent (interleaved.jl:124-183), so the threaded run must reproduce the sequential loop exactly --
    edges, weights, directed lists, p-values, test count; both kinds of test object, with and without feed-forward."""
    rng = np.random.default_rng(7)
    n, p = 200, 240
    base = rng.standard_normal((n, 10))
    data = (base @ rng.standard_normal((10, p)) + 1.2 * rng.standard_normal((n, p))).astype(np.float32)
    cm = O.cor(data.astype(np.float64), "f32")
    cases = [(O.Oracle("fz", cor_mat=cm, n_obs=n), 3)]
    lat = rng.standard_normal((n, 6))
    disc = ((lat @ rng.standard_normal((6, 90)) + rng.standard_normal((n, 90))) > 0.3).astype(np.int32)
    disc *= 1 + (rng.random((n, 90)) < 0.4)
    cases.append((O.Oracle("mi_nz", disc), 3))
    cases.append((O.Oracle("mi", (disc > 0).astype(np.int32)), 2))
    for orc, mk in cases:
        for ff, rs in ((True, 32), (True, 5), (False, 1)):
            a = orc.learn(max_k=mk, feed_forward=ff, round_size=rs)
            b = orc.learn(max_k=mk, feed_forward=ff, round_size=rs, threads=4)
            assert a["edges"] == b["edges"] and a["n_cond_tests"] == b["n_cond_tests"]
            for k in ("pc_off", "pc_idx", "pc_weight", "pc_pval"):
                assert np.array_equal(a[k], b[k], equal_nan=True), k
        orc.close()
    assert len(a["edges"]) >= 0
