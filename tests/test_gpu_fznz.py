"""GPU parity tests of the HE-S ("fz_nz", zero-ignoring Fisher-z) HIP path against the CPU oracle, through the C ABI.
The device sums run sequentially over the rows in Float64, like the oracle's: correlations, partial correlations and
edge weights are compared bit-exact; p-values within 1e-12 relative (device log/erfc)."""
import numpy as np
import pytest

import flashweave_jl_amd as fw
from flashweave_jl_amd import preprocess as pre
from flashweave_jl_amd import synth
from oracle import oracle as O
from tests.util import GOLDEN, read_edgelist, rel

pytestmark = pytest.mark.gpu


def _synth(p, n, seed):
    counts = synth.generate(p, n, seed, mode="S", habitats=4)
    data, _, _ = pre.normalize(counts, "fz_nz", prec=32)
    return np.asfortranarray(data)


@pytest.fixture(scope="module")
def ctx():
    data = _synth(250, 400, 31)
    n, p = data.shape
    eng = fw.Engine("fz_nz", n, p, max_k=3)
    eng.set_data(data)
    orc = O.Oracle("fz_nz", data=data.astype(np.float64))  # Float32 values, Float64 arithmetic (as on the device)
    return dict(data=data, n=n, p=p, eng=eng, orc=orc)


def _same(a, b):
    return a == b or (np.isnan(a) and np.isnan(b))


def test_single_tests(ctx):
    eng, orc, p = ctx["eng"], ctx["orc"], ctx["p"]
    rng = np.random.default_rng(2)
    X, Y, Zs = [], [], []
    for _ in range(1500):
        k = int(rng.integers(0, 4))
        v = rng.choice(p, size=k + 2, replace=False)
        X.append(int(v[0])); Y.append(int(v[1])); Zs.append(tuple(int(t) for t in v[2:]))
    got = eng.test_batch(X, Y, Zs)
    npow = 0
    for x, y, z, g in zip(X, Y, Zs, got):
        s, pv, df, pw = orc.test(x, y, z, n_obs_min=20)
        assert _same(g.stat, s) and g.suff_power == pw and g.df == 0, (x, y, z, g, (s, pv, pw))
        assert _same(g.pval, pv) or rel(g.pval, pv) < 1e-12
        npow += pw
    assert npow > 100


def test_level0(ctx):
    got = ctx["eng"].pw_univar_neighbors()
    exp = ctx["orc"].level0(alpha=0.01, n_obs_min=20)
    assert (got["off"] == exp["off"]).all() and (got["idx"] == exp["idx"]).all()
    assert (got["stat"] == exp["stat"]).all()
    assert np.allclose(got["pval"], exp["pval"], rtol=1e-12, atol=0) and len(exp["idx"]) > 0


def test_test_subsets(ctx):
    eng, orc, p = ctx["eng"], ctx["orc"], ctx["p"]
    nb = orc.level0(alpha=0.01, n_obs_min=20)
    rng = np.random.default_rng(5)
    T, C, A = [], [], []
    for _ in range(150):
        a = int(rng.integers(0, 14))
        v = rng.choice(p, size=a + 2, replace=False)
        T.append(int(v[0])); C.append(int(v[1])); A.append([int(t) for t in v[2:]])
    for t in range(p):
        nbr = [int(u) for u in nb["idx"][nb["off"][t]:nb["off"][t + 1]]]
        if len(nbr) >= 4:
            T.append(t); C.append(nbr[0]); A.append(nbr[1:12])
    got = eng.test_subsets_batch(T, C, A)
    kinds = set()
    for t, c, a, g in zip(T, C, A, got):
        e = orc.test_subsets(t, c, a, max_k=3, alpha=0.01, n_obs_min=20)
        assert g["status"] == e["status"] and g["num_tests"] == e["num_tests"], (t, c, a, g, e)
        if e["status"] == 0:
            continue
        assert g["Zs"] == e["Zs"] and _same(g["stat"], e["stat"]) and g["suff_power"] == e["suff_power"], (g, e)
        assert _same(g["pval"], e["pval"]) or rel(g["pval"], e["pval"]) < 1e-12
        kinds.add((e["status"], e["num_tests"] == 0))
    assert (1, True) in kinds and (1, False) in kinds  # "too few common rows" and ordinary rejections both occur


@pytest.mark.parametrize("ff,R", [(False, 0), (True, 1), (True, 16)])
def test_network_matches_oracle(ctx, ff, R):
    data, n, p, orc = ctx["data"], ctx["n"], ctx["p"], ctx["orc"]
    eng = fw.Engine("fz_nz", n, p, max_k=3)
    eng.set_data(data)
    got = eng.lgl(feed_forward=ff, round_size=R)
    exp = orc.learn(max_k=3, feed_forward=ff, round_size=max(R, 1) if ff else 1)
    assert set(got["edges"]) == set(exp["edges"]) and len(exp["edges"]) > 0
    for e, w in exp["edges"].items():
        assert got["edges"][e] == w
    assert eng.counters()["cond_tests_ref"] == exp["n_cond_tests"]
    eng.close()


@pytest.mark.parametrize("ff,R", [(False, 0), (True, 64), (True, 100)])
def test_device_rounds_equal_host_pool_and_oracle(ctx, ff, R, monkeypatch):
    """r05: fz_nz on the device-resident rounds (fw_devhiton.hip: per-target record slots written by dh_nz_recs_kernel, the sub-matrix
    kernel in front of the segment kernel of every round, matrices kept across the windows of a job) against the host job pool
    (FW_NZ_DEV=0, the only driver of r01-r04: tests.jl:293-308, statfuns.jl:138-155 per pool round) and the oracle: edges, weights,
    directed lists and p-values to the bit between the two drivers (same kernels, same sums), reference-order test count."""
    data, n, p, orc = ctx["data"], ctx["n"], ctx["p"], ctx["orc"]
    res = {}
    for dev in ("0", "1"):
        monkeypatch.setenv("FW_NZ_DEV", dev)
        eng = fw.Engine("fz_nz", n, p, max_k=3)
        eng.set_data(data)
        res[dev] = (eng.lgl(feed_forward=ff, round_size=R, edge_dict=False), eng.counters())
        eng.close()
    (n0, c0), (n1, c1) = res["0"], res["1"]
    for key in ("edge_src", "edge_dst", "pc_off", "pc_idx"):
        assert np.array_equal(n0[key], n1[key]), key
    for key in ("edge_weight", "pc_weight", "pc_pval"):
        assert np.array_equal(n0[key], n1[key], equal_nan=True), key
    assert c0["cond_tests_ref"] == c1["cond_tests_ref"] and c0["subsets_calls"] == c1["subsets_calls"]
    assert c1["kernel_launches"] != c0["kernel_launches"]          # the two drivers really are different paths
    exp = orc.learn(max_k=3, feed_forward=ff, round_size=max(R, 1) if ff else 1)
    ge = dict(zip(zip(n1["edge_src"].tolist(), n1["edge_dst"].tolist()), n1["edge_weight"].tolist()))
    assert ge == exp["edges"] and len(ge) > 0
    assert c1["cond_tests_ref"] == exp["n_cond_tests"]


@pytest.mark.parametrize("max_k", [0, 3])
def test_golden_networks_fz_nz(max_k):
    # reference test/learning.jl:176-237: exp_fz_nz_maxk{0,3}.edgelist (prec = 64).  The device takes the Float32 matrix
    # (the reference's default prec = 32): weights agree to Float32-input precision.
    raw = np.loadtxt(GOLDEN + "/HMP_SRA_gut_small.tsv", delimiter="\t", skiprows=1, usecols=range(1, 51))
    data, _, _ = pre.normalize(raw, "fz_nz", prec=32)
    exp = read_edgelist("%s/learning_expected/exp_fz_nz_maxk%d.edgelist" % (GOLDEN, max_k))
    eng = fw.Engine("fz_nz", data.shape[0], data.shape[1], max_k=max_k)
    eng.set_data(data)
    got = eng.lgl(feed_forward=True, round_size=1)["edges"]
    assert set(got) == set(exp)
    for e in exp:
        assert abs(got[e] - exp[e]) <= 2e-5
    eng.close()


# ---- fz_nz without a correlation matrix (recursive_pcor = False) ---------------------------------------------------------------
# The reference's FzTestCond with an empty cor_mat on the row views of hiton.jl:85 (tests.jl:253 -> statfuns.jl:19-21): every
# conditional test is StatsBase.partialcor of the view's columns.  Device: Float64 view correlations per job (fznz_submat_kernel) +
# the conditioning kernels of fw_fzs.hip with the job's own sample size.  Summation orders differ from the oracle's: 1e-10.
RTOL = 1e-10


@pytest.fixture(scope="module")
def ctx_s():
    data = _synth(250, 400, 37)
    n, p = data.shape
    eng = fw.Engine("fz_nz", n, p, max_k=3, recursive_pcor=False)
    eng.set_data(data)
    orc = O.Oracle("fz_nz", data=data.astype(np.float64))
    orc.set_fz_nz_stream(True)
    return dict(data=data, n=n, p=p, eng=eng, orc=orc)


def _near(a, b):
    return _same(a, b) or rel(a, b) < RTOL or abs(a - b) < 1e-14


def test_no_matrix_single_tests(ctx_s):
    eng, orc, p = ctx_s["eng"], ctx_s["orc"], ctx_s["p"]
    rng = np.random.default_rng(12)
    X, Y, Zs = [], [], []
    for _ in range(1200):
        k = int(rng.integers(0, 4))
        v = rng.choice(p, size=k + 2, replace=False)
        X.append(int(v[0])); Y.append(int(v[1])); Zs.append(tuple(int(t) for t in v[2:]))
    got = eng.test_batch(X, Y, Zs)
    npow = ndiff = 0
    rec = fw.Engine("fz_nz", ctx_s["n"], p, max_k=3)  # the recursive form on the same data: a different statistic for k >= 2
    rec.set_data(ctx_s["data"])
    got_rec = rec.test_batch(X, Y, Zs)
    rec.close()
    for x, y, z, g, gr in zip(X, Y, Zs, got, got_rec):
        s, pv, df, pw = orc.test(x, y, z, n_obs_min=20)
        assert g.suff_power == pw and g.df == 0 and _near(g.stat, s), (x, y, z, g, (s, pv, pw))
        assert _near(g.pval, pv) or abs(g.pval - pv) < 1e-12, (x, y, z, g, pv)
        npow += pw
        ndiff += pw and len(z) >= 2 and g.stat != gr.stat
    assert npow > 100 and ndiff > 20


def test_no_matrix_test_subsets(ctx_s):
    eng, orc, p = ctx_s["eng"], ctx_s["orc"], ctx_s["p"]
    nb = orc.level0(alpha=0.01, n_obs_min=20)
    rng = np.random.default_rng(15)
    T, C, A = [], [], []
    for _ in range(120):
        a = int(rng.integers(1, 14))
        v = rng.choice(p, size=a + 2, replace=False)
        T.append(int(v[0])); C.append(int(v[1])); A.append([int(t) for t in v[2:]])
    for t in range(p):
        nbr = [int(u) for u in nb["idx"][nb["off"][t]:nb["off"][t + 1]]]
        if len(nbr) >= 4:
            T.append(t); C.append(nbr[0]); A.append(nbr[1:12])
    got = eng.test_subsets_batch(T, C, A)
    kinds = set()
    for t, c, a, g in zip(T, C, A, got):
        e = orc.test_subsets(t, c, a, max_k=3, alpha=0.01, n_obs_min=20)
        assert g["status"] == e["status"] and g["num_tests"] == e["num_tests"], (t, c, a, g, e)
        if e["status"] == 0:
            continue
        assert g["Zs"] == e["Zs"] and _near(g["stat"], e["stat"]) and g["suff_power"] == e["suff_power"], (g, e)
        assert _near(g["pval"], e["pval"]) or abs(g["pval"] - e["pval"]) < 1e-12
        kinds.add((e["status"], e["num_tests"] == 0))
    assert (1, True) in kinds and (1, False) in kinds


@pytest.mark.parametrize("ff,R", [(False, 0), (True, 16)])
def test_no_matrix_network_matches_oracle(ctx_s, ff, R):
    data, n, p, orc = ctx_s["data"], ctx_s["n"], ctx_s["p"], ctx_s["orc"]
    eng = fw.Engine("fz_nz", n, p, max_k=3, recursive_pcor=False)
    eng.set_data(data)
    got = eng.lgl(feed_forward=ff, round_size=R)
    exp = orc.learn(max_k=3, feed_forward=ff, round_size=max(R, 1) if ff else 1)
    assert set(got["edges"]) == set(exp["edges"]) and len(exp["edges"]) > 0
    for e, w in exp["edges"].items():
        assert _near(got["edges"][e], w), (e, got["edges"][e], w)
    assert eng.counters()["cond_tests_ref"] == exp["n_cond_tests"]
    eng.close()


def test_no_matrix_constant_column_in_a_view():
    """A conditioning variable that is constant inside a job's row view (absent in every row where T and the candidate are both
    present): StatsBase.partialcor meets a zero sum of squares, 0 / 0 = NaN, `fz_pval(NaN)` = NaN, and `issig` is false for a NaN
    p-value (tests.jl:326-336) -- the enumeration STOPS at the first subset that holds the variable and the candidate is dropped.
    (The matrix form never sees this: `cor_subset!` writes 0 for such a pair, statfuns.jl:150-152.)  Device against oracle: status,
    num_tests, conditioning set; the statistic and the p-value are NaN on both sides; explicit tests likewise."""
    data = _synth(120, 400, 41).copy()
    n, p = data.shape
    nzc = (data != 0)
    co = nzc.T.astype(np.int32) @ nzc.astype(np.int32)
    np.fill_diagonal(co, 0)
    T, Cn = np.unravel_index(np.argmax(co), co.shape)
    T, Cn = int(T), int(Cn)
    both = nzc[:, T] & nzc[:, Cn]
    assert both.sum() >= 40
    others = [v for v in np.argsort(-(nzc & both[:, None]).sum(axis=0)) if v not in (T, Cn)]
    a, b, c, z = (int(v) for v in others[:4])
    data[both, z] = 0.0                      # z: present elsewhere, absent in every row of the (T, Cn) view
    assert (data[:, z] != 0).sum() > 5
    data = np.asfortranarray(data)
    eng = fw.Engine("fz_nz", n, p, max_k=3, recursive_pcor=False)
    eng.set_data(data)
    orc = O.Oracle("fz_nz", data=data.astype(np.float64))
    orc.set_fz_nz_stream(True)
    jobs = [[z], [a, z], [z, a, b], [a, b, c, z], [a, z, b, c]]
    got = eng.test_subsets_batch([T] * len(jobs), [Cn] * len(jobs), jobs)
    nan_stops = 0
    for acc, g in zip(jobs, got):
        e = orc.test_subsets(T, Cn, acc, max_k=3, alpha=0.01, n_obs_min=20)
        assert g["status"] == e["status"] and g["num_tests"] == e["num_tests"] and g["Zs"] == e["Zs"], (acc, g, e)
        assert _near(g["stat"], e["stat"]) and (_near(g["pval"], e["pval"]) or abs(g["pval"] - e["pval"]) < 1e-12), (acc, g, e)
        if np.isnan(e["stat"]):
            assert z in e["Zs"] and np.isnan(e["pval"]) and np.isnan(g["stat"]) and np.isnan(g["pval"])
            nan_stops += 1
    assert nan_stops >= 3
    s, pv, df, pw = orc.test(T, Cn, (a, z), n_obs_min=20)
    g = eng.test_batch([T], [Cn], [(a, z)])[0]
    assert pw and g.suff_power and np.isnan(s) and np.isnan(g.stat) and np.isnan(pv) and np.isnan(g.pval)
    orc.close()
    eng.close()
