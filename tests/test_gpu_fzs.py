"""GPU parity of FlashWeave-S WITHOUT a correlation matrix (fw_params.recursive_pcor = 0): conditional tests stream their
sample columns and compute StatsBase.partialcor's recursion from the data (reference statfuns.jl:19-21, tests.jl:253).
Checked against the oracle's restatement (pinned on test/statfuns.jl:24-37 in tests/test_oracle_golden.py).

Tolerance: both sides accumulate the centred cross products in Float64, in different orders (64 lane partials + a wave
reduction vs a sequential loop): partial correlations agree to 1e-11 absolute, p-values to 1e-8 relative."""
import numpy as np
import pytest

import flashweave_jl_amd as fw
from flashweave_jl_amd import preprocess as pre
from flashweave_jl_amd import synth
from oracle import oracle as O
from tests.util import GOLDEN

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    counts = synth.generate(300, 260, 31, mode="S")
    data, _, _ = pre.normalize(counts, "fz", prec=32)
    data = np.asfortranarray(data)
    n, p = data.shape
    eng = fw.Engine("fz", n, p, max_k=3, recursive_pcor=False)
    eng.set_data(data)
    cm = eng.cor()
    orc = O.Oracle("fz", cor_mat=cm, n_obs=n)
    orc.set_fz_data(data.astype(np.float64))
    return dict(data=data, n=n, p=p, eng=eng, orc=orc, cm=cm)


def test_reference_known_answers_on_gpu():
    # test/statfuns.jl:24-37: pcor(1, 16, (41,)) and pcor(31, 21, (7, 14, 18)) on the clr-transformed HMP table
    raw = np.loadtxt(GOLDEN + "/HMP_SRA_gut_small.tsv", delimiter="\t", skiprows=1, usecols=range(1, 51))
    clr, _, _ = pre.normalize(raw, "fz", prec=64)
    eng = fw.Engine("fz", clr.shape[0], clr.shape[1], max_k=3, recursive_pcor=False, n_obs_min=0)
    eng.set_data(clr.astype(np.float32))
    r1 = eng.test(0, 15, (40,))
    r3 = eng.test(30, 20, (6, 13, 17))
    assert abs(r1.stat - (-0.16393307352649356)) < 2e-6   # Float32 input data: the reference's own rtol is 1e-6 on Float64
    assert abs(r3.stat - (-0.07643814205965811)) < 2e-6
    eng.close()


def test_single_tests_stream(ctx):
    eng, orc, p = ctx["eng"], ctx["orc"], ctx["p"]
    rng = np.random.default_rng(5)
    X, Y, Zs = [], [], []
    for _ in range(3000):
        k = int(rng.integers(0, 4))
        v = rng.choice(p, size=k + 2, replace=False)
        X.append(int(v[0])); Y.append(int(v[1])); Zs.append(tuple(int(t) for t in v[2:]))
    got = eng.test_batch(X, Y, Zs)
    for x, y, z, g in zip(X, Y, Zs, got):
        s, pv, df, pw = orc.test(x, y, z, n_obs_min=20)
        assert g.suff_power == pw
        if len(z) == 0:
            assert g.stat == s   # univariate tests stay on the (shared) Float32 matrix: bit-exact
        else:
            assert abs(g.stat - s) < 1e-11, (x, y, z, g, s)
            assert abs(g.pval - pv) <= 1e-8 * max(pv, 1e-300) + 1e-300, (x, y, z, g, pv)


def test_stream_differs_from_recursive_only_by_rounding(ctx):
    # pcor (data) vs pcor_rec (Float32 matrix, 5-digit rounding at every level): same quantity, 1e-4 apart at most
    eng, p, n = ctx["eng"], ctx["p"], ctx["n"]
    rec = fw.Engine("fz", n, p, max_k=3)
    rec.set_cor_mat(ctx["cm"])
    rng = np.random.default_rng(6)
    X, Y, Zs = [], [], []
    for _ in range(500):
        v = rng.choice(p, size=5, replace=False)
        X.append(int(v[0])); Y.append(int(v[1])); Zs.append(tuple(int(t) for t in v[2:]))
    a, b = eng.test_batch(X, Y, Zs), rec.test_batch(X, Y, Zs)
    assert max(abs(u.stat - w.stat) for u, w in zip(a, b)) < 2e-4
    rec.close()


def test_test_subsets_and_network_stream(ctx):
    eng, orc, p, n = ctx["eng"], ctx["orc"], ctx["p"], ctx["n"]
    nb = orc.level0(alpha=0.01, n_obs_min=20)
    T, C, A = [], [], []
    for t in range(p):
        nbr = [int(u) for u in nb["idx"][nb["off"][t]:nb["off"][t + 1]]]
        if len(nbr) >= 3:
            T.append(t); C.append(nbr[0]); A.append(nbr[1:8])
    got = eng.test_subsets_batch(T, C, A)
    nstop = 0
    for t, c, a, g in zip(T, C, A, got):
        e = orc.test_subsets(t, c, a, max_k=3, alpha=0.01, n_obs_min=20)
        assert g["status"] == e["status"] and g["num_tests"] == e["num_tests"] and g["Zs"] == e["Zs"], (t, c, a, g, e)
        assert abs(g["stat"] - e["stat"]) < 1e-11
        nstop += e["status"] == 1
    assert nstop > 0 and len(T) > 20
    eng.reset_counters()
    net = eng.lgl(feed_forward=False, round_size=0)
    exp = orc.learn(max_k=3, feed_forward=False)
    assert set(net["edges"]) == set(exp["edges"])
    for e_, w in exp["edges"].items():
        assert abs(net["edges"][e_] - w) < 1e-11
    assert eng.counters()["cond_tests_ref"] == exp["n_cond_tests"]


def test_test_subsets_streams_without_a_correlation_matrix():
    # recursive_pcor = 0 needs the DATA only: fw_test_subsets_batch must not ask for a resident Pearson matrix (r02 gated both
    # entry points on it).  Also the generic form of the kernel: n = 346 is not a multiple of 4, so X and Y are streamed too
    # (no 16-byte rows, nothing held in registers).
    raw = np.loadtxt(GOLDEN + "/HMP_SRA_gut_small.tsv", delimiter="\t", skiprows=1, usecols=range(1, 51))
    clr, _, _ = pre.normalize(raw, "fz", prec=32)
    data = np.asfortranarray(clr)
    n, p = data.shape
    assert n % 4 != 0
    eng = fw.Engine("fz", n, p, max_k=3, recursive_pcor=False)
    eng.set_data(data)                      # no cor(), no set_cor_mat()
    orc = O.Oracle("fz", cor_mat=O.cor(data.astype(np.float64), "f32"), n_obs=n)
    orc.set_fz_data(data.astype(np.float64))
    rng = np.random.default_rng(11)
    T, C, A = [], [], []
    for _ in range(60):
        v = rng.choice(p, size=int(rng.integers(3, 12)), replace=False)
        T.append(int(v[0])); C.append(int(v[1])); A.append([int(t) for t in v[2:]])
    got = eng.test_subsets_batch(T, C, A)
    for t, c, a, g in zip(T, C, A, got):
        e = orc.test_subsets(t, c, a, max_k=3, alpha=0.01, n_obs_min=20)
        assert g["status"] == e["status"] and g["num_tests"] == e["num_tests"] and g["Zs"] == e["Zs"], (t, c, a, g, e)
        assert abs(g["stat"] - e["stat"]) < 1e-11
    eng.close()


@pytest.mark.parametrize("max_k", [3, 5])
def test_job_matrix_and_streamed_forms_agree(ctx, monkeypatch, max_k):
    # default: the correlations of a job come from its (a + 2) x (a + 2) Float64 matrix (fzs_gram_kernel), staged in LDS up to 80
    # variables and read through L2 beyond; FW_FZS_GRAM=0: every test streams its columns.  Same sums, same order per pair -> the two
    # forms agree far inside the oracle tolerance; both are compared with the oracle.  Lists of 3 .. 120 accepted variables.
    data, n, p, orc = ctx["data"], ctx["n"], ctx["p"], ctx["orc"]
    rng = np.random.default_rng(5 + max_k)
    T, C, A = [], [], []
    for la in [3, 5, 8, 13, 21, 40, 64, 78, 79, 95, 120] * 2:
        v = rng.choice(p, size=la + 2, replace=False)
        T.append(int(v[0])); C.append(int(v[1])); A.append([int(t) for t in v[2:]])
    res = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("FW_FZS_GRAM", mode)
        eng = fw.Engine("fz", n, p, max_k=max_k, recursive_pcor=False, max_tests=3000)
        eng.set_data(data)
        res[mode] = eng.test_subsets_batch(T, C, A)
        eng.close()
    for t, c, a, g, s in zip(T, C, A, res["1"], res["0"]):
        assert g["status"] == s["status"] and g["num_tests"] == s["num_tests"] and g["Zs"] == s["Zs"], (t, c, len(a), g, s)
        assert abs(g["stat"] - s["stat"]) < 1e-12 and abs(g["pval"] - s["pval"]) <= 1e-9 * max(abs(s["pval"]), 1e-300)
    for t, c, a, g in list(zip(T, C, A, res["1"]))[:12]:
        e = orc.test_subsets(t, c, a, max_k=max_k, alpha=0.01, n_obs_min=20, max_tests=3000)
        assert g["status"] == e["status"] and g["num_tests"] == e["num_tests"] and g["Zs"] == e["Zs"], (t, c, len(a), g, e)
        assert abs(g["stat"] - e["stat"]) < 1e-11


def test_job_matrix_cache_with_mixed_list_lengths_and_an_arena_regrow(monkeypatch):
    # A launch packs the job records in two passes (lists up to 512 variables first), i.e. not in pool order; when the matrix arena
    # has to grow -- always on the first launch of a context -- every record gets a fresh offset, and the JOB must remember the
    # offset of ITS record (r03 paired them by a running counter: a job that took a second window then conditioned a foreign
    # matrix).  Long and short lists interleaved, alpha near 1 so that every job runs through several windows; the cached-matrix
    # form must agree with the streamed form (no cache at all) and with the oracle.
    counts = synth.generate(900, 200, 77, mode="S")
    data, _, _ = pre.normalize(counts, "fz", prec=32)
    data = np.asfortranarray(data)
    n, p = data.shape
    assert p > 640
    rng = np.random.default_rng(9)
    T, C, A = [], [], []
    for la in [600, 12, 530, 30, 7, 620, 45, 513, 512]:
        v = rng.choice(p, size=la + 2, replace=False)
        T.append(int(v[0])); C.append(int(v[1])); A.append([int(t) for t in v[2:]])
    res = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("FW_FZS_GRAM", mode)
        eng = fw.Engine("fz", n, p, max_k=2, alpha=0.9999, recursive_pcor=False, max_tests=60_000, n_obs_min=0)
        eng.set_data(data)
        res[mode] = eng.test_subsets_batch(T, C, A)
        eng.close()
    orc = O.Oracle("fz", cor_mat=np.eye(p, dtype=np.float32), n_obs=n)
    orc.set_fz_data(data.astype(np.float64))
    multi = 0
    for t, c, a, g, s in zip(T, C, A, res["1"], res["0"]):
        assert g["status"] == s["status"] and g["num_tests"] == s["num_tests"] and g["Zs"] == s["Zs"], (t, c, len(a), g, s)
        assert abs(g["stat"] - s["stat"]) < 1e-12
        multi += g["num_tests"] > 5000
        e = orc.test_subsets(t, c, a, max_k=2, alpha=0.9999, n_obs_min=0, max_tests=60_000)
        assert g["status"] == e["status"] and g["num_tests"] == e["num_tests"] and g["Zs"] == e["Zs"], (t, c, len(a), g, e)
        assert abs(g["stat"] - e["stat"]) < 1e-11
    assert multi >= 4   # several windows per job: the cache was used
    orc.close()


@pytest.mark.parametrize("round_size", [32, 100])
def test_feed_forward_network_without_a_correlation_matrix(ctx, round_size):
    # the reference's default schedule (feed-forward rounds with whitelists) on the recursive_pcor = 0 path
    eng, orc = ctx["eng"], ctx["orc"]
    eng.reset_counters()
    net = eng.lgl(feed_forward=True, round_size=round_size)
    exp = orc.learn(max_k=3, feed_forward=True, round_size=round_size)
    assert set(net["edges"]) == set(exp["edges"]), (len(net["edges"]), len(exp["edges"]))
    for e_, w in exp["edges"].items():
        assert abs(net["edges"][e_] - w) < 1e-11
    assert eng.counters()["cond_tests_ref"] == exp["n_cond_tests"]


def test_network_without_any_correlation_matrix(ctx):
    # dense_cor = False (fw_params.no_cor_mat, learning.jl:42 / tests.jl:118-147): level 0 multiplies the centred columns on the matrix
    # cores and screens every tile in the epilogue -- the same Float32 correlations and thresholds as the matrix path, so the
    # network is the one of dense_cor = True, recursive_pcor = False to the bit; no p x p matrix is ever allocated.
    data, n, p, eng = ctx["data"], ctx["n"], ctx["p"], ctx["eng"]
    with pytest.raises(fw.FlashWeaveError):
        fw.Engine("fz", n, p, max_k=3, recursive_pcor=True, dense_cor=False)
    e2 = fw.Engine("fz", n, p, max_k=3, recursive_pcor=False, dense_cor=False)
    e2.set_data(data)
    with pytest.raises(fw.FlashWeaveError):
        e2.cor()
    for ff, rs in ((False, 0), (True, 64)):
        a = eng.lgl(feed_forward=ff, round_size=rs)
        b = e2.lgl(feed_forward=ff, round_size=rs)
        assert a["edges"] == b["edges"] and len(a["edges"]) > 100
    # explicit tests: conditional ones are the data path's; univariate ones come from the data too (Float64 sums of the two
    # columns instead of the Float32 matrix entry)
    cm = ctx["cm"]
    got = e2.test_batch([0, 3, 5], [1, 4, 9], [[], [], [2, 7]])
    ref = eng.test_batch([0, 3, 5], [1, 4, 9], [[], [], [2, 7]])
    assert abs(got[0].stat - cm[0, 1]) < 1e-6 and abs(got[1].stat - cm[3, 4]) < 1e-6
    assert got[2].stat == ref[2].stat and got[2].pval == ref[2].pval
    e2.close()


def test_learn_network_dense_cor_false():
    raw = np.loadtxt(GOLDEN + "/HMP_SRA_gut_small.tsv", delimiter="\t", skiprows=1, usecols=range(1, 51))
    a = fw.learn_network(raw, sensitive=True, heterogeneous=False, recursive_pcor=False)
    b = fw.learn_network(raw, sensitive=True, heterogeneous=False, recursive_pcor=False, dense_cor=False)
    assert a["edges"] == b["edges"] and len(a["edges"]) > 5


def test_no_matrix_level0_never_allocates_the_matrix():
    # p = 40 000: the fp32 matrix takes 5.96 GiB.  The same level 0 with dense_cor = False must use that much less device memory
    # (what remains is proportional to the data and to the number of significant pairs) and find the same neighbour lists.
    import torch
    rng = np.random.default_rng(3)
    n, p = 128, 40_000
    # (nearly independent columns, no FDR: about 1 % of the 8e8 pairs pass)
    base = rng.standard_normal((n, 4)).astype(np.float32)
    data = np.asfortranarray(0.2 * base @ rng.standard_normal((4, p)).astype(np.float32) + rng.standard_normal((n, p)).astype(np.float32))
    used, nbs = {}, {}
    for dense in (False, True):
        torch.cuda.synchronize()
        free0, _ = torch.cuda.mem_get_info()
        eng = fw.Engine("fz", n, p, max_k=0, recursive_pcor=False, dense_cor=dense, FDR=False)
        eng.set_data(data)
        nbs[dense] = eng.pw_univar_neighbors()
        torch.cuda.synchronize()
        free1, _ = torch.cuda.mem_get_info()
        used[dense] = free0 - free1
        eng.close()
    assert used[True] - used[False] > 0.9 * 4 * p * p, (used[True] / 2**30, used[False] / 2**30)
    a, b = nbs[True], nbs[False]
    assert (a["off"] == b["off"]).all() and (a["idx"] == b["idx"]).all() and (a["pval"] == b["pval"]).all() and a["off"][-1] > 10**6
    # a sample of the neighbour lists against Float64 correlations of the same columns
    d64 = data.astype(np.float64)
    dc = (d64 - d64.mean(axis=0)) / d64.std(axis=0)
    for t in (0, 17, 39_999):
        idx = b["idx"][b["off"][t]:b["off"][t + 1]]
        r = (dc[:, [t]] * dc[:, idx]).mean(axis=0)
        assert np.abs(r - b["stat"][b["off"][t]:b["off"][t + 1]]).max() < 1e-5
