"""GPU parity tests of the FlashWeave-S (Fisher-z) HIP path against the CPU oracle, through the C ABI.

Tolerances (stated per the north star):
  * Pearson matrix (fp32 MFMA accumulation vs Float64 accumulation rounded to Float32): |diff| <= 5e-6.
  * partial correlations / edge weights, given the SAME Float32 matrix: bit-exact (only + - * / sqrt rint).
  * p-values: relative 1e-12 (device log/erfc vs libm).
"""
import numpy as np
import pytest

import flashweave_jl_amd as fw
from flashweave_jl_amd import preprocess as pre
from flashweave_jl_amd import synth
from oracle import oracle as O
from tests.util import GOLDEN, read_edgelist, rel

pytestmark = pytest.mark.gpu


def _synth_fz(p, n, seed):
    counts = synth.generate(p, n, seed, mode="S")
    data, _, _ = pre.normalize(counts, "fz", prec=32)
    return np.asfortranarray(data)


@pytest.fixture(scope="module")
def small():
    data = _synth_fz(400, 300, 11)
    n, p = data.shape
    eng = fw.Engine("fz", n, p, max_k=3)
    eng.set_data(data)
    cm = eng.cor()
    orc = O.Oracle("fz", cor_mat=cm, n_obs=n)
    return dict(data=data, n=n, p=p, eng=eng, cm=cm, orc=orc)


def test_cor_matrix_tolerance_and_structure(small):
    cm, data = small["cm"], small["data"]
    ref = O.cor(data.astype(np.float64), "f32")
    assert np.abs(cm - ref).max() <= 5e-6
    assert (cm == cm.T).all()                      # mirrored writes: exactly symmetric
    assert (np.diag(cm) == 1.0).all()
    assert np.abs(cm).max() <= 1.0


def test_cor_matrix_ragged_shapes():
    # n not a multiple of the k-tile (32), p not a multiple of the 128 tile nor of 4
    rng = np.random.default_rng(5)
    for n, p in ((37, 5), (100, 131), (333, 257)):
        data = np.asfortranarray(rng.standard_normal((n, p)).astype(np.float32))
        data[:, 0] = 1.5  # zero-variance column -> NaN row/col except the unit diagonal (Statistics.cor)
        eng = fw.Engine("fz", n, p)
        eng.set_data(data)
        cm = eng.cor()
        ref = O.cor(data.astype(np.float64), "f32")
        assert np.isnan(cm[0, 1:]).all() and np.isnan(cm[1:, 0]).all() and cm[0, 0] == 1.0
        assert np.abs(cm[1:, 1:] - ref[1:, 1:]).max() <= 5e-6
        eng.close()


def test_single_tests_bit_exact(small):
    eng, orc, p = small["eng"], small["orc"], small["p"]
    rng = np.random.default_rng(3)
    X, Y, Zs = [], [], []
    for _ in range(3000):
        k = int(rng.integers(0, 6))
        v = rng.choice(p, size=k + 2, replace=False)
        X.append(int(v[0])); Y.append(int(v[1])); Zs.append(tuple(int(t) for t in v[2:]))
    # duplicated conditioning variables (feed-forward whitelist) and Z == X
    X += [1, 2, 3]; Y += [5, 6, 7]; Zs += [(9, 9), (11, 12, 11), (3, 8)]
    got = eng.test_batch(X, Y, Zs)
    for x, y, z, g in zip(X, Y, Zs, got):
        s, pv, df, pw = orc.test(x, y, z, n_obs_min=20)
        assert (g.stat == s) or (np.isnan(g.stat) and np.isnan(s)), (x, y, z, g.stat, s)
        assert rel(g.pval, pv) < 1e-12 or (np.isnan(g.pval) and np.isnan(pv))
        assert g.df == 0 and g.suff_power == pw


def _check_subsets(eng, orc, T, C, A, max_k, alpha=0.01, max_tests=10_000_000):
    got = eng.test_subsets_batch(T, C, A)
    for t, c, a, g in zip(T, C, A, got):
        e = orc.test_subsets(t, c, a, max_k=max_k, alpha=alpha, n_obs_min=20, max_tests=max_tests)
        assert g["status"] == e["status"], (t, c, a, g, e)
        assert g["num_tests"] == e["num_tests"], (t, c, a, g, e)
        if e["status"] == 0:
            assert np.isnan(g["stat"]) and np.isnan(g["pval"]) and g["df"] == -1
            continue
        assert g["Zs"] == e["Zs"], (t, c, a, g, e)
        assert g["stat"] == e["stat"], (g, e)
        assert rel(g["pval"], e["pval"]) < 1e-12
        assert rel(g["frac"], e["frac"]) < 1e-12


def test_test_subsets_matches_reference_order(small):
    eng, orc, p = small["eng"], small["orc"], small["p"]
    rng = np.random.default_rng(4)
    T, C, A = [], [], []
    for _ in range(400):
        a = int(rng.integers(0, 30))
        v = rng.choice(p, size=a + 2, replace=False)
        T.append(int(v[0])); C.append(int(v[1])); A.append([int(t) for t in v[2:]])
    # jobs built from the strongest neighbours: long all-significant runs -> exercises the max-p rule
    cm = small["cm"]
    for t in range(20):
        order = np.argsort(-np.abs(cm[t]))
        nb = [int(v) for v in order if v != t][:14]
        T.append(t); C.append(nb[0]); A.append(nb[1:])
    A[3] = A[3] + A[3][:1]  # duplicate in the accepted list
    _check_subsets(eng, orc, T, C, A, max_k=3)


def test_test_subsets_max_tests_and_large_pool(small):
    data, n, p, cm = small["data"], small["n"], small["p"], small["cm"]
    eng = fw.Engine("fz", n, p, max_k=3, max_tests=37)
    eng.set_cor_mat(cm)
    orc = small["orc"]
    T, C, A = [], [], []
    for t in range(30):
        order = np.argsort(-np.abs(cm[t]))
        nb = [int(v) for v in order if v != t][:12]
        T.append(t); C.append(nb[0]); A.append(nb[1:])
    _check_subsets(eng, orc, T, C, A, max_k=3, max_tests=37)
    eng.close()
    # accepted pool larger than the LDS staging limit (2048): global-memory path; cap the work with max_tests
    rng = np.random.default_rng(8)
    eng = fw.Engine("fz", n, p, max_k=2, max_tests=3000, alpha=0.9999)
    eng.set_cor_mat(cm)
    big = [int(v) for v in rng.integers(2, p, size=2100)]
    _check_subsets(eng, orc, [0], [1], [big], max_k=2, alpha=0.9999, max_tests=3000)
    eng.close()


def test_test_subsets_pvalue_underflow_ties():
    # X and Y share a private component: every rho(X, Y | S) ~ 0.999 -> p underflows to exactly 0 for every subset,
    # so `pval >= lowest.pval` (tests.jl:338) makes the LAST enumerated subset win; near-zero but non-zero p-values
    # (subnormal range) are mixed in through a second, slightly noisier pair.
    rng = np.random.default_rng(12)
    n, p = 400, 40
    data = rng.standard_normal((n, p))
    c = rng.standard_normal(n)
    data[:, 0] = c + 0.02 * rng.standard_normal(n)
    data[:, 1] = c + 0.02 * rng.standard_normal(n)
    d = rng.standard_normal(n)
    data[:, 2] = d + 0.21 * rng.standard_normal(n)
    data[:, 3] = d + 0.21 * rng.standard_normal(n)
    data = np.asfortranarray(data.astype(np.float32))
    eng = fw.Engine("fz", n, p, max_k=3)
    eng.set_data(data)
    cm = eng.cor()
    orc = O.Oracle("fz", cor_mat=cm, n_obs=n)
    T = [0, 1, 2, 3, 0]
    C = [1, 0, 3, 2, 1]
    A = [list(range(4, 16)), list(range(10, 30)), list(range(4, 18)), list(range(20, 40)), [5, 6, 7, 5, 6]]
    got = eng.test_subsets_batch(T, C, A)
    pv = [g["pval"] for g in got]
    assert pv[0] == 0.0 and pv[1] == 0.0          # underflow regime reached
    _check_subsets(eng, orc, T, C, A, max_k=3)
    eng.close()


@pytest.mark.parametrize("max_k", [1, 2, 4, 5])
def test_test_subsets_other_max_k(small, max_k):
    n, p, cm, orc = small["n"], small["p"], small["cm"], small["orc"]
    eng = fw.Engine("fz", n, p, max_k=max_k)
    eng.set_cor_mat(cm)
    T, C, A = [], [], []
    for t in range(25):
        order = np.argsort(-np.abs(cm[t]))
        nb = [int(v) for v in order if v != t][:9]
        T.append(t); C.append(nb[0]); A.append(nb[1:])
    _check_subsets(eng, orc, T, C, A, max_k=max_k)
    eng.close()


def test_level0_neighbours(small):
    eng, orc, p = small["eng"], small["orc"], small["p"]
    got = eng.pw_univar_neighbors()
    exp = orc.level0(alpha=0.01, n_obs_min=20)
    assert (got["off"] == exp["off"]).all()
    assert (got["idx"] == exp["idx"]).all()
    assert (got["stat"] == exp["stat"]).all()
    assert np.allclose(got["pval"], exp["pval"], rtol=1e-12, atol=0)


@pytest.mark.parametrize("ff,R", [(False, 0), (True, 1), (True, 16)])
def test_network_matches_oracle(small, ff, R):
    n, p, cm, orc = small["n"], small["p"], small["cm"], small["orc"]
    eng = fw.Engine("fz", n, p, max_k=3)
    eng.set_cor_mat(cm)
    got = eng.lgl(feed_forward=ff, round_size=R)
    exp = orc.learn(max_k=3, feed_forward=ff, round_size=max(R, 1) if ff else 1)
    assert set(got["edges"]) == set(exp["edges"])
    for e, w in exp["edges"].items():
        assert got["edges"][e] == w            # weights are partial correlations: bit-exact
    cn = eng.counters()
    assert cn["cond_tests_ref"] == exp["n_cond_tests"]      # same sequential test count as the reference order
    assert cn["level0_tests"] == p * (p - 1) // 2
    assert cn["cond_tests_evaluated"] >= cn["cond_tests_ref"]
    eng.close()


@pytest.mark.parametrize("max_k,wtol", [(0, 1e-7), (3, 2e-7)])
def test_golden_networks_fz(max_k, wtol):
    # reference test/learning.jl:176-237 (exp_fz_maxk{0,3}.edgelist, generated with single_il + prec=64)
    raw = np.loadtxt(GOLDEN + "/HMP_SRA_gut_small.tsv", delimiter="\t", skiprows=1, usecols=range(1, 51))
    data, _, _ = pre.normalize(raw, "fz", prec=64)
    cm = O.cor(data, "f32")  # prec=64 path: cor in Float64, stored as Float32 (learning.jl:44)
    exp = read_edgelist("%s/learning_expected/exp_fz_maxk%d.edgelist" % (GOLDEN, max_k))
    eng = fw.Engine("fz", data.shape[0], data.shape[1], max_k=max_k)
    eng.set_cor_mat(cm)
    got = eng.lgl(feed_forward=True, round_size=1)["edges"]
    assert set(got) == set(exp)
    for e in exp:
        assert abs(got[e] - exp[e]) <= wtol
    # and with the matrix computed on the device from the Float32 data (prec=32 path): same edge set here
    eng2 = fw.Engine("fz", data.shape[0], data.shape[1], max_k=max_k)
    eng2.set_data(data.astype(np.float32))
    eng2.cor()
    got2 = eng2.lgl(feed_forward=True, round_size=1)["edges"]
    assert set(got2) == set(exp)
    for e in exp:
        assert abs(got2[e] - exp[e]) <= 5e-5  # 5-digit rounding inside pcor_rec amplifies fp32 differences
    eng.close(); eng2.close()


def test_full_size_properties():
    # BASELINE size class (p in the thousands): size-independent properties instead of an oracle run
    data = _synth_fz(3000, 600, 21)
    n, p = data.shape
    eng = fw.Engine("fz", n, p, max_k=3)
    eng.set_data(data)
    cm = eng.cor()
    assert (cm == cm.T).all() and (np.diag(cm) == 1).all() and np.nanmax(np.abs(cm)) <= 1.0
    r1 = eng.lgl(feed_forward=False)
    c1 = eng.counters()
    # idempotence: a second run on the same context gives the identical network and test count
    eng.reset_counters()
    r2 = eng.lgl(feed_forward=False)
    c2 = eng.counters()
    assert r1["edges"] == r2["edges"] and c1["cond_tests_ref"] == c2["cond_tests_ref"]
    # every edge is a level-0 neighbour pair (HITON-PC only prunes), weights are correlations in [-1, 1]
    nb = eng.pw_univar_neighbors()
    pairs = set()
    for v in range(p):
        for u in nb["idx"][nb["off"][v]:nb["off"][v + 1]]:
            pairs.add((min(v, int(u)), max(v, int(u))))
    assert set(r1["edges"]) <= pairs
    assert all(abs(w) <= 1.0 for w in r1["edges"].values())
    # spot-check 200 random directed results against the oracle given the device matrix
    orc = O.Oracle("fz", cor_mat=cm, n_obs=n)
    exp = orc.learn(max_k=3, feed_forward=False, max_targets=1500)
    # the first 1500 targets of the schedule have identical directed results
    off, idx, w = r1["pc_off"], r1["pc_idx"], r1["pc_weight"]
    eoff, eidx, ew = exp["pc_off"], exp["pc_idx"], exp["pc_weight"]
    checked = 0
    for T in range(p):
        if eoff[T + 1] > eoff[T]:
            assert list(idx[off[T]:off[T + 1]]) == list(eidx[eoff[T]:eoff[T + 1]])
            assert list(w[off[T]:off[T + 1]]) == list(ew[eoff[T]:eoff[T + 1]])
            checked += 1
    assert checked > 0
    eng.close()


def test_unscaled_division_equals_the_compilers_division():
    # fz_div_nn (fw_fz_core.h): the IEEE division sequence without v_div_scale / v_div_fmas for the operand ranges of the NaN-free
    # partial-correlation path (numerators: round5 values, denominators: products of roots of 1 - v^2).  v_rcp_f64 is a hardware
    # approximation, so the comparison runs on the device: 2^31 hashed operand pairs against the compiler's `n / d`, bit for bit.
    eng = fw.Engine("fz", 64, 8, max_k=3)
    for seed in (1, 2):
        assert eng.selftest(which=1, cases=1 << 30, seed=seed) == 0
    eng.close()


def test_learn_network_api_reproduces_all_golden_networks(tmp_path):
    # the reference's entry point on the bundled table (BASELINE config 1): all four modes x max_k in {0, 3}
    from flashweave_jl_amd import io as fio
    raw, header, _ = fio.read_table(GOLDEN + "/HMP_SRA_gut_small.tsv")
    for sensitive, het, name, wtol in ((True, False, "fz", 5e-5), (True, True, "fz_nz", 2e-5),
                                       (False, False, "mi", 1e-13), (False, True, "mi_nz", 1e-13)):
        for max_k in (0, 3):
            # the DEFAULT call: up to 512 variables learn_network runs the reference's deterministic single_il schedule, the one the
            # goldens were generated with (larger tables take default_round_size(p) targets per device round)
            net = fw.learn_network(raw, sensitive=sensitive, heterogeneous=het, max_k=max_k)
            assert net["parameters"]["round_size"] == 1 and net["parameters"]["schedule"].startswith("single_il")
            assert net["counters"]["normalized_on_device"]
            exp = read_edgelist("%s/learning_expected/exp_%s_maxk%d.edgelist" % (GOLDEN, name, max_k))
            assert set(net["edges"]) == set(exp), (name, max_k)
            assert all(abs(net["edges"][e] - exp[e]) <= wtol for e in exp)
    # dense_cor=False in its most natural call (recursive_pcor left at its default): runs, warns, and says what it ran
    with pytest.warns(UserWarning, match="recursive_pcor=False"):
        nm = fw.learn_network(raw, sensitive=True, heterogeneous=False, max_k=3, dense_cor=False)
    assert nm["parameters"]["recursive_pcor"] is False and nm["parameters"]["dense_cor"] is False and len(nm["edges"]) > 0
    # ... and the flag is ignored where the reference ignores it (no matrix exists for these tests anyway)
    ni = fw.learn_network(raw, sensitive=False, heterogeneous=True, max_k=3, dense_cor=False)
    assert ni["edges"] == net["edges"]
    out = tmp_path / "net.edgelist"
    net.save(str(out))
    back, hdr, _ = fio.read_edgelist(str(out))
    assert back == net["edges"] and len(hdr) == 50


@pytest.mark.parametrize("kind", ["fz", "fz_nz", "mi", "mi_nz"])
@pytest.mark.parametrize("fdr", [True, False])
def test_device_bh_equals_host_bh(kind, fdr, monkeypatch):
    # benjamini_hochberg! + condensed_stats_to_dict on the device (fw_bh.hip) against the host restatement
    # (FW_HOST_BH=1): offsets, partners, statistics and adjusted p-values must agree to the bit
    from flashweave_jl_amd import preprocess as pre, synth
    if kind in ("fz", "fz_nz"):
        counts = synth.generate(700, 300, 5, mode="S")
        data, _, _ = pre.normalize(counts, kind, prec=32)
    else:
        counts = synth.generate(700, 300, 5, mode="F")
        data, _, _ = pre.normalize(counts, kind)
    data = np.ascontiguousarray(data)
    n, p = data.shape
    res = []
    for host in ("1", "0"):
        monkeypatch.setenv("FW_HOST_BH", host)
        eng = fw.Engine(kind, n, p, max_k=3, FDR=fdr)
        eng.set_data(data)
        res.append(eng.pw_univar_neighbors())
        eng.close()
    a, b = res
    assert a["off"][-1] > 100
    for key in ("off", "idx", "stat", "pval"):
        assert np.array_equal(a[key], b[key]), key


@pytest.mark.parametrize("ff,R", [(False, 0), (True, 100), (True, 64)])
def test_device_rounds_equal_host_driver_and_oracle(small, ff, R, monkeypatch):
    # fw_devhiton.hip (rounds of >= 64 targets: state machines, merge and launch construction on the device) against
    # the host driver (FW_HOST_HITON=1) and the oracle: same directed results, weights to the bit, same test count.
    # R = 100 / 64 with feed_forward exercises the whitelist path (hiton.jl:20-30) on the device.
    n, p, cm, orc = small["n"], small["p"], small["cm"], small["orc"]
    res = {}
    for host in ("1", "0"):
        monkeypatch.setenv("FW_HOST_HITON", host)
        eng = fw.Engine("fz", n, p, max_k=3)
        eng.set_cor_mat(cm)
        net = eng.lgl(feed_forward=ff, round_size=R)
        res[host] = (net, eng.counters())
        eng.close()
    (nh, ch), (nd, cd) = res["1"], res["0"]
    assert nh["edges"] == nd["edges"]
    for key in ("pc_off", "pc_idx", "pc_weight", "pc_pval"):
        assert np.array_equal(nh[key], nd[key], equal_nan=True), key
    assert ch["cond_tests_ref"] == cd["cond_tests_ref"] and ch["subsets_calls"] == cd["subsets_calls"]
    assert cd["subsets_launches"] > 0
    exp = orc.learn(max_k=3, feed_forward=ff, round_size=max(R, 1) if ff else 1)
    assert set(nd["edges"]) == set(exp["edges"])
    for e, w in exp["edges"].items():
        assert nd["edges"][e] == w
    assert cd["cond_tests_ref"] == exp["n_cond_tests"]


def test_device_rounds_long_accepted_lists(monkeypatch):
    # one shared factor: every variable stays associated with every other one given any 3 others, so accepted lists grow
    # past FW_TAB_A = 512 and the device rounds have to switch on the in-lane kernel next to the table kernel
    # (max_tests caps the job sizes).  Device rounds == host driver, bit for bit.
    rng = np.random.default_rng(7)
    n, p = 2000, 600  # n large enough that every partial correlation (~0.2) clears the threshold (~0.06) with margin
    data = (rng.standard_normal((n, 1)) + 0.9 * rng.standard_normal((n, p))).astype(np.float32)
    res = {}
    for host in ("1", "0"):
        monkeypatch.setenv("FW_HOST_HITON", host)
        eng = fw.Engine("fz", n, p, max_k=3, max_tests=300)
        eng.set_data(data)
        eng.cor()
        net = eng.lgl(feed_forward=False, round_size=0)
        res[host] = (net, eng.counters())
        eng.close()
    (nh, ch), (nd, cd) = res["1"], res["0"]
    assert np.diff(nd["pc_off"]).max() > 512          # some PC sets (hence accepted lists) exceed the table kernel's limit
    assert nh["edges"] == nd["edges"]
    for key in ("pc_off", "pc_idx", "pc_weight", "pc_pval"):
        assert np.array_equal(nh[key], nd[key], equal_nan=True), key
    assert ch["cond_tests_ref"] == cd["cond_tests_ref"]


@pytest.mark.parametrize("max_k", [1, 2, 5])
def test_device_rounds_other_max_k(small, max_k, monkeypatch):
    # max_k = 5 runs the HIGHK variant of the segment kernel grid-strided under the device rounds; 1 and 2 the table
    # kernel without any size-3 subsets
    n, p, cm, orc = small["n"], small["p"], small["cm"], small["orc"]
    res = {}
    for host in ("1", "0"):
        monkeypatch.setenv("FW_HOST_HITON", host)
        eng = fw.Engine("fz", n, p, max_k=max_k)
        eng.set_cor_mat(cm)
        res[host] = (eng.lgl(feed_forward=False, round_size=0), eng.counters())
        eng.close()
    (nh, ch), (nd, cd) = res["1"], res["0"]
    assert nh["edges"] == nd["edges"] and ch["cond_tests_ref"] == cd["cond_tests_ref"]
    exp = orc.learn(max_k=max_k, feed_forward=False, round_size=1)
    assert set(nd["edges"]) == set(exp["edges"])
    for e, w in exp["edges"].items():
        assert nd["edges"][e] == w
    assert cd["cond_tests_ref"] == exp["n_cond_tests"]


def test_device_rounds_with_host_side_level0_lists(small, monkeypatch):
    # FW_HOST_BH=1 leaves no device copy of the level-0 neighbour lists: the device rounds upload them themselves
    n, p, cm = small["n"], small["p"], small["cm"]
    nets = []
    for host_bh in ("0", "1"):
        monkeypatch.setenv("FW_HOST_BH", host_bh)
        eng = fw.Engine("fz", n, p, max_k=3)
        eng.set_cor_mat(cm)
        nets.append(eng.lgl(feed_forward=False, round_size=0))
        eng.close()
    assert nets[0]["edges"] == nets[1]["edges"]


def test_single_il_first_two_targets_have_no_whitelist():
    """round_size = 1: the first two targets of the schedule both run with an empty whitelist (interleaved.jl:62,76-86)."""
    from tests.test_oracle_golden import single_il_first_two_matrix
    cm = single_il_first_two_matrix()
    eng = fw.Engine("fz", 200, 6, max_k=3)
    eng.set_cor_mat(cm)
    got = eng.lgl(feed_forward=True, round_size=1)
    exp = O.Oracle("fz", cor_mat=cm, n_obs=200).learn(max_k=3, feed_forward=True, round_size=1)
    assert got["edges"] == exp["edges"]
    assert (got["pc_off"] == exp["pc_off"]).all() and (got["pc_idx"] == exp["pc_idx"]).all()
    assert np.array_equal(np.isnan(got["pc_pval"]), np.isnan(exp["pc_pval"]))
    assert not np.isnan(got["pc_pval"][got["pc_off"][1]])  # target 1 tested variable 0
    eng.close()


def _unrank_subset(rank, a, max_k):
    """positions of the subset with this rank in the enumeration of tests.jl:281-346 (sizes max_k..1, lexicographic)"""
    from math import comb
    s = max_k
    while s > 1 and rank >= comb(a, s):
        rank -= comb(a, s)
        s -= 1
    pos, prev = [], -1
    for d in range(s):
        t = s - d
        c = prev + 1
        while rank >= comb(a - 1 - c, t - 1):
            rank -= comb(a - 1 - c, t - 1)
            c += 1
        pos.append(c)
        prev = c
    return pos


@pytest.mark.parametrize("max_k", [4, 5])
def test_size_4_5_table_kernels_value_at_random_ranks(max_k):
    """The level-2 table kernel (|accepted| <= 88) and the level-1 table path of the generic kernel (longer lists) against
    the explicit-test kernel (fz_test_batch_kernel: plain fz_pcor_dp on gathered entries).  alpha ~ 1 makes every test
    significant, so a job stops at rank max_tests - 1 (tests.jl:326-336) and reports THAT test: the statistic at
    a random rank deep inside the enumeration -- any z1-block, any (z1, z2) sub-block -- is compared bit for bit, together
    with the conditioning set (host-side unranking) and the test count."""
    from math import comb
    rng = np.random.default_rng(40 + max_k)
    n, p = 300, 260
    base = rng.standard_normal((n, 5))
    data = np.asfortranarray((base @ rng.standard_normal((5, p)) * 0.7 + rng.standard_normal((n, p))).astype(np.float32))
    ref = fw.Engine("fz", n, p, max_k=max_k, alpha=0.999999)
    ref.set_data(data)
    cm = ref.cor()
    lens = [6, 17, 40, 63, 88, 89, 100, 130, 200]
    checked = 0
    for rep in range(10):
        a_for_m = int(rng.choice(lens))
        total = sum(comb(a_for_m, s) for s in range(1, max_k + 1))
        # ranks spread over the whole enumeration of one of the list lengths (log-uniform: early and late blocks alike)
        M = int(np.exp(rng.uniform(0.0, np.log(min(total, 3_000_000))))) + 1
        eng = fw.Engine("fz", n, p, max_k=max_k, alpha=0.999999, max_tests=M)
        eng.set_cor_mat(cm)
        T, Cn, A = [], [], []
        for a in lens:
            for _ in range(3):
                v = [int(x) for x in rng.choice(p, size=a + 2, replace=False)]
                T.append(v[0]); Cn.append(v[1]); A.append(v[2:])
        got = eng.test_subsets_batch(T, Cn, A)
        X, Y, Z, G = [], [], [], []
        for t, c, acc, g in zip(T, Cn, A, got):
            tot = sum(comb(len(acc), s) for s in range(1, max_k + 1))
            if tot < M:
                continue  # the enumeration ends before max_tests: nothing stops this job (covered elsewhere)
            # (a partial correlation that rounds to exactly 0 has p = 1 and stops the job earlier: then THAT test is the one)
            assert g["num_tests"] <= M and (g["num_tests"] == M or g["pval"] >= 0.999999), (len(acc), M, g)
            pos = _unrank_subset(g["num_tests"] - 1, len(acc), max_k)
            zs = tuple(acc[q] for q in pos)
            assert tuple(g["Zs"]) == zs, (len(acc), M, g, zs)
            X.append(t); Y.append(c); Z.append(zs); G.append(g)
        exp = ref.test_batch(X, Y, Z)
        for g, e in zip(G, exp):
            assert g["stat"] == e.stat and g["pval"] == e.pval, (M, g, e)
            checked += 1
        eng.close()
    ref.close()
    assert checked > 60


def test_device_rounds_max_k5_table_kernels_equal_gather_form(monkeypatch):
    """Device rounds at max_k = 5 on data whose accepted lists run past 88 variables: the level-2 table kernel (short lists)
    and the level-1 table path (long lists) against the plain gather form of the same kernels (FW_NO_HK=1, FW_FZ_DBG=1:
    fz_pcor_dp on gathered matrix entries) -- network, directed lists, weights, p-values and reference-order test counts
    must agree bit for bit; max_tests bounds the enumerations (C(100, 5) = 7.5e7 subsets per job)."""
    rng = np.random.default_rng(77)
    n, p = 2000, 260
    # two blocks of noisy copies of one hidden factor each: no subset of five members separates two others (the factor is
    # not observed), so candidates keep being accepted and the lists grow to the block size (130)
    f = rng.standard_normal((n, 2))
    data = np.asfortranarray((f[:, np.arange(p) % 2] + 2.0 * rng.standard_normal((n, p))).astype(np.float32))
    res = {}
    for tag, env in (("tables", {}), ("gather", {"FW_NO_HK": "1", "FW_FZ_DBG": "1"})):
        for k in ("FW_NO_HK", "FW_FZ_DBG"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        eng = fw.Engine("fz", n, p, max_k=5, alpha=0.05, max_tests=60_000)
        eng.set_data(data)
        eng.cor()
        res[tag] = (eng.lgl(feed_forward=True, round_size=128), eng.counters())
        eng.close()
    (nt, ct), (ng, cg) = res["tables"], res["gather"]
    assert nt["edges"] == ng["edges"] and len(nt["edges"]) > 200
    for key in ("pc_off", "pc_idx", "pc_weight", "pc_pval"):
        assert np.array_equal(nt[key], ng[key], equal_nan=True), key
    assert ct["cond_tests_ref"] == cg["cond_tests_ref"] and ct["subsets_calls"] == cg["subsets_calls"]
    assert ct["cond_tests_ref"] > 50_000_000  # (and FW_TRACE_HOST=1 shows accepted lists beyond 88 entries)


def test_env_knobs_are_inert_without_fw_knobs():
    # The FW_* environment variables of DESIGN.md section 5 steer profiling and tests; the library reads them only under
    # FW_KNOBS=1 (csrc/fw_internal.h: fw_knob).  A stray FW_HOST_HITON=1 in a user's environment must not move the conditional
    # stage onto the host job pool.
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    snippet = ("import sys; sys.path.insert(0, %r)\n"
               "import numpy as np, flashweave_jl_amd as fw\n"
               "rng = np.random.default_rng(3)\n"
               "d = (rng.standard_normal((300, 1)) + rng.standard_normal((300, 200))).astype(np.float32)\n"
               "e = fw.Engine('fz', 300, 200, max_k=2); e.set_data(d); e.compute_cor(); e.lgl(feed_forward=False, round_size=0)\n"
               "print('HOST_ADVANCE', e.counters()['t_host_advance_s'] > 0)\n") % root
    out = {}
    for knobs in ("0", "1"):
        env = dict(os.environ, FW_HOST_HITON="1", FW_KNOBS=knobs)
        r = subprocess.run([sys.executable, "-c", snippet], env=env, cwd=root, check=True, capture_output=True, text=True).stdout
        out[knobs] = [ln for ln in r.splitlines() if ln.startswith("HOST_ADVANCE")][-1].split()[-1]
    assert out == {"0": "False", "1": "True"}, out
