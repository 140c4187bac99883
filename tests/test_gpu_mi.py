"""GPU parity tests of the discrete (FlashWeave-F "mi" / FlashWeaveHE-F "mi_nz") HIP path against the CPU oracle,
through the C ABI.  Integer outputs (df, suff_power, test counts, edge sets, conditioning sets) are compared
bit-exact; MI statistics within 1e-12 relative (fp64 summation order / device log), p-values within 1e-10
relative (device lgamma / exp in the incomplete gamma function)."""
import numpy as np
import pytest

import flashweave_jl_amd as fw
from flashweave_jl_amd import preprocess as pre
from flashweave_jl_amd import synth
from oracle import oracle as O
from tests.util import GOLDEN, load_norm, read_edgelist, read_tests_expected, rel

pytestmark = pytest.mark.gpu
STOL, PTOL = 1e-12, 1e-10


def _close(a, b, tol):
    return (a == b) or (np.isnan(a) and np.isnan(b)) or rel(a, b) < tol


def _synth(kind, p, n, seed):
    if kind == "mi":
        counts = synth.generate(p, n, seed, mode="F")
        data, _, _ = pre.normalize(counts, "mi")
    else:
        counts, meta = synth.generate(p, n, seed, mode="F", habitats=4, n_meta=20)
        data, rm, _ = pre.normalize(counts, "mi_nz")
        meta = meta[rm]
        keep = [j for j in range(meta.shape[1]) if len(np.unique(meta[:, j])) == 2]
        data = np.concatenate([data, meta[:, keep]], axis=1)  # binary meta variables (max_val 1 -> not zero-adjusted)
    return np.ascontiguousarray(data)


@pytest.fixture(scope="module", params=["mi", "mi_nz"])
def ctx(request):
    kind = request.param
    data = _synth(kind, 300, 400, 23)
    n, p = data.shape
    eng = fw.Engine(kind, n, p, max_k=3)
    eng.set_data(data)
    orc = O.Oracle(kind, data, sparse=True, max_k=3)
    return dict(kind=kind, data=data, n=n, p=p, eng=eng, orc=orc)


def test_levels(ctx):
    lv, mv = ctx["eng"].levels()
    elv, emv = ctx["orc"].levels()
    assert (lv == elv).all() and (mv == emv).all()
    assert ctx["eng"].n_obs_min == ctx["orc"].auto_n_obs_min(-1, 5, 3)


def test_tests_expected_tsv():
    # reference test/tests.jl:41-74 (integer inputs are the preprocessing fixtures)
    exp = read_tests_expected()
    for kind, fx in (("mi", "pres_abs"), ("mi_nz", "clr_nonzero_binned")):
        data = load_norm(fx, np.int64)
        eng = fw.Engine(kind, data.shape[0], data.shape[1], max_k=3, n_obs_min=0)
        eng.set_data(data)
        got = eng.test_batch([0] * 49, list(range(1, 50)), [()] * 49)
        for g, e in zip(got, exp["exp_uni_" + kind]):
            assert (g.df, g.suff_power) == (e[2], e[3])
            assert _close(g.stat, e[0], 1e-11) and _close(g.pval, e[1], 1e-9)
        for key, Zs in (("condZ1", (6,)), ("condZ3", (6, 13, 17))):
            g = eng.test(30, 20, Zs)
            e = exp["exp_%s_%s" % (key, kind)][0]
            assert (g.df, g.suff_power) == (e[2], e[3])
            assert _close(g.stat, e[0], 1e-11) and _close(g.pval, e[1], 1e-9)
        eng.close()


def test_single_tests(ctx):
    eng, orc, p = ctx["eng"], ctx["orc"], ctx["p"]
    rng = np.random.default_rng(3)
    X, Y, Zs = [], [], []
    for _ in range(4000):
        k = int(rng.integers(0, 4))
        v = rng.choice(p, size=k + 2, replace=False)
        X.append(int(v[0])); Y.append(int(v[1])); Zs.append(tuple(int(t) for t in v[2:]))
    X += [1, 2, 3]; Y += [5, 6, 7]; Zs += [(9, 9), (11, 12, 11), (3, 8)]  # duplicates, Z == X
    got = eng.test_batch(X, Y, Zs)
    nom = eng.n_obs_min
    npow = 0
    for x, y, z, g in zip(X, Y, Zs, got):
        s, pv, df, pw = orc.test(x, y, z, hps=5, n_obs_min=nom)
        assert (g.df, g.suff_power) == (df, pw), (x, y, z, g, (s, pv, df, pw))
        assert _close(g.stat, s, STOL) and _close(g.pval, pv, PTOL), (x, y, z, g, (s, pv, df, pw))
        npow += pw
    assert npow > 100  # the comparison is not vacuous


def _check_subsets(eng, orc, T, C, A, max_k, nom, alpha=0.01, max_tests=10_000_000):
    got = eng.test_subsets_batch(T, C, A)
    nstop = nall = 0
    for t, c, a, g in zip(T, C, A, got):
        e = orc.test_subsets(t, c, a, max_k=max_k, alpha=alpha, hps=5, n_obs_min=nom, max_tests=max_tests)
        assert g["status"] == e["status"] and g["num_tests"] == e["num_tests"], (t, c, a, g, e)
        if e["status"] == 0:
            continue
        assert g["Zs"] == e["Zs"] and g["df"] == e["df"] and g["suff_power"] == e["suff_power"], (t, c, a, g, e)
        assert _close(g["stat"], e["stat"], STOL) and _close(g["pval"], e["pval"], PTOL), (g, e)
        nstop += e["status"] == 1
        nall += e["status"] == 2
    return nstop, nall


def test_test_subsets(ctx):
    eng, orc, p = ctx["eng"], ctx["orc"], ctx["p"]
    nb = orc.level0(alpha=0.01, hps=5, n_obs_min=eng.n_obs_min)
    rng = np.random.default_rng(4)
    T, C, A = [], [], []
    for _ in range(150):  # random pools
        a = int(rng.integers(0, 12))
        v = rng.choice(p, size=a + 2, replace=False)
        T.append(int(v[0])); C.append(int(v[1])); A.append([int(t) for t in v[2:]])
    for t in range(p):  # pools of true neighbours: long significant runs
        nbr = [int(u) for u in nb["idx"][nb["off"][t]:nb["off"][t + 1]]]
        if len(nbr) >= 4:
            T.append(t); C.append(nbr[0]); A.append(nbr[1:9])
    nstop, nall = _check_subsets(eng, orc, T, C, A, 3, eng.n_obs_min)
    assert nstop > 0


def test_test_subsets_max_tests(ctx):
    kind, data, n, p, orc = ctx["kind"], ctx["data"], ctx["n"], ctx["p"], ctx["orc"]
    eng = fw.Engine(kind, n, p, max_k=3, max_tests=5, alpha=0.5)
    eng.set_data(data)
    T, C, A = list(range(20)), list(range(20, 40)), [list(range(40, 48))] * 20
    _check_subsets(eng, orc, T, C, A, 3, eng.n_obs_min, alpha=0.5, max_tests=5)
    eng.close()


def test_level0(ctx):
    eng, orc = ctx["eng"], ctx["orc"]
    got = eng.pw_univar_neighbors()
    exp = orc.level0(alpha=0.01, hps=5, n_obs_min=eng.n_obs_min)
    assert (got["off"] == exp["off"]).all() and (got["idx"] == exp["idx"]).all()
    assert np.allclose(got["stat"], exp["stat"], rtol=STOL, atol=0)
    assert np.allclose(got["pval"], exp["pval"], rtol=PTOL, atol=0)
    assert len(exp["idx"]) > 0


@pytest.mark.parametrize("kind", ["mi_nz", "mi"])
def test_level0_matrix_core_form_equals_oracle(kind, monkeypatch):
    """mi_level0_mfma_kernel (r04: the pair counts as an int8 matrix-core Gram product, 128 x 128 tiles; default for mi_nz from 1 024
    variables on) forced on a small table: several tiles incl. diagonal ones and a ragged last tile, binary meta variables (pairs the
    branch-free first pass hands to the full screen), and for the plain three-valued kind EVERY pair through the survivor-list overflow
    path.  Lists, statistics and p-values against the oracle, and the same bytes as the popcount form."""
    data = _synth(kind, 300, 400, 29)
    n, p = data.shape
    res = {}
    for knob in ("2", "0"):
        monkeypatch.setenv("FW_L0_MFMA", knob)
        eng = fw.Engine(kind, n, p, max_k=3)
        eng.set_data(data)
        res[knob] = eng.pw_univar_neighbors()
        res[knob + "m"] = eng.counters()["level0_tests"]
        nom = eng.n_obs_min
        eng.close()
    exp = O.Oracle(kind, data, sparse=True, max_k=3).level0(alpha=0.01, hps=5, n_obs_min=nom)
    got = res["2"]
    assert (got["off"] == exp["off"]).all() and (got["idx"] == exp["idx"]).all() and len(exp["idx"]) > 100
    assert np.allclose(got["stat"], exp["stat"], rtol=STOL, atol=0) and np.allclose(got["pval"], exp["pval"], rtol=PTOL, atol=0)
    for f in ("off", "idx", "stat", "pval"):
        assert np.array_equal(got[f], res["0"][f]), f
    assert res["2m"] == res["0m"] == p * (p - 1) // 2


def test_level0_matrix_core_form_many_samples(monkeypatch):
    """Counts above 2^15 (the epilogue packs two 16-bit counts per word, the MX-fp4 accumulators hold them as Float32) and a sample
    count that is no multiple of the 64-sample words or the 8-word stages: 40 037 samples, dense dependent three-valued columns."""
    rng = np.random.default_rng(77)
    n, p = 40037, 150
    base = rng.integers(0, 3, size=(n, 6))
    data = np.empty((n, p), dtype=np.int32)
    for j in range(p):
        flip = rng.random(n) < 0.55
        data[:, j] = np.where(flip, rng.choice(3, size=n, p=[0.25, 0.4, 0.35]), base[:, j % 6])
    res = {}
    for knob in ("2", "0"):
        monkeypatch.setenv("FW_L0_MFMA", knob)
        eng = fw.Engine("mi_nz", n, p, max_k=3)
        eng.set_data(data)
        res[knob] = eng.pw_univar_neighbors()
        nom = eng.n_obs_min
        eng.close()
    for f in ("off", "idx", "stat", "pval"):
        assert np.array_equal(res["2"][f], res["0"][f]), f
    exp = O.Oracle("mi_nz", data, sparse=True, max_k=3).level0(alpha=0.01, hps=5, n_obs_min=nom)
    assert (res["2"]["off"] == exp["off"]).all() and (res["2"]["idx"] == exp["idx"]).all() and len(exp["idx"]) > 1000
    assert np.allclose(res["2"]["stat"], exp["stat"], rtol=STOL, atol=0) and np.allclose(res["2"]["pval"], exp["pval"], rtol=PTOL, atol=0)


@pytest.mark.parametrize("ff,R", [(False, 0), (True, 1), (True, 16)])
def test_network_matches_oracle(ctx, ff, R):
    kind, data, n, p, orc = ctx["kind"], ctx["data"], ctx["n"], ctx["p"], ctx["orc"]
    eng = fw.Engine(kind, n, p, max_k=3)
    eng.set_data(data)
    got = eng.lgl(feed_forward=ff, round_size=R)
    exp = orc.learn(max_k=3, feed_forward=ff, round_size=max(R, 1) if ff else 1)
    assert set(got["edges"]) == set(exp["edges"])       # edge sets: bit-exact
    for e, w in exp["edges"].items():
        assert _close(got["edges"][e], w, STOL)
    cn = eng.counters()
    assert cn["cond_tests_ref"] == exp["n_cond_tests"]
    assert cn["level0_tests"] == p * (p - 1) // 2
    eng.close()


@pytest.mark.parametrize("kind,fx,max_k", [("mi", "pres_abs", 0), ("mi", "pres_abs", 3),
                                           ("mi_nz", "clr_nonzero_binned", 0), ("mi_nz", "clr_nonzero_binned", 3)])
def test_golden_networks(kind, fx, max_k):
    # reference test/learning.jl:176-237: exp_{mi,mi_nz}_maxk{0,3}.edgelist (single_il schedule, sparse data)
    data = load_norm(fx, np.int64)
    exp = read_edgelist("%s/learning_expected/exp_%s_maxk%d.edgelist" % (GOLDEN, kind, max_k))
    eng = fw.Engine(kind, data.shape[0], data.shape[1], max_k=max_k)
    eng.set_data(data)
    got = eng.lgl(feed_forward=True, round_size=1)["edges"]
    assert set(got) == set(exp)
    for e in exp:
        assert abs(got[e] - exp[e]) <= 1e-13
    eng.close()


def test_csc_input_equals_dense_input(ctx):
    kind, data, n, p = ctx["kind"], ctx["data"], ctx["n"], ctx["p"]
    colptr, rowval, nzval = O.dense_to_csc(data)
    e1 = fw.Engine(kind, n, p, max_k=3)
    e1.set_data((colptr, rowval, nzval))
    a = e1.pw_univar_neighbors()
    b = ctx["eng"].pw_univar_neighbors()
    assert (a["off"] == b["off"]).all() and (a["idx"] == b["idx"]).all() and (a["pval"] == b["pval"]).all()
    e1.close()


def test_rejects_unsupported_values():
    bad = np.zeros((50, 4), dtype=np.int32)
    bad[:, 1] = 62  # 0..61 are served (values above 2: the generic form; r01-r04: 0..7); 62 and negative values are not
    eng = fw.Engine("mi_nz", 50, 4)
    with pytest.raises(fw.FlashWeaveError) as ei:
        eng.set_data(bad)
    assert ei.value.code == -5
    eng.close()


def test_full_size_properties():
    # cfg2 size class (1k OTUs x 500 samples, FlashWeave-F): properties + a spot check against the oracle
    data = _synth("mi", 1000, 500, 20260930)
    n, p = data.shape
    eng = fw.Engine("mi", n, p, max_k=3)
    eng.set_data(data)
    r1 = eng.lgl(feed_forward=False)
    c1 = eng.counters()
    eng.reset_counters()
    r2 = eng.lgl(feed_forward=False)
    assert r1["edges"] == r2["edges"] and c1["cond_tests_ref"] == eng.counters()["cond_tests_ref"]
    nb = eng.pw_univar_neighbors()
    pairs = {(min(v, int(u)), max(v, int(u))) for v in range(p) for u in nb["idx"][nb["off"][v]:nb["off"][v + 1]]}
    assert set(r1["edges"]) <= pairs
    orc = O.Oracle("mi", data, sparse=True, max_k=3)
    exp = orc.learn(max_k=3, feed_forward=False)
    assert set(r1["edges"]) == set(exp["edges"])
    assert c1["cond_tests_ref"] == exp["n_cond_tests"]
    eng.close()


def test_dense_matrix_rules(ctx):
    # contingency.jl:7-56 + level_map! (misc.jl:162-184): the Matrix methods the reference uses with make_sparse=false.
    # Oracle(sparse=False) follows them and is pinned by tests_expected.tsv; levels_z differs from the sparse rules for
    # mi_nz, so power verdicts can differ between the two engines below.
    kind, data, n, p = ctx["kind"], ctx["data"], ctx["n"], ctx["p"]
    eng = fw.Engine(kind, n, p, max_k=3, dense_rules=True)
    eng.set_data(data)
    orc = O.Oracle(kind, data, sparse=False, max_k=3)
    rng = np.random.default_rng(11)
    X, Y, Zs = [], [], []
    for _ in range(4000):
        k = int(rng.integers(0, 4))
        v = rng.choice(p, size=k + 2, replace=False)
        X.append(int(v[0])); Y.append(int(v[1])); Zs.append(tuple(int(t) for t in v[2:]))
    got = eng.test_batch(X, Y, Zs)
    sparse_got = ctx["eng"].test_batch(X, Y, Zs)
    nom = eng.n_obs_min
    npow = ndiff = 0
    for x, y, z, g, gs in zip(X, Y, Zs, got, sparse_got):
        s, pv, df, pw = orc.test(x, y, z, hps=5, n_obs_min=nom)
        assert (g.df, g.suff_power) == (df, pw), (x, y, z, g, (s, pv, df, pw))
        assert _close(g.stat, s, STOL) and _close(g.pval, pv, PTOL), (x, y, z, g, (s, pv, df, pw))
        npow += pw
        ndiff += (g.suff_power != gs.suff_power)
    assert npow > 100
    if kind == "mi":
        assert ndiff == 0  # without zero adjustment the two rule sets coincide
    # test_subsets on the full dense matrix (tests.jl:281-346 called without a row view)
    nb = orc.level0(alpha=0.01, hps=5, n_obs_min=nom)
    T, C, A = [], [], []
    for t in range(p):
        nbrs = [int(v) for v in nb["idx"][nb["off"][t]:nb["off"][t + 1]]]
        if len(nbrs) >= 3:
            for c in nbrs[:2]:
                acc = [v for v in nbrs if v != c][:6]
                T.append(t); C.append(c); A.append(acc)
        if len(T) >= 300:
            break
    assert len(T) > 20
    _check_subsets(eng, orc, T, C, A, 3, nom)
    eng.close()  # (networks under the dense rules: test_dense_mi_nz_network_with_row_views)


def test_tests_expected_tsv_dense_rules():
    # the reference's expected TestResults were produced on dense matrices (test/tests.jl:41-74)
    exp = read_tests_expected()
    for kind, fx in (("mi", "pres_abs"), ("mi_nz", "clr_nonzero_binned")):
        data = load_norm(fx, np.int64)
        eng = fw.Engine(kind, data.shape[0], data.shape[1], max_k=3, n_obs_min=0, dense_rules=True)
        eng.set_data(data)
        for key, Zs in (("condZ1", (6,)), ("condZ3", (6, 13, 17))):
            g = eng.test(30, 20, Zs)
            e = exp["exp_%s_%s" % (key, kind)][0]
            assert (g.df, g.suff_power) == (e[2], e[3])
            assert _close(g.stat, e[0], 1e-11) and _close(g.pval, e[1], 1e-9)
        eng.close()


@pytest.mark.parametrize("ff,R", [(False, 0), (True, 100)])
def test_device_rounds_equal_host_driver_and_oracle(ctx, ff, R, monkeypatch):
    # fw_devhiton.hip for the discrete kinds: same directed results / weights / test counts as the host driver and
    # the oracle (rounds of >= 64 targets run on the device; R = 100 with feed_forward exercises the whitelists)
    kind, data, n, p, orc = ctx["kind"], ctx["data"], ctx["n"], ctx["p"], ctx["orc"]
    res = {}
    monkeypatch.setenv("FW_DEV_MIN_TARGETS", "64")  # the default threshold for discrete kinds is 4096 targets
    for host in ("1", "0"):
        monkeypatch.setenv("FW_HOST_HITON", host)
        eng = fw.Engine(kind, n, p, max_k=3)
        eng.set_data(data)
        net = eng.lgl(feed_forward=ff, round_size=R)
        res[host] = (net, eng.counters())
        eng.close()
    (nh, ch), (nd, cd) = res["1"], res["0"]
    # same edges, directed lists and counts; statistics to 1e-12: the host pool runs one test per wavefront (Float64 sums over 64
    # lanes), the persistent kernel four per wavefront (mi_test_core4: sums over the 16 lanes of a row) -- a different summation
    # order of the same terms (discrete MI rel <= 1e-12)
    assert set(nh["edges"]) == set(nd["edges"])
    for e, w in nh["edges"].items():
        assert _close(nd["edges"][e], w, 1e-12), e
    for key in ("pc_off", "pc_idx"):
        assert np.array_equal(nh[key], nd[key]), key
    assert np.allclose(nh["pc_weight"], nd["pc_weight"], rtol=1e-12, atol=1e-15, equal_nan=True)
    assert np.allclose(nh["pc_pval"], nd["pc_pval"], rtol=1e-10, atol=0.0, equal_nan=True)
    assert ch["cond_tests_ref"] == cd["cond_tests_ref"] and ch["subsets_calls"] == cd["subsets_calls"]
    exp = orc.learn(max_k=3, feed_forward=ff, round_size=max(R, 1) if ff else 1)
    assert set(nd["edges"]) == set(exp["edges"])
    assert cd["cond_tests_ref"] == exp["n_cond_tests"]


@pytest.mark.parametrize("R", [64, 100, 257])
def test_whole_schedule_on_the_device_equals_the_per_round_loop_and_oracle(ctx, R, monkeypatch):
    """r05: the discrete kinds' whole feed-forward schedule on the device (fwi_devhiton_mi_schedule: per-target state built from the
    level-0 CSR, whitelists appended between the launches by dh_wl_append_kernel, one download) against the per-round loop with the
    host in between (FW_MI_SCHED=0: whitelists from the host's running graph, interleaved.jl:124-183) and the oracle: directed lists,
    weights, p-values, reference-order test count; the whitelists must really be in play (NaN weights = joined without a test)."""
    kind, data, n, p, orc = ctx["kind"], ctx["data"], ctx["n"], ctx["p"], ctx["orc"]
    monkeypatch.setenv("FW_DEV_MIN_TARGETS", "64")
    res = {}
    for sched in ("0", "1"):
        monkeypatch.setenv("FW_MI_SCHED", sched)
        eng = fw.Engine(kind, n, p, max_k=3)
        eng.set_data(data)
        res[sched] = (eng.lgl(feed_forward=True, round_size=R, edge_dict=False), eng.counters())
        eng.close()
    (n0, c0), (n1, c1) = res["0"], res["1"]
    for key in ("edge_src", "edge_dst", "pc_off", "pc_idx"):
        assert np.array_equal(n0[key], n1[key]), key
    # (a last round of fewer than FW_DEV_MIN_TARGETS targets runs on the host pool in the per-round loop: one test per wavefront
    # there, four per wavefront in the persistent kernel -- another summation order of the same terms)
    for key in ("edge_weight", "pc_weight"):
        assert np.allclose(n0[key], n1[key], rtol=1e-12, atol=1e-15, equal_nan=True), key
    assert np.allclose(n0["pc_pval"], n1["pc_pval"], rtol=1e-10, atol=0.0, equal_nan=True)
    assert c0["cond_tests_ref"] == c1["cond_tests_ref"] and c0["subsets_calls"] == c1["subsets_calls"]
    exp = orc.learn(max_k=3, feed_forward=True, round_size=R)
    assert np.array_equal(n1["pc_off"], exp["pc_off"]) and np.array_equal(n1["pc_idx"], exp["pc_idx"])
    assert np.allclose(n1["pc_weight"], exp["pc_weight"], rtol=1e-11, atol=1e-15, equal_nan=True)
    assert c1["cond_tests_ref"] == exp["n_cond_tests"]
    if R < p // 2 and kind == "mi":
        assert int(np.isnan(exp["pc_weight"]).sum()) > 0


@pytest.mark.parametrize("seq,win0,cmin", [(1, 2, 1), (2, 8, 2), (4, 64, 8)])
def test_persistent_kernel_boards_equal_oracle(ctx, seq, win0, cmin, monkeypatch):
    """dh_mi_target_kernel with the board machinery forced on for nearly every job (the owner runs `seq` tests alone, then
    publishes windows of win0, 8 win0, ... ranks in records of >= cmin ranks that any wavefront may claim): directed
    results, weights and the reference-order test count must not depend on how an enumeration was cut up."""
    kind, data, n, p, orc = ctx["kind"], ctx["data"], ctx["n"], ctx["p"], ctx["orc"]
    monkeypatch.setenv("FW_DEV_MIN_TARGETS", "1")
    monkeypatch.setenv("FW_MI_SEQ", str(seq))
    monkeypatch.setenv("FW_MI_WIN0", str(win0))
    monkeypatch.setenv("FW_MI_CHUNK_MIN", str(cmin))
    eng = fw.Engine(kind, n, p, max_k=3)
    eng.set_data(data)
    net = eng.lgl(feed_forward=False, round_size=0)
    cn = eng.counters()
    eng.close()
    exp = orc.learn(max_k=3, feed_forward=False)
    assert set(net["edges"]) == set(exp["edges"])
    assert (net["pc_off"] == exp["pc_off"]).all() and (net["pc_idx"] == exp["pc_idx"]).all()
    assert np.allclose(net["pc_weight"], exp["pc_weight"], rtol=1e-11, atol=1e-15, equal_nan=True)
    assert cn["cond_tests_ref"] == exp["n_cond_tests"]
    assert cn["cond_tests_evaluated"] >= cn["cond_tests_ref"]


@pytest.mark.parametrize("max_k", [4, 5])
def test_discrete_max_k_4_5(ctx, max_k):
    """Conditioning sets of 4 and 5 variables (L^k = 81 / 243 strata): the reference sizes its tables for any max_k
    (types.jl:98-117); single tests, test_subsets and a small network against the oracle."""
    kind, data, n, p = ctx["kind"], ctx["data"], ctx["n"], ctx["p"]
    orc = O.Oracle(kind, data, sparse=True, max_k=max_k)
    eng = fw.Engine(kind, n, p, max_k=max_k, n_obs_min=0, hps=1)  # hps = 1: enough power left at 81+ strata to see statistics
    eng.set_data(data)
    rng = np.random.default_rng(40 + max_k)
    X, Y, Zs = [], [], []
    for _ in range(1500):
        k = int(rng.integers(3, max_k + 1))
        v = rng.choice(p, size=k + 2, replace=False)
        X.append(int(v[0])); Y.append(int(v[1])); Zs.append(tuple(int(t) for t in v[2:]))
    got = eng.test_batch(X, Y, Zs)
    npow = 0
    for x, y, z, g in zip(X, Y, Zs, got):
        s_, pv, df, pw = orc.test(x, y, z, hps=1, n_obs_min=0)
        assert (g.df, g.suff_power) == (df, pw), (x, y, z, g, (s_, pv, df, pw))
        assert _close(g.stat, s_, STOL) and _close(g.pval, pv, PTOL), (x, y, z, g, (s_, pv, df, pw))
        npow += pw
    assert npow > 50
    T, C, A = [], [], []
    for _ in range(60):
        a = int(rng.integers(1, 9))
        v = rng.choice(p, size=a + 2, replace=False)
        T.append(int(v[0])); C.append(int(v[1])); A.append([int(t) for t in v[2:]])
    got = eng.test_subsets_batch(T, C, A)
    for t, c, a, g in zip(T, C, A, got):
        e = orc.test_subsets(t, c, a, max_k=max_k, alpha=0.01, hps=1, n_obs_min=0)
        assert g["status"] == e["status"] and g["num_tests"] == e["num_tests"], (t, c, a, g, e)
        assert g["Zs"] == e["Zs"] and g["df"] == e["df"]
    eng.close()
    eng = fw.Engine(kind, n, p, max_k=max_k)
    eng.set_data(data)
    net = eng.lgl(feed_forward=False, round_size=0)
    exp = orc.learn(max_k=max_k, feed_forward=False)
    assert set(net["edges"]) == set(exp["edges"])
    assert eng.counters()["cond_tests_ref"] == exp["n_cond_tests"]
    eng.close()


@pytest.mark.parametrize("max_k", [0, 1, 3])
def test_dense_mi_nz_network_with_row_views(max_k):
    """mi_nz networks under the dense-matrix rules: HITON-PC tests every (T, candidate) pair on the rows where T / the candidate
    are non-zero (if they have more than two levels): prepare_nzdata, hiton.jl:41-50,85,193.  Device (one extra AND plane per
    view in the popcount kernels) against the oracle's dense path, whose views are pinned by the reference's dense == sparse
    property at max_k <= 1 (tests/test_oracle_golden.py::test_dense_equals_sparse_mi_nz); that property is re-checked here
    on the device."""
    from tests.test_oracle_golden import dense_sparse_property_matrix
    for A in (dense_sparse_property_matrix(), _synth("mi_nz", 250, 300, 29)):
        n, p = A.shape
        eng = fw.Engine("mi_nz", n, p, max_k=max_k, dense_rules=True)
        eng.set_data(A)
        got = eng.lgl(feed_forward=True, round_size=1)
        cn = eng.counters()
        eng.close()
        exp = O.Oracle("mi_nz", A, sparse=False, max_k=max_k).learn(max_k=max_k, feed_forward=True, round_size=1)
        assert set(got["edges"]) == set(exp["edges"])
        for e, w in exp["edges"].items():
            assert _close(got["edges"][e], w, STOL)
        assert cn["cond_tests_ref"] == exp["n_cond_tests"]
        if max_k <= 1:  # test/learning.jl:369-383
            sp = fw.Engine("mi_nz", n, p, max_k=max_k)
            sp.set_data(A)
            gs = sp.lgl(feed_forward=True, round_size=1)
            sp.close()
            assert set(gs["edges"]) == set(got["edges"])
            for e, w in gs["edges"].items():
                assert _close(got["edges"][e], w, 1e-9)


def test_row_views_flag_for_test_subsets():
    # fw_set_row_views: the ABI's test_subsets on the view hiton.jl would pass.  Checked on one-candidate HITON runs: the
    # first job of a target whose accepted list is [c0] is test_subsets(T, c1, [c0]) on the (T, c1) view -- so a network run
    # (views always on) and explicit batches with the flag on must agree on significance for those jobs.
    A = _synth("mi_nz", 120, 300, 31)
    n, p = A.shape
    eng = fw.Engine("mi_nz", n, p, max_k=1, dense_rules=True)
    eng.set_data(A)
    lv, _ = eng.levels()
    rng = np.random.default_rng(3)
    T = [int(v) for v in rng.integers(0, p, 200)]
    C = [int((t + 1 + rng.integers(0, p - 1)) % p) for t in T]
    acc = [[int((t + c) % p)] if (t + c) % p not in (t, c) else [int((t + c + 1) % p)] for t, c in zip(T, C)]
    off = eng.test_subsets_batch(T, C, acc)
    eng.set_row_views(True)
    on = eng.test_subsets_batch(T, C, acc)
    eng.set_row_views(False)
    ndiff = 0
    for t, c, a, r0, r1 in zip(T, C, acc, off, on):
        if lv[t] <= 2 and lv[c] <= 2:
            assert r0 == r1  # no view for two-level variables: the flag changes nothing
        else:
            ndiff += (r0["suff_power"], r0["df"]) != (r1["suff_power"], r1["df"])
    assert ndiff > 0  # rows where T / the candidate are zero really left the tables
    eng.close()


def _big_n_data(kind, n, p, seed):
    """Discretised data with more rows than a 16-bit count holds: 'mi' through the usual front-end; 'mi_nz' built directly
    (0 = absent, 1 / 2 = the two bins of the non-zeros: the front-end drops too many rows of the synthetic habitats)."""
    if kind == "mi":
        return np.ascontiguousarray(_synth(kind, 24, n, seed)[:, :p])
    rng = np.random.default_rng(seed)
    base = rng.standard_normal((n, 3))
    x = base @ rng.standard_normal((3, p)) + 1.5 * rng.standard_normal((n, p))
    habitat = rng.random((n, 1)) < 0.5
    present = rng.random((n, p)) < np.where(habitat, 0.8, 0.45)
    return np.ascontiguousarray(np.where(present, 1 + (x > 0), 0).astype(np.int64))


@pytest.mark.parametrize("kind,rows", [("mi", 70_000), ("mi_nz", 70_000)])
def test_more_than_65535_samples(kind, rows):
    # cell counts beyond 16 bits: the segment / batch kernels switch to 32-bit counts and tables (fw_mi_core.h, WIDE), level 0 counts in
    # 32 bits anyway; the HITON-PC runs through the level-synchronous rounds.  Single tests of every size and a max_k = 2 network of ten
    # variables against the oracle (with this many samples every pair is associated, and the oracle pays every row of every test).
    data = _big_n_data(kind, rows, 10, 41)
    n, p = data.shape
    assert n > 65_535
    eng = fw.Engine(kind, n, p, max_k=3)
    eng.set_data(data)
    orc = O.Oracle(kind, data, sparse=True, max_k=3)
    rng = np.random.default_rng(9)
    X, Y, Zs = [], [], []
    for _ in range(300):
        k = int(rng.integers(0, 4))
        v = rng.choice(p, size=k + 2, replace=False)
        X.append(int(v[0])); Y.append(int(v[1])); Zs.append(tuple(int(t) for t in v[2:]))
    got = eng.test_batch(X, Y, Zs)
    nom = eng.n_obs_min
    big = 0
    for x, y, z, g in zip(X, Y, Zs, got):
        s, pv, df, pw = orc.test(x, y, z, hps=5, n_obs_min=nom)
        assert (g.df, g.suff_power) == (df, pw), (x, y, z, g, (s, pv, df, pw))
        assert _close(g.stat, s, STOL) and _close(g.pval, pv, PTOL), (x, y, z, g, (s, pv, df, pw))
        big += pw
    assert big > 50
    eng.close()
    eng = fw.Engine(kind, n, p, max_k=2)
    eng.set_data(data)
    orc = O.Oracle(kind, data, sparse=True, max_k=2)
    net = eng.lgl(feed_forward=False)
    exp = orc.learn(max_k=2, feed_forward=False)
    assert set(net["edges"]) == set(exp["edges"]) and len(exp["edges"]) > 5
    for e_, w in exp["edges"].items():
        assert _close(net["edges"][e_], w, STOL)
    assert eng.counters()["cond_tests_ref"] == exp["n_cond_tests"]
    eng.close()
    # ... and through the persistent per-target kernel in its 32-bit-count form (r04: dh_mi_target_kernel<.., PRE = 2>; the knob only
    # lifts the "at least 256 targets" rule so that ten variables reach it)
    import os
    os.environ["FW_DEV_MIN_TARGETS"] = "1"
    try:
        eng = fw.Engine(kind, n, p, max_k=2)
        eng.set_data(data)
        net2 = eng.lgl(feed_forward=False, round_size=0)
        assert set(net2["edges"]) == set(exp["edges"])
        for e_, w in exp["edges"].items():
            assert _close(net2["edges"][e_], w, STOL)
        assert eng.counters()["cond_tests_ref"] == exp["n_cond_tests"]
        eng.close()
    finally:
        del os.environ["FW_DEV_MIN_TARGETS"]


def _multi_level_data(n, p, levels, seed, zero_frac=0.3):
    rng = np.random.default_rng(seed)
    lat = rng.standard_normal((n, 4))
    x = lat @ rng.standard_normal((4, p)) + rng.standard_normal((n, p))
    cuts = np.quantile(x, np.linspace(0, 1, levels + 1)[1:-1])
    data = np.digitize(x, cuts).astype(np.int32)            # 0 .. levels - 1
    data[rng.random((n, p)) < zero_frac] = 0
    data[:, 0] = (data[:, 0] > 0).astype(np.int32)          # one binary and one three-level variable among them
    data[:, 1] = np.minimum(data[:, 1], 2)
    return np.ascontiguousarray(data)


@pytest.mark.parametrize("kind,levels,max_k_", [("mi", 4, 3), ("mi_nz", 4, 3), ("mi", 5, 2), ("mi_nz", 5, 2),
                                                # r05: tables beyond a wavefront's LDS slot live in device memory -- six levels at max_k = 3
                                                # (7 992 words), twelve levels (more than the eight of r04) at max_k = 2 (20 880 words)
                                                ("mi", 6, 3), ("mi_nz", 12, 2), ("mi", 12, 1)])
def test_more_than_three_levels(kind, levels, max_k_):
    """Discrete variables with more than three levels (r04: the generic form -- one byte per value, 32-bit LDS tables of L x L x L^k
    cells; the reference sizes its tables for any L, types.jl:98-117, misc.jl:64-97, reachable with make_onehot = false meta data,
    preprocessing.jl:42-117).  Levels / max_vals, single tests of every conditioning-set size, test_subsets jobs, level 0 and the
    network against the oracle (whose table code is the reference's for any L): integers exact, MI 1e-12, p 1e-10."""
    n, p = (700, 40) if levels <= 5 else (3000, 30)          # (more levels: more samples, or no test has power)
    data = _multi_level_data(n, p, levels, 100 + levels)
    max_k = max_k_                                           # L^k (L^2 + 1) <= 3840 words: the table sits in LDS; else in device memory
    eng = fw.Engine(kind, n, p, max_k=max_k)
    eng.set_data(data)
    orc = O.Oracle(kind, data, sparse=True, max_k=max_k)
    lev, mxv = eng.levels()
    olev, omxv = orc.levels()
    assert list(lev) == list(olev) and list(mxv) == list(omxv) and max(mxv) == levels - 1
    nom = eng.n_obs_min
    rng = np.random.default_rng(7)
    X, Y, Zs = [], [], []
    for _ in range(400):
        k = int(rng.integers(0, max_k + 1))
        v = rng.choice(p, size=k + 2, replace=False)
        X.append(int(v[0])); Y.append(int(v[1])); Zs.append(tuple(int(t) for t in v[2:]))
    got = eng.test_batch(X, Y, Zs)
    npow = 0
    for x, y, z, g in zip(X, Y, Zs, got):
        s, pv, df, pw = orc.test(x, y, z, hps=5, n_obs_min=nom)
        assert (g.df, g.suff_power) == (df, pw), (x, y, z, g, (s, pv, df, pw))
        assert _close(g.stat, s, STOL) and _close(g.pval, pv, PTOL), (x, y, z, g, (s, pv, df, pw))
        npow += pw
    assert npow > 50
    # test_subsets jobs
    T, C, A = [], [], []
    for _ in range(60):
        v = rng.choice(p, size=int(rng.integers(3, 9)), replace=False)
        T.append(int(v[0])); C.append(int(v[1])); A.append([int(t) for t in v[2:]])
    res = eng.test_subsets_batch(T, C, A)
    for t, c, a, g in zip(T, C, A, res):
        e = orc.test_subsets(t, c, a, max_k=max_k, alpha=0.01, hps=5, n_obs_min=nom)
        assert g["status"] == e["status"] and g["num_tests"] == e["num_tests"] and g["df"] == e["df"], (t, c, a, g, e)
        assert _close(g["stat"], e["stat"], STOL) and _close(g["pval"], e["pval"], PTOL)
    # level 0 and the network (host job pool: the persistent kernel holds two bit planes per variable)
    got0 = eng.pw_univar_neighbors()
    exp0 = orc.level0(alpha=0.01, hps=5, n_obs_min=nom)
    assert (got0["off"] == exp0["off"]).all() and (got0["idx"] == exp0["idx"]).all() and len(exp0["idx"]) > 10
    assert np.allclose(got0["stat"], exp0["stat"], rtol=1e-12, atol=1e-15) and np.allclose(got0["pval"], exp0["pval"], rtol=1e-10, atol=0.0)
    for ff, R in ((False, 0), (True, 8)):
        net = eng.lgl(feed_forward=ff, round_size=R)
        exp = orc.learn(max_k=max_k, feed_forward=ff, round_size=R)
        assert set(net["edges"]) == set(exp["edges"]) and len(exp["edges"]) > 3, (ff, len(net["edges"]), len(exp["edges"]))
        for e_, w in exp["edges"].items():
            assert _close(net["edges"][e_], w, STOL)
        assert np.array_equal(net["pc_off"], exp["pc_off"]) and np.array_equal(net["pc_idx"], exp["pc_idx"])
    eng.close()


def test_more_than_three_levels_table_limit():
    # L^max_k (L^2 + 1) words per test: up to 3 840 in LDS, up to 64 M (256 MB) in device memory (r05), beyond that FW_ERR_LIMIT, loudly;
    # values up to 61
    data = _multi_level_data(300, 10, 40, 3)
    eng = fw.Engine("mi", 300, 10, max_k=3)                  # 64 000 strata x 1 601 words
    with pytest.raises(fw.FlashWeaveError) as ei:
        eng.set_data(data)
    assert ei.value.code == -5  # FW_ERR_LIMIT
    eng.close()
    eng = fw.Engine("mi", 300, 10, max_k=1)
    eng.set_data(data)                                       # 40 strata x 1 601 words: device-memory tables
    assert eng.levels()[1].max() == 39
    eng.close()
    data[0, 3] = 62
    eng = fw.Engine("mi", 300, 10, max_k=0)
    with pytest.raises(fw.FlashWeaveError) as ei:
        eng.set_data(data)
    assert ei.value.code == -5
    eng.close()
