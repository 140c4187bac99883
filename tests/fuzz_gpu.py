"""Randomised end-to-end parity sweep of the HIP path against the CPU oracle (GPU only; not collected by pytest --
tests/test_gpu_fuzz.py runs a fixed handful of these cases, `python -m tests.fuzz_gpu --cases 300` runs many).

Every case draws a kind, a shape, the learn_network keywords (max_k, alpha, max_tests, FDR, feed_forward, round size,
hps) and a schedule (host pool / device rounds, host / device Benjamini-Hochberg) from its seed, learns the network
through the C ABI and compares it with the oracle's: same edge set, same directed PC lists, weights bit-exact (fz,
fz_nz) or within the discrete tolerance, same number of reference-order conditional tests.
"""
import argparse
import os
import sys

import numpy as np

import flashweave_jl_amd as fw
from flashweave_jl_amd import preprocess as pre
from flashweave_jl_amd import synth
from oracle import oracle as O

STOL = 1e-12  # discrete statistics: summation order, device log (tests/test_gpu_mi.py)
PTOL = 1e-10  # discrete p-values: device lgamma / exp in Q(a, x) (tests/test_gpu_mi.py)
ENV_KEYS = ("FW_HOST_HITON", "FW_DEV_MIN_TARGETS", "FW_HOST_BH")
HIGHK = False  # set by --highk
os.environ.setdefault("FW_KNOBS", "1")  # the library reads FW_* knobs only under FW_KNOBS=1 (csrc/fw_internal.h)


def _rel(a, b):
    return abs(a - b) / max(abs(a), abs(b), 1e-300)


def _close(a, b, tol):
    # tol > 0 (discrete kinds): relative, plus an absolute floor of a few ulps of the O(1) terms the statistic is summed
    # from -- a mutual information of 1e-8 is the residue of a cancellation and carries their rounding, not its own
    return (a == b) or (np.isnan(a) and np.isnan(b)) or (tol > 0 and (_rel(a, b) < tol or abs(a - b) < 1e-15))


def draw(seed):
    r = np.random.default_rng(seed)
    c = dict(seed=seed)
    c["kind"] = str(r.choice(["fz", "fz", "fz_nz", "mi", "mi_nz"]))
    c["p"] = int(r.integers(24, 220))
    c["n"] = int(r.integers(60, 420))
    c["max_k"] = int(r.choice([0, 1, 2, 3, 3, 3]))
    c["alpha"] = float(r.choice([0.01, 0.01, 0.05, 0.2]))
    c["max_tests"] = int(r.choice([10_000_000, 10_000_000, 40, 700]))
    c["fdr"] = bool(r.integers(0, 2))
    c["hps"] = int(r.choice([5, 5, 3, 10]))
    c["ff"] = bool(r.integers(0, 2))
    c["R"] = int(r.choice([1, 7, 64, 1000])) if c["ff"] else 0
    c["source"] = str(r.choice(["synth", "synth", "factor"])) if c["kind"] in ("fz",) else "synth"
    c["dups"] = bool(r.integers(0, 5) == 0) and c["kind"] == "fz"
    if c["kind"] == "fz" and c["source"] == "synth" and r.integers(0, 8) == 0:  # occasionally a wider problem
        c["p"] = int(r.integers(300, 800))
        c["n"] = int(r.integers(100, 700))
    c["env"] = dict(FW_HOST_HITON=str(int(r.integers(0, 3) == 0)), FW_DEV_MIN_TARGETS=str(int(r.choice([1, 16, 64]))),
                    FW_HOST_BH=str(int(r.integers(0, 4) == 0)))
    if HIGHK:  # --highk: Fisher-z networks with max_k 4 / 5 (level-2 / level-3 table kernels, device rounds with local matrices); jobs capped so that the oracle finishes
        c["kind"] = "fz"
        c["max_k"] = int(r.choice([4, 5]))
        c["source"] = str(r.choice(["synth", "factor", "factor"]))
        c["p"] = int(r.integers(24, 260))
        c["n"] = int(r.integers(60, 420))
        c["alpha"] = float(r.choice([0.01, 0.05, 0.2, 0.5]))
        c["max_tests"] = int(r.choice([40, 700, 700, 5000, 30000])) if c["p"] > 48 else 10_000_000
        c["dups"] = False
        c["env"]["FW_HOST_HITON"] = str(int(r.integers(0, 5) == 0))
    return c


def make_data(c):
    kind, p, n, seed = c["kind"], c["p"], c["n"], c["seed"]
    if kind == "fz":
        if c["source"] == "factor":
            r = np.random.default_rng(seed + 1)
            nf = int(r.integers(1, 4))
            load = r.standard_normal((nf, p)) * r.choice([0.3, 0.9])
            data = (r.standard_normal((n, nf)) @ load + r.standard_normal((n, p))).astype(np.float32)
        else:
            counts = synth.generate(p, n, seed, mode="S")
            data, _, _ = pre.normalize(counts, "fz", prec=32)
        data = np.array(data, dtype=np.float32)
        if c["dups"] and data.shape[1] > 8:  # exact duplicates (|r| = 1 up to rounding, exact ties) and a constant column
            data[:, 3] = data[:, 1]
            data[:, 5] = data[:, 2]
            data[:, 7] = 0.25
            # (not a sign-flipped copy: r and -r give level-0 p-values that differ in the 14th digit -- log((1+r)/(1-r))
            # is not odd in floating point -- so the order of the two candidates hangs on the last bit of log/erfc,
            # which differs between glibc (oracle), the device library and Julia)
        return np.asfortranarray(data)
    if kind == "fz_nz":
        counts = synth.generate(p, n, seed, mode="S", habitats=int(1 + seed % 4))
        data, _, _ = pre.normalize(counts, "fz_nz", prec=32)
        return np.asfortranarray(data)
    if kind == "mi":
        counts = synth.generate(p, n, seed, mode="F")
        data, _, _ = pre.normalize(counts, "mi")
        return np.ascontiguousarray(data)
    counts, meta = synth.generate(p, n, seed, mode="F", habitats=4, n_meta=20)
    data, rm, _ = pre.normalize(counts, "mi_nz")
    meta = meta[rm]
    keep = [j for j in range(meta.shape[1]) if len(np.unique(meta[:, j])) == 2]
    return np.ascontiguousarray(np.concatenate([data, meta[:, keep]], axis=1))


def run_case(seed):
    """Returns (case, None) on parity, (case, message) on the first difference."""
    c = draw(seed)
    data = make_data(c)
    n, p = data.shape
    if n < 10 or p < 4:
        return c, None
    old = {k: os.environ.get(k) for k in ENV_KEYS}
    os.environ.update(c["env"])
    eng = orc = None
    try:
        kw = dict(max_k=c["max_k"], alpha=c["alpha"], hps=c["hps"], max_tests=c["max_tests"], FDR=c["fdr"])
        eng = fw.Engine(c["kind"], n, p, **kw)
        eng.set_data(data)
        if c["kind"] == "fz":
            cm = eng.cor()
            orc = O.Oracle("fz", cor_mat=cm, n_obs=n)
        elif c["kind"] == "fz_nz":
            orc = O.Oracle("fz_nz", data=data.astype(np.float64))
        else:
            orc = O.Oracle(c["kind"], data, sparse=True, max_k=c["max_k"])
        try:
            got = eng.lgl(feed_forward=c["ff"], round_size=c["R"])
            gerr = None
        except fw.FlashWeaveError as e:
            got, gerr = None, e
        try:
            exp = orc.learn(feed_forward=c["ff"], round_size=max(c["R"], 1) if c["ff"] else 1, **kw)
            eerr = None
        except ValueError as e:
            exp, eerr = None, e
        if (gerr is None) != (eerr is None):
            return c, "error behaviour differs: hip %r oracle %r" % (gerr, eerr)
        if gerr is not None:
            return c, None
        tol = 0.0 if c["kind"] in ("fz", "fz_nz") else STOL
        if set(got["edges"]) != set(exp["edges"]):
            d = set(got["edges"]) ^ set(exp["edges"])
            return c, "edge sets differ (%d of %d): %s" % (len(d), len(exp["edges"]), sorted(d)[:5])
        for e, w in exp["edges"].items():
            if not _close(got["edges"][e], w, tol):
                return c, "weight of %s: %r vs %r" % (e, got["edges"][e], w)
        if not np.array_equal(got["pc_off"], exp["pc_off"]) or not np.array_equal(got["pc_idx"], exp["pc_idx"]):
            return c, "directed PC lists differ"
        for a, b in zip(got["pc_weight"], exp["pc_weight"]):
            if not _close(a, b, tol):
                return c, "directed weight %r vs %r" % (a, b)
        cn = eng.counters()
        if cn["cond_tests_ref"] != exp["n_cond_tests"]:
            return c, "reference-order test count %d vs %d" % (cn["cond_tests_ref"], exp["n_cond_tests"])
        c["edges"] = len(exp["edges"])
        c["cond"] = exp["n_cond_tests"]
        return c, None
    finally:
        if eng is not None:
            eng.close()
        if orc is not None:
            orc.close()
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _n_enum(a, max_k):
    from math import comb
    return sum(comb(a, s) for s in range(1, max_k + 1))


def run_subsets_case(seed, jobs=48, budget=300_000):
    """test_subsets batches (tests.jl:281-346) with random accepted lists: status, number of tests in reference order,
    winning conditioning set, statistic (bit-exact for fz) and p-value against the oracle."""
    r = np.random.default_rng(1_000_003 + seed)
    c = dict(seed=seed, kind=str(r.choice(["fz", "fz", "fz", "mi", "mi_nz"])))
    c["max_k"] = int(r.choice([1, 2, 3, 3, 3, 4, 5])) if c["kind"] == "fz" else int(r.choice([1, 2, 3, 3]))
    c["alpha"] = float(r.choice([0.01, 0.01, 0.2, 0.9999]))
    c["max_tests"] = int(r.choice([10_000_000, 10_000_000, int(r.integers(5, 60_000))]))
    c["hps"] = int(r.choice([5, 5, 3]))
    if c["kind"] == "fz":
        p, n = int(r.integers(60, 900)), int(r.choice([40, 150, 400, 3000]))
        nf = int(r.integers(1, 4))
        load = r.standard_normal((nf, p)) * r.choice([0.2, 0.6, 1.5])
        data = np.asfortranarray((r.standard_normal((n, nf)) @ load + r.standard_normal((n, p))).astype(np.float32))
        lens = [0, 1, 2, 3, 4, 7, 17, 40, 63, 64, 65, 120, 300, 511, 512, 513, 700]
    else:
        c2 = dict(c, p=int(r.integers(40, 200)), n=int(r.integers(80, 420)), source="synth", dups=False)
        data = make_data(c2)
        lens = [0, 1, 2, 3, 4, 7, 12, 20, 33]
    n, p = data.shape
    c["p"], c["n"] = p, n
    eng = orc = None
    try:
        kw = dict(max_k=c["max_k"], alpha=c["alpha"], hps=c["hps"], max_tests=c["max_tests"])
        eng = fw.Engine(c["kind"], n, p, **kw)
        eng.set_data(data)
        if c["kind"] == "fz":
            cm = eng.cor()
            orc = O.Oracle("fz", cor_mat=cm, n_obs=n)
        else:
            orc = O.Oracle(c["kind"], data, sparse=True, max_k=c["max_k"])
        nom = eng.n_obs_min
        T, Cn, A = [], [], []
        for _ in range(jobs):
            ok = [a for a in lens if a + 2 <= p and min(_n_enum(a, c["max_k"]), c["max_tests"]) <= budget]
            a = int(r.choice(ok))
            if c["kind"] == "fz" and r.integers(0, 3) == 0:  # strongest neighbours: long all-significant runs
                t = int(r.integers(0, p))
                order = [int(v) for v in np.argsort(-np.abs(np.nan_to_num(cm[t]))) if v != t][:a + 1]
                v = [t] + order
            else:
                v = [int(x) for x in r.choice(p, size=a + 2, replace=False)]
            acc = v[2:]
            # duplicate entry (feed-forward whitelist).  fz only: two orderings of the same conditioning set
            # tie exactly there (bit-exact arithmetic on both sides); the discrete sums run in a different order on the
            # device, so which of the two tied orderings attains the maximum p is a rounding matter (tolerance class)
            if c["kind"] == "fz" and len(acc) > 2 and r.integers(0, 6) == 0:
                acc = acc + acc[:1]
            T.append(v[0]); Cn.append(v[1]); A.append(acc)
        got = eng.test_subsets_batch(T, Cn, A)
        tol = 0.0 if c["kind"] == "fz" else STOL
        for t, cd, a, g in zip(T, Cn, A, got):
            e = orc.test_subsets(t, cd, a, max_k=c["max_k"], alpha=c["alpha"], hps=c["hps"], n_obs_min=nom,
                                 max_tests=c["max_tests"])
            where = "T %d cand %d |accepted| %d: hip %r oracle %r" % (t, cd, len(a), g, e)
            if g["status"] != e["status"] or g["num_tests"] != e["num_tests"]:
                return c, "status / num_tests: " + where
            if e["status"] == 0:
                continue
            if g["df"] != e["df"] or g["suff_power"] != e["suff_power"]:
                return c, "df / power: " + where
            if g["Zs"] != e["Zs"]:
                # discrete kinds, no test stopped the job (status 2 = maximum p over the enumeration): two subsets whose
                # tables agree cell for cell (a conditioning variable that is constant on the rows in play) tie exactly
                # in exact arithmetic; which of them attains the maximum is decided by the last bits of the sums
                tie = tol > 0 and e["status"] == 2 and _close(g["stat"], e["stat"], tol) and _close(g["pval"], e["pval"], PTOL)
                if not tie:
                    return c, "conditioning set: " + where
            if not _close(g["stat"], e["stat"], tol) or not _close(g["pval"], e["pval"], PTOL if tol else 1e-12):
                return c, "statistic / p-value: " + where
        c["jobs"] = jobs
        return c, None
    finally:
        if eng is not None:
            eng.close()
        if orc is not None:
            orc.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=100)
    ap.add_argument("--first", type=int, default=0)
    ap.add_argument("--verbose", action="store_true")
    ap.add_argument("--subsets", action="store_true", help="test_subsets batches instead of whole networks")
    ap.add_argument("--highk", action="store_true", help="whole networks with fz, max_k 4 / 5")
    a = ap.parse_args()
    global HIGHK
    HIGHK = a.highk
    bad = 0
    tot_edges = tot_cond = 0
    if a.subsets:
        for s in range(a.first, a.first + a.cases):
            c, msg = run_subsets_case(s)
            if msg:
                bad += 1
                print("FAIL seed %d: %s\n     %s" % (s, msg, c), flush=True)
            elif a.verbose:
                print("ok   seed %d: %s" % (s, c), flush=True)
        print("%d test_subsets cases, %d failures" % (a.cases, bad))
        return 1 if bad else 0
    for s in range(a.first, a.first + a.cases):
        c, msg = run_case(s)
        tot_edges += c.get("edges", 0)
        tot_cond += c.get("cond", 0)
        if msg:
            bad += 1
            print("FAIL seed %d: %s\n     %s" % (s, msg, c), flush=True)
        elif a.verbose:
            print("ok   seed %d: %s" % (s, c), flush=True)
    print("%d cases, %d failures, %d edges and %d conditional tests compared" % (a.cases, bad, tot_edges, tot_cond))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
