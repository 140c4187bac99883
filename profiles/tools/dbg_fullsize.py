import sys, numpy as np
sys.path.insert(0, '.')
import flashweave_jl_amd as fw
from flashweave_jl_amd import preprocess as pre, synth
from oracle import oracle as O
c = synth.CONFIGS["cfg3"]
counts = synth.generate(c["p"], c["n"], c["seed"], mode=c["mode"])
data, _, _ = pre.normalize(counts, "fz", prec=32)
n, p = data.shape
eng = fw.Engine("fz", n, p, max_k=3); eng.set_data(data); cm = eng.cor()
got0 = eng.pw_univar_neighbors()
orc = O.Oracle("fz", cor_mat=cm, n_obs=n)
r1 = eng.lgl(feed_forward=False)
exp = orc.learn(max_k=3, feed_forward=False, max_targets=6000)
off, idx, w, pp = r1["pc_off"], r1["pc_idx"], r1["pc_weight"], r1["pc_pval"]
eoff, eidx, ew, ep = exp["pc_off"], exp["pc_idx"], exp["pc_weight"], exp["pc_pval"]
deg = np.diff(got0["off"]); order = np.argsort(deg, kind="stable")[:6000]
bad = []
for T in order:
    a = list(idx[off[T]:off[T+1]]); b = list(eidx[eoff[T]:eoff[T+1]])
    if a != b or list(w[off[T]:off[T+1]]) != list(ew[eoff[T]:eoff[T+1]]):
        bad.append(int(T))
print("n bad", len(bad), bad[:10])
for T in bad[:3]:
    print("T", T, "deg", deg[T])
    print(" gpu", list(zip(idx[off[T]:off[T+1]], w[off[T]:off[T+1]], pp[off[T]:off[T+1]])))
    print(" orc", list(zip(eidx[eoff[T]:eoff[T+1]], ew[eoff[T]:eoff[T+1]], ep[eoff[T]:eoff[T+1]])))
    # replay interleaving with the ABI test_subsets against the oracle
    o = got0["off"]; nb = got0["idx"][o[T]:o[T+1]]; nbp = got0["pval"][o[T]:o[T+1]]
    cands = [int(x) for x in nb[np.argsort(nbp, kind="stable")]]
    acc = []
    for cnd in cands:
        if not acc:
            acc.append(cnd); continue
        g = eng.test_subsets(T, cnd, acc); e = orc.test_subsets(T, cnd, acc, max_k=3, alpha=0.01, n_obs_min=20)
        same = (g["status"], g["num_tests"], g["Zs"], g["stat"]) == (e["status"], e["num_tests"], e["Zs"], e["stat"])
        if not same:
            print("  MISMATCH cand", cnd, "acc", acc, "\n   gpu", g, "\n   orc", e)
        if e["pval"] < 0.01: acc.append(cnd)
exp0 = orc.level0(alpha=0.01, n_obs_min=20)
gp, epv = got0["pval"], exp0["pval"]
relv = np.abs(gp - epv) / np.maximum(np.abs(epv), 1e-320)
print("level0 adj-p max rel diff", relv.max(), "n exact equal", (gp == epv).sum(), "of", len(gp))
for T in bad[:2]:
    o = got0["off"]
    print("T", T)
    for q in range(o[T], o[T+1]):
        print("   nbr", got0["idx"][q], "gpu adjp %.17g orc adjp %.17g stat %.9g" % (gp[q], epv[q], got0["stat"][q]))
