"""Debug aid for one tests/fuzz_gpu.py case: prints where the HIP path and the oracle part ways."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import flashweave_jl_amd as fw
from oracle import oracle as O
from tests import fuzz_gpu as F

seed = int(sys.argv[1])
c = F.draw(seed)
print(c)
data = F.make_data(c)
n, p = data.shape
os.environ.update(c["env"])
kw = dict(max_k=c["max_k"], alpha=c["alpha"], hps=c["hps"], max_tests=c["max_tests"], FDR=c["fdr"])
eng = fw.Engine(c["kind"], n, p, **kw)
eng.set_data(data)
cm = eng.cor()
orc = O.Oracle("fz", cor_mat=cm, n_obs=n)
got = eng.lgl(feed_forward=c["ff"], round_size=c["R"])
exp = orc.learn(feed_forward=c["ff"], round_size=max(c["R"], 1) if c["ff"] else 1, **kw)
print("edge diff", sorted(set(got["edges"]) ^ set(exp["edges"])))
l0g = eng.pw_univar_neighbors()
l0e = orc.level0(alpha=c["alpha"], hps=c["hps"], n_obs_min=0, FDR=c["fdr"])
print("level0 equal:", [bool(np.array_equal(l0g[k], l0e[k])) for k in ("off", "idx", "stat")],
      "max rel p", np.max(np.abs(l0g["pval"] - l0e["pval"]) / np.maximum(l0e["pval"], 1e-300)) if len(l0e["pval"]) else 0)
bad = []
for t in range(p):
    g = got["pc_idx"][got["pc_off"][t]:got["pc_off"][t + 1]]
    e = exp["pc_idx"][exp["pc_off"][t]:exp["pc_off"][t + 1]]
    if not np.array_equal(g, e):
        bad.append(t)
        if len(bad) <= 4:
            print("target", t, "\n  hip   ", list(g), "\n  oracle", list(e))
            o0, o1 = l0e["off"][t], l0e["off"][t + 1]
            order = np.argsort(l0e["pval"][o0:o1], kind="stable")
            print("  level-0 candidates (oracle p order):", [(int(l0e["idx"][o0 + i]), float(l0e["pval"][o0 + i]), float(l0g["pval"][o0 + i])) for i in order][:12])
print("targets with different PC:", bad)
print("cm[2,5], cm[1,3]:", repr(float(cm[2, 5])), repr(float(cm[1, 3])))
