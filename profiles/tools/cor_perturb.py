"""How much of the cfg3 network hangs on the last bits of the Pearson matrix?  The device GEMM accumulates in Float32 (MFMA), a host
`cor` in Float64 (|difference| <= 5e-6, tests/test_gpu_fullsize.py); pcor_rec rounds to 5 digits at every level, so a matrix entry
that moves across a rounding boundary moves a partial correlation by 1e-5.  Learns the headline network (feed_forward = 1, R = 1024)
once on the device's matrix and once on numpy's Float64 correlation matrix rounded to Float32 -> JSON on stdout."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
import flashweave_jl_amd as fw
args = bench.parse_args([])
cfg, _, data, _ = bench.make_input("cfg3", args)
n, p = data.shape
eng = fw.Engine("fz", n, p, max_k=3)
eng.set_data(data)
cm = eng.cor()
a = eng.lgl(feed_forward=True, round_size=1024)["edges"]
d64 = data.astype(np.float64)
d64 -= d64.mean(axis=0)
d64 /= np.sqrt((d64 * d64).sum(axis=0))
c64 = d64.T @ d64
c64 = 0.5 * (c64 + c64.T)
np.fill_diagonal(c64, 1.0)
c32 = np.clip(c64, -1.0, 1.0).astype(np.float32)
diff = np.abs(c32.astype(np.float64) - cm.astype(np.float64))
eng.set_cor_mat(c32)
b = eng.lgl(feed_forward=True, round_size=1024)["edges"]
ea, eb = set(a), set(b)
common = ea & eb
wd = np.array([abs(a[e] - b[e]) for e in common])
print(json.dumps({"workload": cfg.get("label") or "cfg3", "edges_device_matrix": len(ea), "edges_float64_host_matrix": len(eb),
                  "only_device": len(ea - eb), "only_host": len(eb - ea), "common": len(common),
                  "matrix_max_abs_diff": float(diff.max()), "matrix_entries_differing": int((diff > 0).sum()), "matrix_entries": int(diff.size),
                  "weight_max_abs_diff_on_common_edges": float(wd.max()), "weights_differing_on_common_edges": int((wd > 0).sum()),
                  "weights_differing_by_more_than_1e-5": int((wd > 1e-5).sum())}))
