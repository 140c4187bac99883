"""Streaming Fisher-z kernel (fw_fzs.hip, recursive_pcor = 0) on uniform batches of whole enumerations: jobs of |accepted| = A over the
cfg3 matrix (10 000 x 2 000, clr_adapt), max_k = 3, alpha = 0.9999 (nearly every test "significant": the enumeration runs to its end).
Prints tests/s and the nominal B_fzS rate ((k + 2) n 4 + 32 bytes per test).   python profiles/tools/fzs_micro.py [A] [jobs]"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
import flashweave_jl_amd as fw
A = int(sys.argv[1]) if len(sys.argv) > 1 else 40
J = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
args = bench.parse_args([])
cfg, _, data, _ = bench.make_input("cfg3", args)
n, p = data.shape
eng = fw.Engine("fz", n, p, max_k=3, alpha=0.9999, recursive_pcor=False)
eng.set_data(data)
rng = np.random.default_rng(1)
T, C, acc = [], [], []
for _ in range(J):
    v = rng.choice(p, size=A + 2, replace=False)
    T.append(int(v[0])); C.append(int(v[1])); acc.append([int(x) for x in v[2:]])
eng.test_subsets_batch(T[:50], C[:50], acc[:50])  # warm-up (column statistics, buffers)
eng.reset_counters()
t0 = time.perf_counter()
res = eng.test_subsets_batch(T, C, acc)
dt = time.perf_counter() - t0
cn = eng.counters()
ev = cn["cond_tests_evaluated"]
ks = cn["t_dev_subsets_s"]
print(json.dumps({"accepted": A, "jobs": J, "n": n, "evaluated_tests": ev, "ref_tests": sum(r["num_tests"] for r in res), "wall_s": dt,
                  "kernel_s": ks, "tests_per_s_in_kernel": ev / ks, "alg_bytes": cn["alg_bytes_subsets"],
                  "nominal_GBps": cn["alg_bytes_subsets"] / ks / 1e9, "frac_of_8TBps": cn["alg_bytes_subsets"] / ks / 8e12,
                  "launches": cn["subsets_launches"]}))
