"""Profiling tool (not a test): kernel throughput on a fixed all-significant test_subsets workload (no early
exits), for A/B comparisons of fz_subsets_seg_kernel variants.  Round-1 ablation of the first kernel (one
unranking per test): full 16.8 ms / no log+erfc 12.4 / no DP 10.5 / no unranking 5.8 / skeleton 2.3 per 1.28e8
tests (a = 40) -- unranking dominated, which led to the run-based kernel."""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import numpy as np
    import flashweave_jl_amd as fw
    rng = np.random.default_rng(0)
    n, p, a, m = 2000, 4000, int(os.environ.get("ABL_A", "40")), int(os.environ.get("ABL_M_OVERRIDE", os.environ.get("ABL_M", "4000")))
    base = rng.standard_normal((n, 6))
    data = (base @ rng.standard_normal((6, p)) + 1.5 * rng.standard_normal((n, p))).astype(np.float32)
    eng = fw.Engine("fz", n, p, max_k=int(os.environ.get("ABL_K", "3")), alpha=0.999999)
    eng.set_data(data); eng.cor()
    T = rng.integers(0, p, m); C = (T + 1 + rng.integers(0, p - 1, m)) % p
    A = [list(rng.choice(p, a, replace=False)) for _ in range(m)]
    eng.test_subsets_batch(T, C, A); eng.reset_counters()
    for _ in range(3):
        eng.test_subsets_batch(T, C, A)
    c = eng.counters()
    print(json.dumps(dict(a=a, tests=c["cond_tests_evaluated"], kernel_s=c["t_dev_subsets_s"],
                          tests_per_s=c["cond_tests_evaluated"] / c["t_dev_subsets_s"], launches=c["subsets_launches"])))
else:
    k = int(os.environ.get("ABL_K", "3"))
    sizes = (12, 40, 120) if k <= 3 else ((100, 200) if os.environ.get("ABL_LONG") else (16, 30, 45))
    for a in sizes:
        env = dict(os.environ, ABL_A=str(a), ABL_M=str(max(200, 160000 // (a * a)) if k <= 3 else (2000 if a <= 16 else 200 if a <= 30 else 40 if a <= 45 else 2)))
        subprocess.run([sys.executable, __file__, "child"], env=env)
