"""Micro-benchmark of the discrete conditional-test kernel: m random (X, Y | Z) tests of a fixed order k through
fw_test_batch on cfg4-like data (mi_nz, n = 5000).  Usage: python profiles/tools/mi_micro.py [k ...]"""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import flashweave_jl_amd as fw
from flashweave_jl_amd import preprocess as pre, synth

kind = os.environ.get("KIND", "mi_nz")
n, p = int(os.environ.get("N", 5000)), 2000
counts = synth.generate(p, n, 7, mode="F", habitats=4 if kind == "mi_nz" else 0)
data, _, _ = pre.normalize(counts, kind, prec=32)
n, p = data.shape
eng = fw.Engine(kind, n, p, max_k=3)
eng.set_data(data)
rng = np.random.default_rng(1)
m = int(os.environ.get("M", 1 << 18))
for k in [int(a) for a in sys.argv[1:]] or [1, 2, 3]:
    V = np.stack([rng.permutation(p)[:k + 2] for _ in range(4096)])
    V = V[rng.integers(0, 4096, m)]
    X, Y = V[:, 0].astype(np.int32), V[:, 1].astype(np.int32)
    zoff = (np.arange(m + 1) * k).astype(np.int64)
    zflat = np.ascontiguousarray(V[:, 2:]).astype(np.int32).ravel()
    import ctypes as C
    out = (fw.engine._TestResult * m)()
    def run():
        eng._ck(eng.L.fw_test_batch(eng.h, m, X.ctypes.data, Y.ctypes.data, zoff.ctypes.data, zflat.ctypes.data, out))
    run()
    t0 = time.perf_counter(); run(); dt = time.perf_counter() - t0
    pw = sum(1 for i in range(0, m, 64) if out[i].suff_power)
    print("k=%d n=%d: %.2f ms for %d tests = %.3f us/test/GPU, %.0f cycles/test/SIMD (1024 SIMDs @2.4GHz); power ok in %.0f%% of a sample" %
          (k, n, 1e3 * dt, m, 1e6 * dt / m, dt / m * 1024 * 2.4e9, 100.0 * pw / (m / 64)))
