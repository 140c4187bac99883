"""Profiling tool: achieved fp32-MFMA rate of fz_cor_gemm_kernel for several (n, p) shapes (wall time of
fw_compute_cor_mat incl. the centring kernel; flops = 2 * n_pad * 128^2 * T(T+1)/2 on the upper-triangular tiles)."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import flashweave_jl_amd as fw
rng = np.random.default_rng(0)
for n, p in ((2000, 10000), (4000, 10000), (8192, 8192), (10000, 20000), (2000, 30000)):
    data = np.asfortranarray(rng.standard_normal((n, p), dtype=np.float32))
    eng = fw.Engine("fz", n, p); eng.set_data(data); eng.compute_cor()
    t = time.perf_counter()
    for _ in range(3):
        eng.compute_cor()
    dt = (time.perf_counter() - t) / 3
    T = (p + 127) // 128; npad = (n + 31) // 32 * 32
    fl = 2.0 * npad * 128 * 128 * T * (T + 1) / 2
    print("n=%d p=%d  %.2f ms  %.1f TFLOP/s (%.0f%% of 157.3)" % (n, p, 1e3 * dt, fl / dt / 1e12, 100 * fl / dt / 157.3e12), flush=True)
    eng.close()
