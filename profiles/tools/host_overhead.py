"""Where the wall time of one bench step goes on the host side (cfg3): C-ABI calls vs Python glue."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import ctypes as C
import bench
import flashweave_jl_amd as fw
from flashweave_jl_amd import engine as E

class A: pass
args = A(); args.p = 0; args.n = 0
cfg, csum, data, _ = bench.make_input("cfg3", args)
n, p = data.shape
eng = fw.Engine("fz", n, p, max_k=3)
eng.set_data(data)
for it in range(3):
    t0 = time.perf_counter(); eng.compute_cor(); t1 = time.perf_counter(); eng.level0(); t2 = time.perf_counter()
    opts = E._LearnOpts(0, 0, 0, 1, 0, 0); ne = C.c_int64(0)
    eng._ck(eng.L.fw_learn_network(eng.h, C.byref(opts), None, None, C.byref(ne))); t3 = time.perf_counter()
    net = eng.lgl(feed_forward=False, round_size=0); t4 = time.perf_counter()
    print("cor %.1f ms  level0 %.1f ms  fw_learn_network (C ABI) %.1f ms  Engine.lgl (C ABI + python result objects) %.1f ms  edges %d"
          % (1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2), 1e3 * (t4 - t3), len(net["edges"])))
