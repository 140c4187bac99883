"""Repeated passes of a discrete configuration (default cfg2) on one engine: the learned network (directed PC lists, weights, p-values) and the
reference-order test count must be identical from pass to pass -- whatever the persistent kernel's helpers, boards and speculation did.
usage: python profiles/tools/determinism_mi.py [passes] [config] [feed_forward]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import bench
import flashweave_jl_amd as fw

class A: pass
args = A(); args.p = 0; args.n = 0; args.host_normalize = False; args.single_device = True
name = sys.argv[2] if len(sys.argv) > 2 else "cfg2"
ff = bool(int(sys.argv[3])) if len(sys.argv) > 3 else True
cfg, csum, data, _ = bench.make_input(name, args)
n, p = data.shape
eng = fw.Engine(cfg["test_name"], n, p, max_k=cfg["max_k"])
eng.set_data(data)
R = 1024 * ((p + 10239) // 10240)
ref = None; bad = 0; prev_ref_tests = 0
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 200):
    import time
    t0 = time.perf_counter(); eng.level0(); t1 = time.perf_counter()
    net = eng.lgl(feed_forward=ff, round_size=R if ff else 0, edge_dict=False); t2 = time.perf_counter()
    if it < 4 or os.environ.get('DET_VERBOSE'): print('pass', it, 'level0 %.1f ms, lgl %.1f ms' % (1e3 * (t1 - t0), 1e3 * (t2 - t1)), flush=True)
    c = eng.counters()["cond_tests_ref"]; nref = c - prev_ref_tests; prev_ref_tests = c
    key = tuple(net[k].tobytes() for k in ("edge_src", "edge_dst", "edge_weight", "pc_off", "pc_idx", "pc_weight", "pc_pval"))
    if ref is None:
        ref = key; ref_net = {k: net[k].copy() for k in net if hasattr(net[k], "copy")}; ref_n = nref
    if key != ref or nref != ref_n:
        bad += 1
        diff_t = [t for t in range(p) if ref_net["pc_off"][t + 1] - ref_net["pc_off"][t] != net["pc_off"][t + 1] - net["pc_off"][t]
                  or not np.array_equal(ref_net["pc_idx"][ref_net["pc_off"][t]:ref_net["pc_off"][t + 1]], net["pc_idx"][net["pc_off"][t]:net["pc_off"][t + 1]])]
        print("pass", it, "DIFFERS: edges", len(net["edge_src"]), "vs", len(ref_net["edge_src"]), "ref tests", nref, "vs", ref_n, "targets with another PC list:", diff_t[:20], flush=True)
print("passes", it + 1, "differing", bad, "edges", len(ref_net["edge_src"]), "ref tests per pass", ref_n)
