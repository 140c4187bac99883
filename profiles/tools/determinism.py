"""Repeated cfg3 passes on one engine: the learned network (directed PC lists, weights, p-values) must be bit-identical
from pass to pass (the device rounds merge in rank order whatever the segmentation and the look-ahead did)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import bench
import flashweave_jl_amd as fw

class A: pass
args = A(); args.p = 0; args.n = 0
cfg, csum, data, _ = bench.make_input("cfg3", args)
n, p = data.shape
eng = fw.Engine("fz", n, p, max_k=3)
eng.set_data(data)
ref = None
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    eng.compute_cor(); eng.level0()
    net = eng.lgl(feed_forward=False, round_size=0, edge_dict=False)
    key = tuple(net[k].tobytes() for k in ("edge_src", "edge_dst", "edge_weight", "pc_off", "pc_idx", "pc_weight", "pc_pval"))
    if ref is None:
        ref = key
    print("pass", it, "edges", len(net["edge_src"]), "identical to pass 0:", key == ref, flush=True)
    assert key == ref
print("deterministic")
