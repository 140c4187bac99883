"""Repeated passes of one discrete configuration on ONE engine (run as a subprocess, usually two at once on the same GPU: contention
is what makes the schedule of the persistent discrete kernel vary).  The learned network -- directed PC lists, weights, p-values, edge
list -- and the reference-order test count of every pass must be the bytes of the first pass (hiton.jl:109-149 is deterministic).
Prints one JSON line: passes, differing, sha256 of the first pass, its edge count / test count, and the differing passes.

usage: python tests/determinism_worker.py --p 1000 --n 500 --seed 20260930 --kind mi --passes 10000 --seconds 25 --feed-forward 1 --round-size 256 [--save net.npz]
       python tests/determinism_worker.py --config cfg4 --passes 1000            (bench.py's input of that configuration)"""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("FW_KNOBS", "1")

import numpy as np  # noqa: E402

KEYS = ("edge_src", "edge_dst", "edge_weight", "pc_off", "pc_idx", "pc_weight", "pc_pval")


def make_data(a):
    from flashweave_jl_amd import preprocess as pre
    from flashweave_jl_amd import synth
    if a.config:
        import bench

        class B:
            pass
        b = B()
        b.p = b.n = 0
        b.host_normalize = False
        b.single_device = True
        cfg, _, data, _ = bench.make_input(a.config, b)
        return cfg["test_name"], cfg["max_k"], np.ascontiguousarray(data)
    if a.kind == "mi":
        counts = synth.generate(a.p, a.n, a.seed, mode="F")
        data, _, _ = pre.normalize(counts, "mi")
    else:
        counts, meta = synth.generate(a.p, a.n, a.seed, mode="F", habitats=4, n_meta=20)
        data, rm, _ = pre.normalize(counts, "mi_nz")
        meta = meta[rm]
        keep = [j for j in range(meta.shape[1]) if len(np.unique(meta[:, j])) == 2]
        data = np.concatenate([data, meta[:, keep]], axis=1)
    return a.kind, a.max_k, np.ascontiguousarray(data)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="")
    ap.add_argument("--kind", default="mi")
    ap.add_argument("--p", type=int, default=1000)
    ap.add_argument("--n", type=int, default=500)
    ap.add_argument("--seed", type=int, default=20260930)
    ap.add_argument("--max-k", type=int, default=3)
    ap.add_argument("--passes", type=int, default=1000)
    ap.add_argument("--seconds", type=float, default=0.0, help="stop after this many seconds of passes (0: run all passes)")
    ap.add_argument("--feed-forward", type=int, default=1)
    ap.add_argument("--round-size", type=int, default=-1, help="-1: 1024 * ceil(p / 10240) (bench.py's headline schedule)")
    ap.add_argument("--save", default="", help="write the first pass's network (npz) here")
    ap.add_argument("--start-file", default="", help="wait until this file exists before the first pass (two workers start together)")
    a = ap.parse_args()
    import flashweave_jl_amd as fw
    kind, max_k, data = make_data(a)
    n, p = data.shape
    eng = fw.Engine(kind, n, p, max_k=max_k)
    eng.set_data(data)
    R = a.round_size if a.round_size >= 0 else 1024 * ((p + 10239) // 10240)
    ff = bool(a.feed_forward)
    eng.level0()
    eng.lgl(feed_forward=ff, round_size=R if ff else 0, edge_dict=False)  # warm-up (allocations), not compared
    if a.start_file:
        open(a.start_file + ".%d" % os.getpid(), "w").close()
        t_wait = time.time()
        while not os.path.exists(a.start_file) and time.time() - t_wait < 120:
            time.sleep(0.01)
    ref = None
    ref_net = None
    ref_struct = None
    bad = []
    prev = eng.counters()["cond_tests_ref"]
    ref_tests = 0
    t0 = time.time()
    done = 0
    for it in range(a.passes):
        eng.level0()
        net = eng.lgl(feed_forward=ff, round_size=R if ff else 0, edge_dict=False)
        c = eng.counters()["cond_tests_ref"]
        nref, prev = c - prev, c
        h = hashlib.sha256()
        for k in KEYS:
            h.update(np.ascontiguousarray(net[k]).tobytes())
        h.update(str(nref).encode())
        key = h.hexdigest()
        done += 1
        if ref is None:
            ref, ref_tests = key, nref
            ref_net = {k: np.array(net[k]) for k in KEYS}
            hs = hashlib.sha256()  # the integer part alone: edge list, directed lists, test count (the statistics of two arithmetic forms differ in their last bits)
            for k in ("edge_src", "edge_dst", "pc_off", "pc_idx"):
                hs.update(np.ascontiguousarray(net[k]).tobytes())
            hs.update(str(nref).encode())
            ref_struct = hs.hexdigest()
            if a.save:
                np.savez(a.save, n_cond_tests=nref, **ref_net)
        elif key != ref:
            po, qo = ref_net["pc_off"], net["pc_off"]
            tg = [t for t in range(p) if po[t + 1] - po[t] != qo[t + 1] - qo[t]
                  or not np.array_equal(ref_net["pc_idx"][po[t]:po[t + 1]], net["pc_idx"][qo[t]:qo[t + 1]])]
            bad.append({"pass": it, "edges": int(len(net["edge_src"])), "ref_tests": int(nref), "targets": tg[:8]})
        if a.seconds > 0 and time.time() - t0 > a.seconds:
            break
    el = time.time() - t0
    print(json.dumps({"passes": done, "differing": len(bad), "sha256": ref, "sha256_integers": ref_struct, "edges": int(len(ref_net["edge_src"])), "ref_tests": int(ref_tests),
                      "kind": kind, "p": int(p), "n": int(n), "feed_forward": int(ff), "round_size": int(R), "ms_per_pass": 1e3 * el / max(done, 1),
                      "bad": bad[:20]}), flush=True)
    eng.close()


if __name__ == "__main__":
    main()
