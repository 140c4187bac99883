"""Runs one oracle.learn(...) call on inputs another test process saved (tests/test_gpu_fullsize.py::_oracle_in_background): the
whole-schedule oracle runs of the full-size tests take minutes of wall time (the chain of each round's heaviest target) and read
nothing but a matrix, so they run beside the other tests instead of in front of them.  Test infrastructure: executes the oracle,
never the product.  usage: python tests/oracle_worker.py <dir with args.json + *.npy>  ->  <dir>/result.npz"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

from oracle import oracle as O  # noqa: E402


def main():
    try:
        os.nice(10)  # the tests of the foreground process (their own oracle calls, numpy) keep the cores they need; this run takes the idle ones
    except OSError:
        pass
    d = sys.argv[1]
    a = json.load(open(os.path.join(d, "args.json")))
    if a["kind"] in ("fz",):
        orc = O.Oracle("fz", cor_mat=np.load(os.path.join(d, "cor_mat.npy")), n_obs=a["n"])
    else:
        raise SystemExit("oracle_worker: kind %r not wired" % a["kind"])
    exp = orc.learn(**a["learn"])
    e = exp["edges"]
    src = np.array([k[0] for k in e], np.int32)
    dst = np.array([k[1] for k in e], np.int32)
    w = np.array(list(e.values()), np.float64)
    np.savez(os.path.join(d, "result.npz"), pc_off=exp["pc_off"], pc_idx=exp["pc_idx"], pc_weight=exp["pc_weight"], pc_pval=exp["pc_pval"],
             n_cond_tests=np.int64(exp["n_cond_tests"]), edge_src=src, edge_dst=dst, edge_weight=w)
    orc.close()


if __name__ == "__main__":
    main()
