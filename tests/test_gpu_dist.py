"""Target-sharded run with 2 ranks (both on GPU 0, gloo transport) equals the single-rank run: results do not
depend on the number of ranks."""
import json
import os
import tempfile

import pytest

from tests.test_dist_cpu import _launch

pytestmark = pytest.mark.gpu


def test_sharded_equals_single():
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "res")
        assert _launch("sharded", (out,)) == [0, 0]
        r0 = json.load(open(out + ".0"))
        r1 = json.load(open(out + ".1"))
        for k in ("fz_ff0", "fz_ff1", "mi_ff0", "mi_ff1"):
            assert r0[k] == r1[k]                 # every rank ends with the full network
            assert len(r0[k]) > 0
            if k.startswith("fz"):
                assert r0[k] == r0[k + "_single"]     # and it equals the single-rank network, weights included
            else:
                # discrete: 150 targets per rank run through the host pool (one test per wavefront), the single rank's 300 through
                # the persistent kernel (four per wavefront at n <= 2048): same edges, statistics to the summation order (1e-12)
                assert [e[:2] for e in r0[k]] == [e[:2] for e in r0[k + "_single"]]
                assert all(abs(a[2] - b[2]) <= 1e-12 * abs(b[2]) for a, b in zip(r0[k], r0[k + "_single"]))
        for kind in ("fz", "mi"):
            assert r0[kind + "_l0"] == r1[kind + "_l0"] == r0[kind + "_l0_single"]  # sharded level 0: same lists, bit for bit
            assert len(r0[kind + "_l0"][1]) > 0
            # ... and with the exchange kept in device memory (fw_level0_sharded_dev)
            assert r0[kind + "_l0_dev"] == r1[kind + "_l0_dev"] == r0[kind + "_l0_single"]
            assert r0[kind + "_ff1_devx"] == r1[kind + "_ff1_devx"] == r0[kind + "_ff1"]   # rounds exchanged through fw_learn_network_dev
        assert r0["mi_l0_dev_records"] > 0          # the discrete kind really exchanged its significant pairs
        # row-block sharding of the Pearson matrix: same bits as the single-rank GEMM, same network on it
        assert r0["fz_cor_sharded_equal"] and r1["fz_cor_sharded_equal"]
        assert r0["fz_ff1_sharded_cor"] == r1["fz_ff1_sharded_cor"] == r0["fz_ff1_single"]
