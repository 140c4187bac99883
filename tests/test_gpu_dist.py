"""Target-sharded run with 2 and 4 ranks (all on GPU 0, gloo transport) equals the single-rank run: results do not
depend on the number of ranks."""
import json
import os
import tempfile

import numpy as np
import pytest

from tests.test_dist_cpu import _launch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("l0_mfma,world", [("1", 2), ("2", 2), ("1", 4)])
def test_sharded_equals_single(l0_mfma, world, monkeypatch):
    # l0_mfma = "2": the discrete level 0 through the matrix-core kernel whatever the size (its tile list is dealt in super-tile
    # order: mi_level0_mfma_kernel); "1": the default choice (popcount form at this size)
    monkeypatch.setenv("FW_KNOBS", "1")
    monkeypatch.setenv("FW_L0_MFMA", l0_mfma)
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "res")
        assert _launch("sharded", (out,), world=world) == [0] * world
        r0 = json.load(open(out + ".0"))
        r1 = json.load(open(out + "." + str(world - 1)))
        for r in range(1, world - 1):  # (world 4: the ranks in between hold the same networks)
            rm = json.load(open(out + "." + str(r)))
            assert all(rm[k] == r0[k] for k in ("fz_ff0", "fz_ff1", "mi_ff0", "mi_ff1", "fz_l0", "mi_l0", "fz_ff1_sharded_cor"))
        for k in ("fz_ff0", "fz_ff1", "mi_ff0", "mi_ff1"):
            assert r0[k] == r1[k]                 # every rank ends with the full network
            assert len(r0[k]) > 0
            if k.startswith("fz"):
                assert r0[k] == r0[k + "_single"]     # and it equals the single-rank network, weights included
            else:
                # discrete: 150 (75) targets per rank run through the host pool (one test per wavefront), the single rank's 300 through
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


@pytest.mark.parametrize("kind", ["fz", "mi_nz"])
def test_library_side_rccl_world_of_one(kind):
    """The exchanges issued by the LIBRARY on its own RCCL communicator (fw_comm_unique_id / fw_comm_init / fw_level0_comm /
    fw_cor_mat_allgather_comm / fw_learn_network_comm, csrc/fw_rccl.cpp).  ncclCommInitRank refuses two ranks on one device, so on a
    one-GPU box the communicator has ONE rank: what is checked is the whole path -- dlopen of librccl, rendezvous id, communicator,
    header + payload all-gathers on the engine's stream, unpacking -- against the plain single-process calls, bit for bit.  The
    record packing / merging with TWO ranks is covered by the gloo tests above (same fw_dev_exchange contract); more than one rank
    over RCCL has never run (no multi-GPU node)."""
    import flashweave_jl_amd as fw
    from flashweave_jl_amd import preprocess as pre, synth
    mode = "S" if kind == "fz" else "F"
    counts = synth.generate(600, 300, 17, mode=mode)
    data, _, _ = pre.normalize(counts, kind, prec=32) if kind == "fz" else pre.normalize(counts, kind)
    data = np.ascontiguousarray(data)
    n, p = data.shape
    ref = fw.Engine(kind, n, p, max_k=3)
    ref.set_data(data)
    if kind == "fz":
        ref.compute_cor()
    ref.level0()
    nb_ref = ref.pw_univar_neighbors_get()
    net_ref = ref.lgl(feed_forward=True, round_size=128, edge_dict=False)
    ref.close()
    eng = fw.Engine(kind, n, p, max_k=3)
    eng.set_data(data)
    eng.comm_init(fw.Engine.comm_unique_id(), 0, 1)
    if kind == "fz":
        import torch
        rpr = 128 * ((p + 127) // 128)
        buf = torch.empty(rpr * p, dtype=torch.float32, device="cuda:0")
        eng.use_cor_buffer(buf.data_ptr(), buf.numel())
        row0, rows = eng.compute_cor_rows(0, 1)
        assert (row0, rows) == (0, rpr)
        eng.cor_allgather_comm(rpr)      # in-place all-gather of the (one) row block
        eng.cor_ready()
        eng.level0()
    else:
        eng.level0_comm()                # this rank's tiles, significant pairs through ncclAllGather
    nb = eng.pw_univar_neighbors_get()
    for k in ("off", "idx", "stat", "pval"):
        assert np.array_equal(nb[k], nb_ref[k]), k
    net = eng.lgl_comm(feed_forward=True, round_size=128, edge_dict=False)
    for k in ("edge_src", "edge_dst", "edge_weight", "pc_off", "pc_idx", "pc_weight", "pc_pval"):
        assert np.array_equal(net[k], net_ref[k], equal_nan=True), k
    st = eng.comm_stats()
    assert st["calls"] >= (p + 127) // 128 and st["collectives"] >= 2 * st["calls"] - 1 and st["entries"] > 0
    eng.comm_destroy()
    eng.close()
