"""Worker for the world_size-2 and -4 tests (launched by tests/test_dist_cpu.py and tests/test_gpu_dist.py)."""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run_callback_only(rank, world):
    import torch
    import torch.distributed as dist
    from flashweave_jl_amd.dist import make_allgather
    from flashweave_jl_amd.engine import ALLGATHER_FN
    dist.init_process_group("gloo", rank=rank, world_size=world)
    stats = {}
    # capacity 2: rounds with more than two entries on a rank take the grow-and-repeat path
    cb = ALLGATHER_FN(make_allgather(dist, torch.device("cpu"), capacity=2, stats=stats))
    ok = True
    for rnd in range(6):
        n = [3, 0, 5][(rank + rnd) % 3]  # ragged, including an empty contribution
        t = np.arange(n, dtype=np.int32) + 100 * rank + 10 * rnd
        u = np.arange(n, dtype=np.int32) + 7
        s = np.linspace(-1, 1, n) if n else np.zeros(0)
        s = s.astype(np.float64)
        if n:
            s[0] = np.nan
        p = np.full(n, 1e-300 * (rank + 1))
        n_total = C.c_int64(0)
        pt, pn = C.POINTER(C.c_int32)(), C.POINTER(C.c_int32)()
        ps, pp = C.POINTER(C.c_double)(), C.POINTER(C.c_double)()
        rc = cb(None, n, t.ctypes.data_as(C.POINTER(C.c_int32)), u.ctypes.data_as(C.POINTER(C.c_int32)),
                s.ctypes.data_as(C.POINTER(C.c_double)), p.ctypes.data_as(C.POINTER(C.c_double)), C.byref(n_total),
                C.byref(pt), C.byref(pn), C.byref(ps), C.byref(pp))
        ok &= rc == 0
        exp_t = []
        for r in range(world):
            nr = [3, 0, 5][(r + rnd) % 3]
            exp_t += list(np.arange(nr) + 100 * r + 10 * rnd)
        got_t = [pt[i] for i in range(n_total.value)]
        ok &= got_t == exp_t
        got_p = [pp[i] for i in range(n_total.value)]
        ok &= all(v in [1e-300 * (r + 1) for r in range(world)] for v in got_p)
        ok &= sum(1 for i in range(n_total.value) if np.isnan(ps[i])) == sum(1 for r in range(world) if [3, 0, 5][(r + rnd) % 3])
    # one collective per round once the capacity has grown (rounds 0..2 may repeat once)
    ok &= stats["calls"] == 6 and 6 <= stats["collectives"] <= 8
    # a NEGATIVE target (the marker record fw_level0_sharded sends: target -1, neighbour = rank, statistic = reliable-test count) must
    # come back as -1 with its neighbour half intact (r02 packed t | u << 32 without masking: the sign extension wiped the rank)
    t = np.array([-1, 5], dtype=np.int32)
    u = np.array([rank, 123456], dtype=np.int32)
    s = np.array([1e9 + rank, -0.5], dtype=np.float64)
    p = np.array([0.0, 1e-310], dtype=np.float64)  # a subnormal p-value travels as its bits
    n_total = C.c_int64(0)
    pt, pn = C.POINTER(C.c_int32)(), C.POINTER(C.c_int32)()
    ps, pp = C.POINTER(C.c_double)(), C.POINTER(C.c_double)()
    rc = cb(None, 2, t.ctypes.data_as(C.POINTER(C.c_int32)), u.ctypes.data_as(C.POINTER(C.c_int32)),
            s.ctypes.data_as(C.POINTER(C.c_double)), p.ctypes.data_as(C.POINTER(C.c_double)), C.byref(n_total),
            C.byref(pt), C.byref(pn), C.byref(ps), C.byref(pp))
    ok &= rc == 0 and n_total.value == 2 * world
    for r in range(world):
        ok &= pt[2 * r] == -1 and pn[2 * r] == r and ps[2 * r] == 1e9 + r
        ok &= pt[2 * r + 1] == 5 and pn[2 * r + 1] == 123456 and pp[2 * r + 1] == 1e-310
    dist.barrier()
    dist.destroy_process_group()
    return ok


def run_sharded_gpu(rank, world, out_path):
    """`world` ranks on the same GPU (device 0), gloo transport: the sharded run must equal the single-rank run."""
    import torch
    import torch.distributed as dist
    import flashweave_jl_amd as fw
    from flashweave_jl_amd import preprocess as pre
    from flashweave_jl_amd import synth
    from flashweave_jl_amd.dist import make_allgather
    dist.init_process_group("gloo", rank=rank, world_size=world)
    res = {}
    for kind, mode in (("fz", "S"), ("mi", "F")):
        counts = synth.generate(300, 250, 17, mode=mode)
        data, _, _ = pre.normalize(counts, kind)
        n, p = data.shape
        eng = fw.Engine(kind, n, p, max_k=3)
        eng.set_data(data)
        cb = make_allgather(dist, torch.device("cpu"))
        # level 0 with the pair tiles sharded over the ranks (discrete kinds) == the single-rank neighbour lists, bit for bit
        eng.level0(rank=rank, world_size=world, allgather=cb)
        nb = eng.pw_univar_neighbors_get()
        res["%s_l0" % kind] = [nb["off"].tolist(), nb["idx"].tolist(), nb["stat"].tolist(), nb["pval"].tolist()]
        if rank == 0:
            single = fw.Engine(kind, n, p, max_k=3)
            single.set_data(data)
            nb1 = single.pw_univar_neighbors()
            res["%s_l0_single" % kind] = [nb1["off"].tolist(), nb1["idx"].tolist(), nb1["stat"].tolist(), nb1["pval"].tolist()]
            single.close()
        for ff, R in ((0, 0), (1, 32)):
            sh = eng.lgl(feed_forward=bool(ff), round_size=R, rank=rank, world_size=world, allgather=cb)
            res["%s_ff%d" % (kind, ff)] = sorted([a, b, w] for (a, b), w in sh["edges"].items())
            if rank == 0:
                single = fw.Engine(kind, n, p, max_k=3)
                single.set_data(data)
                one = single.lgl(feed_forward=bool(ff), round_size=R)
                res["%s_ff%d_single" % (kind, ff)] = sorted([a, b, w] for (a, b), w in one["edges"].items())
                single.close()
        # r03: the same exchange with the payload kept in device memory (fw_level0_sharded_dev: pack / unpack kernels around a
        # collective on torch-owned device buffers; gloo stages them through the host for the collective only)
        from flashweave_jl_amd.dist import make_dev_exchange, sharded_cor
        xs = {}
        eng.level0_dev(rank, world, make_dev_exchange(dist, torch.device("cuda", 0), stats=xs))
        nb = eng.pw_univar_neighbors_get()
        res["%s_l0_dev" % kind] = [nb["off"].tolist(), nb["idx"].tolist(), nb["stat"].tolist(), nb["pval"].tolist()]
        res["%s_l0_dev_records" % kind] = xs.get("level0_records", 0)
        # ... and the per-round exchange of the conditional stage the same way (fw_learn_network_dev)
        net = eng.lgl(feed_forward=True, round_size=32, rank=rank, world_size=world,
                      dev_exchange=make_dev_exchange(dist, torch.device("cuda", 0)))
        res["%s_ff1_devx" % kind] = sorted([a, b, w] for (a, b), w in net["edges"].items())
        if kind == "fz":
            # row-block sharding of cor(): each rank computes half of the rows, the blocks are gathered in place inside a
            # torch tensor the engine uses as its matrix; bit-identical to the single-rank matrix, and so is the network on it
            cm1 = eng.cor()
            e2 = fw.Engine(kind, n, p, max_k=3)
            e2.set_data(data)
            buf = sharded_cor(e2, dist, torch.device("cuda", 0), rank, world)
            cm2 = e2.cor_mat()
            res["fz_cor_sharded_equal"] = bool(np.array_equal(cm1, cm2, equal_nan=True))
            net = e2.lgl(feed_forward=True, round_size=32, rank=rank, world_size=world, allgather=cb)
            res["fz_ff1_sharded_cor"] = sorted([a, b, w] for (a, b), w in net["edges"].items())
            e2.close()
            del buf
        eng.close()
    json.dump(res, open(out_path + ".%d" % rank, "w"))
    dist.barrier()
    dist.destroy_process_group()
    return True


if __name__ == "__main__":
    mode, rank, world = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    if mode == "callback":
        ok = run_callback_only(rank, world)
    else:
        ok = run_sharded_gpu(rank, world, sys.argv[4])
    sys.exit(0 if ok else 1)
