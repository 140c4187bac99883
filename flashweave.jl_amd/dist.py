"""Multi-GPU exchange step of the target-sharded conditional stage.

The path shards by target variable: the targets of a feed-forward round are dealt to the ranks longest-estimated-work
first (LPT on (level-0 degree)^min(max_k, 3) + 64, computed identically on every rank: fw_hiton.cpp), every rank holds
the full (replicated) packed data / correlation matrix, so the data path needs no collective.  The one real
exchange is the per-round all-gather of the newly found directed neighbour entries (target, neighbour, stat, p)
that feeds the next round's whitelists (the role of the master's running graph, reference
src/interleaved.jl:124-140,166-183).  `make_allgather` builds the callback fw_learn_network expects
(include/flashweave_amd.h: fw_allgather_fn) on top of torch.distributed -- backend "nccl" (= RCCL over xGMI) in
bench.py, "gloo" in the CPU tests.

Wire format: ONE all_gather per round of a fixed-capacity int64 message per rank,
    [count | cap x (target | neighbour << 32, bits(stat), bits(p))]
-- integers stay integers and the two Float64 travel as their bit patterns (NaN payloads and subnormal p-values
survive).  The message is packed in a pinned host buffer, copied to the device once, gathered over RCCL and copied
back once; if any rank's count exceeds the capacity every rank sees that in the gathered headers, doubles the
capacity to the same value and repeats the round (a first-use event, not a steady-state one).
"""
import ctypes as C
import time
import traceback

import numpy as np


def make_allgather(dist, device, capacity=4096, stats=None):
    """-> python callable with the fw_allgather_fn signature (wrap with engine.ALLGATHER_FN or pass to Engine.lgl).
    stats: optional dict that accumulates {"calls", "collectives", "seconds", "entries"} (bench.py reports them)."""
    import torch
    on_gpu = device.type == "cuda"
    st = {"cap": 0, "send": None, "send_dev": None, "recv_dev": None, "recv": None}
    keep = {}

    def ensure(cap, world):
        if st["cap"] >= cap and st.get("world") == world:
            return
        st["world"] = world
        words = 1 + 3 * cap
        st["cap"] = cap
        st["send"] = torch.zeros(words, dtype=torch.int64, pin_memory=on_gpu)
        st["recv"] = torch.zeros(world * words, dtype=torch.int64, pin_memory=on_gpu)  # flat: gloo insists
        if on_gpu:
            st["send_dev"] = torch.zeros(words, dtype=torch.int64, device=device)
            st["recv_dev"] = torch.zeros(world * words, dtype=torch.int64, device=device)

    def cb(user, n_local, tgt, nbr, stat, pval, n_total, tgt_all, nbr_all, stat_all, pval_all):
        try:
            t0 = time.perf_counter()
            n = int(n_local)
            world = dist.get_world_size()
            cap = max(st["cap"], capacity)
            ncoll = 0
            while True:
                ensure(cap, world)
                cap = st["cap"]
                send = st["send"].numpy()
                send[0] = n
                m = min(n, cap)
                if m:
                    rows = send[1:1 + 3 * m].reshape(m, 3)
                    t = np.ctypeslib.as_array(tgt, shape=(n,))[:m].astype(np.int64)
                    u = np.ctypeslib.as_array(nbr, shape=(n,))[:m].astype(np.int64)
                    rows[:, 0] = (t & 0xFFFFFFFF) | (u << 32)  # low half masked: a negative marker (level-0 sharding sends target -1) must not sign-extend into the neighbour half
                    rows[:, 1] = np.ctypeslib.as_array(stat, shape=(n,))[:m].view(np.int64)
                    rows[:, 2] = np.ctypeslib.as_array(pval, shape=(n,))[:m].view(np.int64)
                if on_gpu:
                    st["send_dev"].copy_(st["send"], non_blocking=True)
                    dist.all_gather_into_tensor(st["recv_dev"], st["send_dev"])
                    st["recv"].copy_(st["recv_dev"], non_blocking=True)
                    torch.cuda.current_stream(device).synchronize()
                else:
                    dist.all_gather_into_tensor(st["recv"], st["send"])
                ncoll += 1
                recv = st["recv"].numpy().reshape(world, 1 + 3 * cap)
                counts = recv[:, 0].copy()
                mx = int(counts.max())
                if mx <= cap:
                    break
                cap = 1 << (mx - 1).bit_length()  # every rank computes the same new capacity from the same headers
            N = int(counts.sum())
            if N:
                rows = np.concatenate([recv[r, 1:1 + 3 * int(counts[r])].reshape(-1, 3) for r in range(world)], axis=0)
                keep["t"] = np.ascontiguousarray((rows[:, 0] & 0xFFFFFFFF).astype(np.uint32).view(np.int32))  # through uint32: -1 comes back as -1
                keep["n"] = np.ascontiguousarray((rows[:, 0] >> 32).astype(np.int32))
                keep["s"] = np.ascontiguousarray(rows[:, 1]).view(np.float64)
                keep["p"] = np.ascontiguousarray(rows[:, 2]).view(np.float64)
            else:
                keep["t"] = keep["n"] = np.zeros(1, np.int32)
                keep["s"] = keep["p"] = np.zeros(1, np.float64)
            n_total[0] = N
            tgt_all[0] = keep["t"].ctypes.data_as(C.POINTER(C.c_int32))
            nbr_all[0] = keep["n"].ctypes.data_as(C.POINTER(C.c_int32))
            stat_all[0] = keep["s"].ctypes.data_as(C.POINTER(C.c_double))
            pval_all[0] = keep["p"].ctypes.data_as(C.POINTER(C.c_double))
            if stats is not None:
                stats["calls"] = stats.get("calls", 0) + 1
                stats["collectives"] = stats.get("collectives", 0) + ncoll
                stats["entries"] = stats.get("entries", 0) + N
                stats["seconds"] = stats.get("seconds", 0.0) + (time.perf_counter() - t0)
            return 0
        except Exception:  # never let an exception cross the C boundary
            traceback.print_exc()
            return 1

    return cb


def make_dev_exchange(dist, device, stats=None):
    """-> (prepare, exchange): the two callbacks of fw_dev_exchange (include/flashweave_amd.h) on torch.distributed.  The send /
    receive buffers are torch tensors in DEVICE memory; the library packs into and unpacks from them with its own kernels, and the
    all-gather runs on them directly (backend nccl = RCCL over xGMI).  With a CPU backend (gloo: the world_size-2 tests, two ranks
    sharing one GPU) the same tensors are staged through the host for the collective only."""
    import torch
    st = {"send": None, "recv": None, "cap": 0, "rec": 0, "world": 0}
    on_gpu_coll = dist.get_backend() == "nccl"

    def prepare(user, n_local, aux_local, rec_bytes, d_send, d_recv, counts, aux, cap_records):
        try:
            st["t0"] = time.perf_counter()
            world = dist.get_world_size()
            hdr = torch.tensor([int(n_local), int(aux_local)], dtype=torch.int64, device=device if on_gpu_coll else "cpu")
            allh = torch.empty(2 * world, dtype=torch.int64, device=hdr.device)
            dist.all_gather_into_tensor(allh, hdr)
            allh = allh.cpu().view(world, 2)
            cap = max(int(allh[:, 0].max()), 1)
            if st["cap"] < cap or st["rec"] != rec_bytes or st["world"] != world:
                cap2 = 1 << (cap - 1).bit_length()
                st["send"] = torch.empty(cap2 * rec_bytes, dtype=torch.uint8, device=device)
                st["recv"] = torch.empty(world * cap2 * rec_bytes, dtype=torch.uint8, device=device)
                st["cap"], st["rec"], st["world"] = cap2, rec_bytes, world
            for r in range(world):
                counts[r] = int(allh[r, 0])
                aux[r] = int(allh[r, 1])
            d_send[0] = st["send"].data_ptr()
            d_recv[0] = st["recv"].data_ptr()
            cap_records[0] = st["cap"]
            st["n_all"] = int(allh[:, 0].sum())
            if stats is not None:
                stats["level0_records"] = st["n_all"]
            return 0
        except Exception:
            traceback.print_exc()
            return 1

    def exchange(user):
        try:
            t0 = time.perf_counter()
            if on_gpu_coll:
                dist.all_gather_into_tensor(st["recv"], st["send"])
                torch.cuda.current_stream(device).synchronize()
            else:
                h = torch.empty(st["recv"].numel(), dtype=torch.uint8)
                dist.all_gather_into_tensor(h, st["send"].cpu())
                st["recv"].copy_(h)
                torch.cuda.synchronize(device)
            if stats is not None:
                stats["level0_exchange_s"] = stats.get("level0_exchange_s", 0.0) + time.perf_counter() - t0
                stats["level0_exchange_bytes"] = st["recv"].numel()
                # the same counters make_allgather keeps (bench.py reports them per step): one call = header + payload collectives
                stats["calls"] = stats.get("calls", 0) + 1
                stats["collectives"] = stats.get("collectives", 0) + 2
                stats["entries"] = stats.get("entries", 0) + st.get("n_all", 0)
                stats["seconds"] = stats.get("seconds", 0.0) + (time.perf_counter() - st.get("t0", t0))
            return 0
        except Exception:
            traceback.print_exc()
            return 1

    return prepare, exchange


def sharded_cor(eng, dist, device, rank, world, keep=None):
    """cor(data_dense) with the row blocks dealt to the ranks: every rank computes rows_per_rank rows on its MFMA units, the
    blocks are all-gathered IN PLACE inside one torch tensor that the engine uses as its matrix (fw_use_cor_buffer), so every
    rank ends with the whole matrix resident.  -> the tensor (keep a reference as long as the engine lives)."""
    import torch
    p = eng.p
    T = (p + 127) // 128
    rpr = 128 * ((T + world - 1) // world)
    buf = keep if keep is not None and keep.numel() >= world * rpr * p else torch.empty(world * rpr * p, dtype=torch.float32, device=device)
    eng.use_cor_buffer(buf.data_ptr(), buf.numel())
    row0, rows = eng.compute_cor_rows(rank, world)
    assert rows == rpr and row0 == rank * rpr
    mine = buf[rank * rpr * p:(rank + 1) * rpr * p]
    if dist.get_backend() == "nccl":
        try:  # in place: this rank's block already sits at offset rank * block inside the output (what RCCL's in-place form expects)
            dist.all_gather_into_tensor(buf[:world * rpr * p], mine)
        except RuntimeError:  # a backend build that refuses aliased input / output: gather through a second buffer
            tmp = torch.empty(world * rpr * p, dtype=torch.float32, device=device)
            dist.all_gather_into_tensor(tmp, mine.clone())
            buf[:world * rpr * p].copy_(tmp)
            del tmp
        torch.cuda.current_stream(device).synchronize()
    else:
        h = torch.empty(world * rpr * p, dtype=torch.float32)
        dist.all_gather_into_tensor(h, mine.cpu())
        buf[:world * rpr * p].copy_(h)
        torch.cuda.synchronize(device)
    eng.cor_ready()
    return buf
