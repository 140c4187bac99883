"""Multi-GPU exchange step of the target-sharded conditional stage.

The path shards by target variable: every rank owns every world_size-th target of a feed-forward round and holds
the full (replicated) packed data / correlation matrix, so the data path needs no collective.  The one real
exchange is the per-round all-gather of the newly found directed neighbour entries (target, neighbour, stat, p)
that feeds the next round's whitelists (the role of the master's running graph, reference
src/interleaved.jl:124-140,166-183).  `make_allgather` builds the callback fw_learn_network expects
(include/flashweave_amd.h: fw_allgather_fn) on top of torch.distributed -- backend "nccl" (= RCCL over xGMI) in
bench.py, "gloo" in the CPU tests.  Payloads are KBs-MBs: latency-bound, one padded all_gather per round.
"""
import ctypes as C
import traceback

import numpy as np


def make_allgather(dist, device):
    """-> python callable with the fw_allgather_fn signature (wrap with engine.ALLGATHER_FN or pass to Engine.lgl)."""
    import torch
    keep = {}

    def cb(user, n_local, tgt, nbr, stat, pval, n_total, tgt_all, nbr_all, stat_all, pval_all):
        try:
            n = int(n_local)
            world = dist.get_world_size()
            local = np.zeros((n, 4), dtype=np.float64)
            if n:
                local[:, 0] = np.ctypeslib.as_array(tgt, shape=(n,))
                local[:, 1] = np.ctypeslib.as_array(nbr, shape=(n,))
                local[:, 2] = np.ctypeslib.as_array(stat, shape=(n,))
                local[:, 3] = np.ctypeslib.as_array(pval, shape=(n,))
            sizes = torch.zeros(world, dtype=torch.int64, device=device)
            mine = torch.tensor([n], dtype=torch.int64, device=device)
            dist.all_gather_into_tensor(sizes, mine)
            sizes = sizes.cpu().numpy()
            mx = int(sizes.max())
            out = np.zeros((0, 4))
            if mx > 0:
                pad = torch.zeros((mx, 4), dtype=torch.float64, device=device)
                if n:
                    pad[:n] = torch.from_numpy(local).to(device)
                allt = torch.zeros((world * mx, 4), dtype=torch.float64, device=device)
                dist.all_gather_into_tensor(allt, pad)
                allt = allt.cpu().numpy().reshape(world, mx, 4)
                out = np.concatenate([allt[r, :int(sizes[r])] for r in range(world)], axis=0)
            N = out.shape[0]
            keep["t"] = np.ascontiguousarray(out[:, 0].astype(np.int32)) if N else np.zeros(1, np.int32)
            keep["n"] = np.ascontiguousarray(out[:, 1].astype(np.int32)) if N else np.zeros(1, np.int32)
            keep["s"] = np.ascontiguousarray(out[:, 2]) if N else np.zeros(1, np.float64)
            keep["p"] = np.ascontiguousarray(out[:, 3]) if N else np.zeros(1, np.float64)
            n_total[0] = N
            tgt_all[0] = keep["t"].ctypes.data_as(C.POINTER(C.c_int32))
            nbr_all[0] = keep["n"].ctypes.data_as(C.POINTER(C.c_int32))
            stat_all[0] = keep["s"].ctypes.data_as(C.POINTER(C.c_double))
            pval_all[0] = keep["p"].ctypes.data_as(C.POINTER(C.c_double))
            return 0
        except Exception:  # never let an exception cross the C boundary
            traceback.print_exc()
            return 1

    return cb
