#!/usr/bin/env python
"""bench.py -- headline benchmark of the MI355X-native FlashWeave CI-test engine.

Metric (BASELINE.json): CI tests/sec + time-to-network on synthetic 10k OTUs x 2k samples, FlashWeave-S
(Fisher-z), max_k = 3 ("cfg3", fwsynth-v1).  One STEP = one full pass of the hot path over
that input with the normalised matrix already resident in HBM: level-0 Pearson matrix (MFMA) + all-pairs tests +
BH, then the conditional HITON-PC stage (level-synchronous test_subsets batches) up to the edge list on the host.
CI tests counted = p(p-1)/2 level-0 tests + the reference-equivalent number of conditional tests (the count the
sequential reference order would execute; identical to the CPU oracle's by construction).

    python bench.py --gpus N --steps K --warmup W

N > 1: this script starts N ranks ITSELF (re-executes under `python -m torch.distributed.run --nproc-per-node N`,
rendezvous on 127.0.0.1), one rank per GPU, backend nccl (= RCCL over xGMI); when it is already running under
torch.distributed.run (WORLD_SIZE in the environment) it uses those ranks and insists that WORLD_SIZE == N.
Targets of every feed-forward round are dealt over the ranks by estimated work (heaviest first, least loaded rank) and the per-round neighbour sets are
all-gathered (flashweave.jl_amd/dist.py).  Total work is fixed as N grows -> "strong".  It fails loudly if fewer
than N GPUs are visible.

Two schedules are measured in every run (both in the ONE JSON line rank 0 prints):
  * headline (`value`, `ms_per_step`): --feed-forward / --round-size as given (defaults below);
  * `other_schedule`: the other one of {feed_forward = 1 in rounds of R targets, feed_forward = 0}.
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)
DEFAULT_ROUND = 1024    # granularity of the default feed-forward round size: R = 1024 * ceil(p / 10240), i.e. ~10 rounds per pass
                        # (cfg3: R = 1024; cfg4: R = 5120); results do not depend on the number of GPUs, only on R


def make_input(cfg, args):
    """-> (config, checksum of the counts, normalised matrix, normalisation record).  The count table is normalised by the
    device front-end (fw_normalize_counts: clr_adapt / clr_nz / binary / binned_nz_clr; --host-normalize: preprocess.py);
    its seconds are reported separately and are never part of `value`."""
    import flashweave_jl_amd as fw
    from flashweave_jl_amd import preprocess as pre
    from flashweave_jl_amd import synth
    if cfg == "cfg1":
        # BASELINE.json configs[0]: the reference's bundled OTU table (tests/golden/HMP_SRA_gut_small.tsv = test/data of the reference),
        # FlashWeave-S, max_k 3 -- a parity case (tests/test_gpu_fz.py reproduces its golden networks), runnable here for completeness
        path = os.path.join(ROOT, "tests", "golden", "HMP_SRA_gut_small.tsv")
        counts = np.loadtxt(path, delimiter="\t", skiprows=1, usecols=range(1, 51)).astype(np.int64)
        c = {"p": counts.shape[1], "n": counts.shape[0], "seed": 0, "mode": "bundled", "test_name": "fz", "max_k": 3,
             "label": "cfg1: bundled HMP_SRA_gut_small table (%d samples x %d OTUs), fz, max_k=3, alpha=0.01" % counts.shape}
        t1 = time.perf_counter()
        data, _, _ = pre.normalize(counts, "fz", prec=32)
        return c, synth.checksum(counts.astype(np.int32)), data, {"seconds": time.perf_counter() - t1, "where": "host (preprocess.py)", "generate_counts_seconds": 0.0}
    c = dict(synth.CONFIGS[cfg])
    if args.p:
        c["p"] = args.p
    if args.n:
        c["n"] = args.n
    dev = (lambda cnt, t: fw.normalize_counts(cnt, t, device=int(os.environ.get("LOCAL_RANK", "0")) if not args.single_device else 0)) \
        if not args.host_normalize else None
    t0 = time.perf_counter()
    if c.get("habitats"):  # HE configs: structural absences + 20 binary meta variables
        counts, meta = synth.generate(c["p"], c["n"], c["seed"], mode=c["mode"], habitats=c["habitats"], n_meta=c["n_meta"])
        t1 = time.perf_counter()
        # the front-end's meta-variable path (preprocess_data with a meta_mask, preprocessing.jl:412-563): row filters, one-hot /
        # discretisation where needed, zero-variance meta columns dropped, appended behind the OTUs
        data = np.ascontiguousarray(pre.normalize_with_meta(counts, c["test_name"], meta.astype(np.float64), prec=32, normalizer=dev)["data"])
    else:
        counts = synth.generate(c["p"], c["n"], c["seed"], mode=c["mode"])
        t1 = time.perf_counter()
        data, _, _ = dev(counts, c["test_name"]) if dev else pre.normalize(counts, c["test_name"], prec=32)
    t2 = time.perf_counter()
    norm = {"seconds": t2 - t1, "where": "host (preprocess.py)" if args.host_normalize else "device (fw_normalize_counts, includes the H2D upload "
            "of the Int32 counts and the D2H copy of the result)", "generate_counts_seconds": t1 - t0}
    return c, synth.checksum(counts), data, norm


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="cfg3")
    ap.add_argument("--p", type=int, default=0, help="override #OTUs (debug)")
    ap.add_argument("--n", type=int, default=0, help="override #samples (debug)")
    ap.add_argument("--feed-forward", type=int, default=1, help="reference default: true (learning.jl:469)")
    ap.add_argument("--round-size", type=int, default=-1,
                    help="targets per feed-forward round (-1: 1024 * ceil(p / 10240), about ten rounds; 0 = one round = feed_forward off)")
    ap.add_argument("--replicate-level0", action="store_true",
                    help="N > 1, discrete kinds: every rank screens ALL level-0 pair tiles (r02 behaviour); default: 1/N of them, "
                         "significant pairs all-gathered in device memory (fw_level0_sharded_dev)")
    ap.add_argument("--host-exchange", action="store_true",
                    help="N > 1: per-round exchange through the host callback (numpy packing, r02 form) instead of fw_learn_network_dev")
    ap.add_argument("--library-rccl", action="store_true",
                    help="N > 1 (or --force-dist): the exchanges are issued by the LIBRARY on its own RCCL communicator (fw_comm_init, "
                         "fw_level0_comm, fw_cor_mat_allgather_comm, fw_learn_network_comm); torch.distributed only carries the 128-byte "
                         "rendezvous id.  Default: torch.distributed collectives on device buffers the library packs (fw_dev_exchange)")
    ap.add_argument("--shard-cor", action="store_true",
                    help="N > 1, fz: row-block sharding of the Pearson GEMM with an in-place all-gather (default from p = 30 000 on)")
    ap.add_argument("--max-targets", type=int, default=0,
                    help="conditional stage of the first M targets of the schedule only (a bounded SAMPLE of configs whose full "
                         "conditional stage takes hours, e.g. cfg5; the JSON line says so and is not a whole-network result)")
    ap.add_argument("--stream-columns", action="store_true",
                    help="fz: recursive_pcor = 0 -- no correlation matrix for the conditional tests, every test streams its sample columns "
                         "(statfuns.jl:19-21; fw_fzs.hip, host job pool); bound the run with --max-targets")
    ap.add_argument("--no-cor-matrix", action="store_true",
                    help="fz: dense_cor = false (fw_params.no_cor_mat) on top of --stream-columns: no p x p matrix at any time, level 0 "
                         "multiplies and screens the centred columns tile by tile (learning.jl:42, tests.jl:118-147)")
    ap.add_argument("--no-other-schedule", action="store_true", help="skip the second (other_schedule) measurement")
    ap.add_argument("--no-one-chain", action="store_true", help="skip the one-chain pass the per-kernel roofline figures come from")
    ap.add_argument("--host-seam", action="store_true",
                    help="also time the pass with FW_HOST_HITON=1: the host job pool over fw_test_subsets_batch-style "
                         "launches, the seam a Julia host would call (hiton.jl:100)")
    ap.add_argument("--check-determinism", action="store_true", help="compare every timed pass with the first (network bytes, reference-order test count) and report on stderr; not for a measured line")
    ap.add_argument("--host-normalize", action="store_true", help="normalise the count table with the host front-end (preprocess.py) instead of the device one")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--cpu-workers", type=int, default=-1,
                    help="threads of the multi-core CPU leg (-1: all hardware threads; 0: skip it)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for debugging)")
    ap.add_argument("--single-device", action="store_true", help="debug: every rank uses GPU 0 (needs --backend gloo)")
    ap.add_argument("--force-dist", action="store_true",
                    help="debug: go through the launcher + process group + exchange callback even for --gpus 1")
    ap.add_argument("--spawn-check", action="store_true",
                    help="start the ranks, initialise the process group, all-reduce a 1 per rank and print the count "
                         "(no engine; used by the CPU test of the launcher)")
    ap.add_argument("--simulate-world", type=int, default=0,
                    help="debug: time rank 0's share of an N-rank job on one GPU (no peers; the JSON line is not a result)")
    ap.add_argument("--simulate-rank", type=int, default=0,
                    help="debug: with --simulate-world N, which rank's share to time (-1: every rank in turn; the line then "
                         "reports the slowest rank, i.e. what the N-rank job would take without its exchange)")
    return ap.parse_args(argv)


def launch(args, argv):
    """Parent process of `python bench.py --gpus N`: start N ranks under torch.distributed.run and relay their output."""
    n = args.gpus
    if not args.spawn_check and not args.single_device:
        import torch
        have = torch.cuda.device_count()
        if have < n:
            sys.stderr.write("bench.py: --gpus %d but only %d GPU(s) visible -- refusing to run (no silent fallback to "
                             "fewer ranks)\n" % (n, have))
            sys.exit(2)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    sys.stderr.write("bench.py: starting %d ranks: %s\n" % (n, " ".join(cmd)))
    sys.exit(subprocess.call(cmd, env=env))


def cpu_baseline(args, cfg, data, n, p, eng, level0_per_step):
    """The oracle (C restatement, kind "port") timed on this box's host cores on a bounded sample of the workload."""
    import threading

    from oracle import oracle as O
    kind, max_k = cfg["test_name"], cfg["max_k"]
    t_all = time.perf_counter()
    shared = {}
    if kind == "fz":
        shared["cm"] = eng.cor_mat()

        def mk():
            return O.Oracle("fz", cor_mat=shared["cm"], n_obs=n)
    elif kind == "fz_nz":
        def mk():
            return O.Oracle("fz_nz", data)
    else:
        shared["csc"] = O.dense_to_csc(data)

        def mk():
            return O.Oracle(kind, csc=shared["csc"], shape=(n, p), sparse=True, max_k=max_k)
    orc = mk()
    nom = orc.auto_n_obs_min(-1, 5, max_k)
    budget = args.cpu_seconds
    nt = args.max_targets or p  # schedule positions the device run covered
    tstride = max(1, nt // 512)
    # level 0: the full pass when it fits the budget (5e7 pairs at cfg3: a few seconds), else every s-th row
    if level0_per_step <= 100_000_000:
        t1 = time.perf_counter()
        full = orc.level0(alpha=0.01, hps=5, n_obs_min=nom)
        l0_tests, l0_secs = full["n_tests"], time.perf_counter() - t1
        nb = dict(off=full["off"], idx=full["idx"], stat=full["stat"], pval=full["pval"], n_tests=l0_tests)
        r = orc.learn(max_k=max_k, feed_forward=False, target_stride=tstride, max_seconds=budget, nbrs=nb, max_targets=args.max_targets)
        l0_note = "full level-0 (%d pair tests, %.2fs)" % (l0_tests, l0_secs)
    else:
        stride = max(1, p // 64)
        l0_tests, l0_secs = orc.level0_sample(hps=5, n_obs_min=nom, x_start=0, x_stride=stride, max_seconds=budget)
        nb = eng.pw_univar_neighbors()  # the conditional stage of the sampled targets needs the neighbour lists: the
        nb["n_tests"] = level0_per_step  # device's (bit-identical to the oracle's, tests/test_gpu_fullsize.py)
        r = orc.learn(max_k=max_k, feed_forward=False, target_stride=tstride, max_seconds=budget, nbrs=nb, max_targets=args.max_targets)
        l0_note = "level-0 rows 0, %d, 2*%d, ... against every later column (%d pair tests, %.2fs; neighbour lists for " \
                  "the conditional sample taken from the device run)" % (stride, stride, l0_tests, l0_secs)
    l0_rate = l0_tests / max(l0_secs, 1e-9)
    c_rate = r["n_cond_tests"] / max(r["t_cond"], 1e-9)
    cpu = {"unit": "tests/s", "cores": 1, "kind": "port", "feed_forward": 0,
           "schedule_note": "the CPU sample runs feed_forward = 0 (one round, no whitelists); the GPU headline runs the schedule in "
                            "config.feed_forward / config.round_size -- rates per test are comparable, the conditioning pools differ",
           "sample": "oracle/fw_oracle.c (C restatement; the Julia reference cannot run here): %s + conditional stage of "
                     "every %d-th target of the schedule (%d targets, %d tests, %.2fs), feed_forward=0" %
                     (l0_note, tstride, r["n_targets"], r["n_cond_tests"], r["t_cond"]),
           "level0_tests_per_s": l0_rate, "cond_tests_per_s": c_rate,
           "value": (l0_tests + r["n_cond_tests"]) / max(l0_secs + r["t_cond"], 1e-9)}
    # the same oracle on every hardware thread: T threads, one oracle context each over the shared read-only inputs,
    # thread w takes schedule positions w, w + S, ... (the analogue of the reference's worker processes,
    # interleaved.jl:90); level-0 stays single-threaded as in the reference (tests.jl:470-479), so only the
    # conditional stage is reported for this leg
    T = args.cpu_workers if args.cpu_workers >= 0 else (os.cpu_count() or 1)
    if T > 1:
        S = max(T, tstride, 1)
        outs = [None] * T
        mbudget = max(3.0, budget * 0.6)

        def work(w):
            o = mk()
            outs[w] = o.learn(max_k=max_k, feed_forward=False, target_stride=S, target_offset=w, max_seconds=mbudget, nbrs=nb,
                              max_targets=args.max_targets)
            o.close()

        t2 = time.perf_counter()
        th = [threading.Thread(target=work, args=(w,)) for w in range(T)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        outs = [o for o in outs if o is not None]
        tot = sum(o["n_cond_tests"] for o in outs)
        tmax = max(o["t_cond"] for o in outs)
        cpu["multicore"] = {"cores": T, "hardware_threads": os.cpu_count(), "cond_tests_per_s": tot / max(tmax, 1e-9),
                            "sample": "%d oracle threads (one context each, shared inputs), thread w = schedule positions "
                                      "w, w + %d, ... (%d targets, %d conditional tests, slowest thread %.2fs)" %
                                      (T, S, sum(o["n_targets"] for o in outs), tot, tmax),
                            "wall_s": time.perf_counter() - t2}
    cpu["wall_s"] = time.perf_counter() - t_all
    orc.close()
    return cpu


def main():
    argv = sys.argv[1:]
    args = parse_args(argv)
    in_dist = "WORLD_SIZE" in os.environ
    if not in_dist and (args.gpus > 1 or args.force_dist):
        launch(args, argv)  # does not return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if in_dist and world != args.gpus:
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%d -- refusing to run\n" % (args.gpus, world))
        sys.exit(2)

    import torch
    import torch.distributed as dist
    if args.spawn_check:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(args.backend if args.backend != "nccl" or torch.cuda.is_available() else "gloo")
        one = torch.ones(1, dtype=torch.int64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(one)
        if rank == 0:
            print(json.dumps({"spawn_check": True, "n_ranks": int(one.item()), "world_size": dist.get_world_size(),
                              "backend": dist.get_backend()}))
        dist.destroy_process_group()
        return

    if args.single_device:
        local_rank = 0
    if not args.single_device and torch.cuda.device_count() < max(world, 1):
        sys.stderr.write("bench.py: %d ranks but %d GPU(s) visible -- refusing to run\n" % (world, torch.cuda.device_count()))
        sys.exit(2)
    use_dist = in_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend)
        if rank == 0:
            sys.stderr.write("bench.py: process group up: backend=%s world_size=%d\n" % (dist.get_backend(), dist.get_world_size()))
    dev = torch.device("cuda", local_rank)
    cdev = dev if args.backend == "nccl" else torch.device("cpu")  # where collective payloads live

    import flashweave_jl_amd as fw
    from flashweave_jl_amd.dist import make_allgather

    cfg, csum, data, norm_rec = make_input(args.config, args)
    n, p = data.shape
    if args.no_cor_matrix:
        args.stream_columns = True
    eng = fw.Engine(cfg["test_name"], n, p, max_k=cfg["max_k"], device=local_rank, recursive_pcor=not args.stream_columns,
                    dense_cor=not args.no_cor_matrix)
    eng.set_data(data)  # host -> HBM once, outside the timed region
    xstats = {}
    cb = make_allgather(dist, cdev, stats=xstats) if use_dist else None
    sim = {"mode": None, "rounds": [], "k": 0, "local": 0}
    if args.simulate_world > 1 and world == 1:
        # One GPU standing in for one rank of an N-rank job.  The learned network does not depend on N (only on the round
        # size), so the union over the ranks of what round k finds IS what a single-rank run finds in round k: a first pass
        # with world_size = 1 RECORDS every round's exchange, and the pass of rank r then gets the recorded round back from
        # its all-gather -- exactly the whitelists it would see among N real ranks, without peers.
        import ctypes as C

        def cb(user, n_local, tgt, nbr, stat, pval, n_total, tgt_all, nbr_all, stat_all, pval_all):
            n = int(n_local)
            if sim["mode"] == "record":
                sim["rounds"].append(tuple(np.ctypeslib.as_array(a, shape=(max(n, 1),))[:n].copy() for a in (tgt, nbr, stat, pval)))
                n_total[0] = n_local
                tgt_all[0], nbr_all[0], stat_all[0], pval_all[0] = tgt, nbr, stat, pval
                return 0
            t, u, s_, q = sim["rounds"][sim["k"]]
            sim["k"] += 1
            sim["local"] += n
            n_total[0] = len(t)
            tgt_all[0] = t.ctypes.data_as(C.POINTER(C.c_int32))
            nbr_all[0] = u.ctypes.data_as(C.POINTER(C.c_int32))
            stat_all[0] = s_.ctypes.data_as(C.POINTER(C.c_double))
            pval_all[0] = q.ctypes.data_as(C.POINTER(C.c_double))
            return 0

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    discrete = cfg["test_name"] in ("mi", "mi_nz")
    # level 0 of an N-rank job.  Discrete kinds: every rank screens 1/N of the pair tiles and the significant pairs are all-gathered
    # in device memory (fw_level0_sharded_dev; --replicate-level0 switches it off).  Fisher-z: the GEMM is row-block sharded with an
    # in-place all-gather of the matrix from p = 30 000 on (--shard-cor forces it): below that the replicated GEMM (cfg3: 2 ms) is
    # cheaper than moving 400 MB.
    shard_l0 = discrete and not args.replicate_level0
    shard_cor = cfg["test_name"] == "fz" and (args.shard_cor or p >= 30000)
    xround = None
    if use_dist and not args.host_exchange:
        from flashweave_jl_amd.dist import make_dev_exchange
        xround = make_dev_exchange(dist, dev, stats=xstats)
    xdev = None
    if use_dist and shard_l0:
        from flashweave_jl_amd.dist import make_dev_exchange
        xdev = make_dev_exchange(dist, dev, stats=xstats)
    l0sim = {"mode": None, "all": None, "counts": None, "aux": None, "send": None, "cap": 0, "rank": 0}
    if args.simulate_world > 1 and world == 1 and shard_l0:
        # one GPU standing in for one rank: pass A lets every rank screen its tiles once and keeps its packed records; pass B
        # (timed) hands every rank the complete gathered buffer, as the collective among N real ranks would
        NW = args.simulate_world

        def l0_prepare(user, n_local, aux_local, rec_bytes, d_send, d_recv, counts, aux, cap_records):
            r = l0sim["rank"]
            if l0sim["mode"] == "collect":
                cap = max(int(n_local), 1)
                if l0sim["cap"] < cap:
                    capn = max(1 << (cap - 1).bit_length(), 2 * l0sim["cap"])
                    old, oc = l0sim["all"], l0sim["cap"]
                    l0sim["all"] = torch.zeros(NW * capn * rec_bytes, dtype=torch.uint8, device=dev)
                    if old is not None:
                        for q in range(NW):
                            l0sim["all"][q * capn * rec_bytes:q * capn * rec_bytes + oc * rec_bytes] = old[q * oc * rec_bytes:(q + 1) * oc * rec_bytes]
                    l0sim["send"] = torch.zeros(capn * rec_bytes, dtype=torch.uint8, device=dev)
                    l0sim["cap"] = capn
                l0sim["counts"][r], l0sim["aux"][r], l0sim["rec"] = int(n_local), int(aux_local), rec_bytes
            for q in range(NW):
                counts[q] = l0sim["counts"][q]
                aux[q] = l0sim["aux"][q]
            d_send[0] = l0sim["send"].data_ptr()
            d_recv[0] = l0sim["all"].data_ptr()
            cap_records[0] = l0sim["cap"]
            return 0

        def l0_exchange(user):
            if l0sim["mode"] == "collect":
                r, cb_, rb = l0sim["rank"], l0sim["cap"], l0sim["rec"]
                n = l0sim["counts"][r]
                l0sim["all"][r * cb_ * rb:r * cb_ * rb + n * rb] = l0sim["send"][:n * rb]
                torch.cuda.synchronize()
            return 0

        def l0_collect():
            npairs = p * (p - 1) // 2
            l0sim.update(mode="collect", counts=[0] * NW, aux=[npairs] * NW)
            for r in range(NW):
                l0sim["rank"] = r
                eng.level0_dev(r, NW, (l0_prepare, l0_exchange))
            l0sim["mode"] = "replay"

    lib_comm = bool(use_dist and args.library_rccl)
    if lib_comm:
        if dist.get_backend() != "nccl":
            raise SystemExit("bench.py --library-rccl needs the nccl (RCCL) backend: one rank per GPU")
        box = [fw.Engine.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        eng.comm_init(box[0], rank, world)

    def step_lib(ff, R):
        # every exchange inside the library (fw_rccl.cpp)
        if cfg["test_name"] == "fz" and not args.no_cor_matrix:
            if shard_cor:
                T_ = (p + 127) // 128
                rpr = 128 * ((T_ + world - 1) // world)
                if sim.get("corbuf") is None or sim["corbuf"].numel() < world * rpr * p:
                    sim["corbuf"] = torch.empty(world * rpr * p, dtype=torch.float32, device=dev)
                eng.use_cor_buffer(sim["corbuf"].data_ptr(), sim["corbuf"].numel())
                eng.compute_cor_rows(rank, world)
                eng.cor_allgather_comm(rpr)
                eng.cor_ready()
            else:
                eng.compute_cor()
        if shard_l0:
            eng.level0_comm()
        else:
            eng.level0()
        return eng.lgl_comm(feed_forward=bool(ff), round_size=R, max_targets=args.max_targets, edge_dict=False)

    def step(ff, R):
        if lib_comm:
            return step_lib(ff, R)
        if cfg["test_name"] == "fz" and not args.no_cor_matrix:
            if use_dist and shard_cor:
                from flashweave_jl_amd.dist import sharded_cor
                sim["corbuf"] = sharded_cor(eng, dist, dev, rank, world, keep=sim.get("corbuf"))
            else:
                eng.compute_cor()  # matrix stays resident in HBM
        if use_dist and xdev is not None:
            eng.level0_dev(rank, world, xdev)
        elif l0sim["mode"] == "replay":
            l0sim["rank"] = sim.get("rank", 0)
            eng.level0_dev(l0sim["rank"], args.simulate_world, (l0_prepare, l0_exchange))
        else:
            eng.level0()
        sim["k"] = 0
        if sim["mode"] == "record":
            sim["rounds"] = []
            return eng.lgl(feed_forward=bool(ff), round_size=R, allgather=cb, max_targets=args.max_targets, edge_dict=False)
        if use_dist and not args.host_exchange:
            # per-round exchange with the payload packed / unpacked by the library and gathered in device memory (fw_learn_network_dev)
            return eng.lgl(feed_forward=bool(ff), round_size=R, rank=rank, world_size=world, dev_exchange=xround,
                           max_targets=args.max_targets, edge_dict=False)
        return eng.lgl(feed_forward=bool(ff), round_size=R, rank=sim.get("rank", rank),
                       world_size=max(world, args.simulate_world) if world == 1 else world, allgather=cb,
                       max_targets=args.max_targets, edge_dict=False)  # the network stays in the arrays the C ABI fills (no Python dictionary of tuples)

    def measure(ff, R, steps, warmup):
        """-> dict with the whole-job numbers of `steps` timed passes under schedule (ff, R)."""
        for _ in range(warmup):
            step(ff, R)
        eng.reset_counters()
        xstats.clear()
        cs0 = eng.comm_stats() if lib_comm else None
        barrier()
        t0 = time.perf_counter()
        net = None
        det = {"first": None, "differing": 0, "ref_first": None, "prev": 0, "which": []}
        for i_step in range(steps):
            net = step(ff, R)
            if args.check_determinism:  # (changes the timing: not for a measured line) every pass against the first: network bytes + reference-order test count
                key = tuple(net[k].tobytes() for k in ("edge_src", "edge_dst", "edge_weight", "pc_off", "pc_idx", "pc_weight", "pc_pval"))
                c_now = eng.counters()["cond_tests_ref"]
                n_ref = c_now - det["prev"]
                det["prev"] = c_now
                if det["first"] is None:
                    det["first"], det["ref_first"] = key, n_ref
                elif key != det["first"] or n_ref != det["ref_first"]:
                    det["differing"] += 1
                    if len(det["which"]) < 16:
                        det["which"].append([i_step, int(len(net["edge_src"])), int(n_ref)])
        barrier()
        dt = time.perf_counter() - t0
        if args.check_determinism:
            print("[bench] determinism check (ff=%d): %d of %d passes differ from the first (edges %d, reference-order tests %d): %s"
                  % (ff, det["differing"], steps, len(net["edge_src"]), det["ref_first"], det["which"]), file=sys.stderr, flush=True)
        if lib_comm:  # the library's own exchange counters (fw_comm_stats), same keys as the Python callbacks keep
            cs1 = eng.comm_stats()
            xstats.update({k: cs1[k] - cs0[k] for k in ("calls", "collectives", "entries", "seconds")})
            xstats["issued_by"] = "library (fw_rccl.cpp: ncclAllGather on the engine's stream)"
        cn = eng.counters()
        cond_ref, cond_eval = cn["cond_tests_ref"], cn["cond_tests_evaluated"]
        if use_dist:
            tt = torch.tensor([dt], dtype=torch.float64, device=cdev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
            tt = torch.tensor([cond_ref, cond_eval], dtype=torch.float64, device=cdev)
            dist.all_reduce(tt, op=dist.ReduceOp.SUM)
            cond_ref, cond_eval = int(tt[0].item()), int(tt[1].item())
        s = max(steps, 1)
        level0 = p * (p - 1) // 2
        rounds = 1 if (not ff or R <= 0) else (p + R - 1) // R
        return {"dt": dt, "cn": cn, "net": net, "level0": level0, "cond_ref": cond_ref // s, "cond_eval": cond_eval // s,
                "value": (level0 + cond_ref // s) * s / dt, "ms_per_step": 1e3 * dt / s, "rounds": rounds,
                "exchange": {"calls_per_step": xstats.get("calls", 0) / s, "collectives_per_step": xstats.get("collectives", 0) / s,
                             "entries_per_step": xstats.get("entries", 0) / s,
                             "seconds_per_step_rank0": xstats.get("seconds", 0.0) / s,
                             "us_per_round_rank0": 1e6 * xstats.get("seconds", 0.0) / max(xstats.get("calls", 0), 1),
                             "issued_by": xstats.get("issued_by", "host language (torch.distributed on buffers the library packs)")}
                if use_dist else None}

    auto_R = DEFAULT_ROUND * ((p + 10 * DEFAULT_ROUND - 1) // (10 * DEFAULT_ROUND))
    ff, R = int(bool(args.feed_forward)), (auto_R if args.round_size < 0 else args.round_size)
    if R <= 0:
        ff = 0
    def measure_sim(f, r, steps, warmup):
        """--simulate-world N: record one single-rank pass, then time every rank's share in turn (see `cb` above); the
        result is the SLOWEST rank's -- what the N-rank job takes apart from the exchange itself."""
        sim["mode"] = "record"
        step(f, r)
        recorded = sum(len(x[0]) for x in sim["rounds"])
        sim["mode"] = "replay"
        if shard_l0:
            l0_collect()
        ranks = range(args.simulate_world) if args.simulate_rank < 0 else [args.simulate_rank]
        per, found = [], 0
        for rk in ranks:
            sim["rank"], sim["local"] = rk, 0
            m = measure(f, r, steps, warmup)
            found += sim["local"] // (steps + warmup)
            per.append(m)
        if args.simulate_rank < 0 and found != recorded:
            raise SystemExit("simulate-world: the ranks' shares found %d directed entries, the single-rank pass %d" % (found, recorded))
        worst = max(per, key=lambda m: m["ms_per_step"])
        worst = dict(worst)
        worst["simulated_world"] = {"world_size": args.simulate_world, "ranks_timed": list(ranks),
                                    "ms_per_step_by_rank": [m["ms_per_step"] for m in per],
                                    "conditional_s_by_rank": [m["cn"]["t_cond_s"] / max(steps, 1) for m in per],
                                    "cond_ref_by_rank": [m["cond_ref"] for m in per],
                                    "directed_entries": recorded,
                                    "note": "one GPU timing each rank's share in turn; all-gathers replayed from a recorded "
                                            "single-rank pass (same whitelists as among N real ranks), no exchange cost"}
        # r05: the replay leaves the collectives out; what IS measured about them -- the library-side exchange with a world of one rank
        # (profiles/r04_bench_cfg3_n1_library_rccl_world1.json: 74-82 us per round incl. the header all-gather and both stream
        # synchronisations) and the xGMI ring rate of MI355X_MICROARCH.md for the payloads -- is added as a second figure, marked as a
        # model: N > 1 ranks over RCCL have never run (no multi-GPU node).
        n_rounds = max(1, len(sim["rounds"]))
        W = args.simulate_world
        per_round_s = 82e-6
        ring_bw = 153e9  # one xGMI link, bytes/s: a ring all-gather moves (W - 1) / W of the gathered payload over it
        payload = 24.0 * recorded  # per-round records of 24 bytes, summed over the rounds
        l0_bytes = 24.0 * float(sum(l0sim["counts"] or [0])) if shard_l0 else 0.0  # the ranks' packed level-0 records (pass A of the replay)
        xs = n_rounds * per_round_s + (W - 1) / W * (payload + l0_bytes) / ring_bw + (per_round_s if shard_l0 else 0.0)
        worst["simulated_world"]["exchange_model"] = {
            "rounds": n_rounds, "seconds_per_round_measured_world1": per_round_s, "payload_bytes": payload + l0_bytes,
            "ring_link_bytes_per_s": ring_bw, "exchange_seconds_added": xs,
            "ms_per_step_with_exchange": worst["ms_per_step"] + 1e3 * xs,
            "note": "measured part: 82 us per exchange with a world of one rank on the library's communicator; modelled part: payload over one "
                    "xGMI link at the ring rate.  UNMEASURED on more than one GPU."}
        # whole-job figures: every rank's tests, the slowest rank's time
        worst["cond_ref"] = sum(m["cond_ref"] for m in per) if args.simulate_rank < 0 else worst["cond_ref"]
        worst["cond_eval"] = sum(m["cond_eval"] for m in per) if args.simulate_rank < 0 else worst["cond_eval"]
        worst["value"] = (worst["level0"] + worst["cond_ref"]) / (worst["ms_per_step"] / 1e3)
        return worst

    simulating = args.simulate_world > 1 and world == 1
    meas = measure_sim if simulating else measure
    main_m = meas(ff, R if ff else 0, args.steps, args.warmup)
    other_m = None
    if not args.no_other_schedule:
        off, oR = (0, 0) if ff else (1, auto_R)
        other_m = meas(off, oR, max(1, min(args.steps, 3)), 1)
    # per-kernel figures: with the default two concurrent chains the launches of the two streams overlap, and per-launch
    # durations double-count the shared time (r02: kernel seconds 0.2355 > ms_per_step 0.2097).  The roofline therefore
    # comes from one extra pass with ONE chain -- the segment kernel alone on the GPU -- so that its kernel seconds are
    # a lower bound of that pass and comparable with the headline step.
    one_m = None
    if world == 1 and not simulating and not args.no_one_chain and not args.stream_columns and cfg["test_name"] in ("fz", "fz_nz"):
        knobs_were = os.environ.get("FW_KNOBS")
        os.environ["FW_KNOBS"], os.environ["FW_DH_CHAINS"] = "1", "1"  # (the library reads FW_* knobs only under FW_KNOBS=1)
        one_m = measure(ff, R if ff else 0, max(1, min(args.steps, 3)), 1)
        del os.environ["FW_DH_CHAINS"]
        if knobs_were is None:
            del os.environ["FW_KNOBS"]
    seam_m = None
    if args.host_seam and world == 1:
        knobs_were = os.environ.get("FW_KNOBS")
        os.environ["FW_KNOBS"], os.environ["FW_HOST_HITON"] = "1", "1"
        seam_m = measure(0, 0, 1, 0)
        del os.environ["FW_HOST_HITON"]
        if knobs_were is None:
            del os.environ["FW_KNOBS"]

    out = None
    if rank == 0:
        steps = max(args.steps, 1)
        cn, dt, net = main_m["cn"], main_m["dt"], main_m["net"]
        launches = max(cn["kernel_launches"], 1)
        kname = "fzs_subsets_seg_kernel" if args.stream_columns else "fz_subsets_seg_kernel" if cfg["test_name"] in ("fz", "fz_nz") else ("dh_mi_target_kernel" if cfg["test_name"] in ("mi", "mi_nz") else "mi_subsets_seg_kernel")
        rcn, rsteps, rsrc = (one_m["cn"], max(1, min(args.steps, 3)), "one-chain pass (FW_DH_CHAINS=1: the kernel alone on the GPU)") if one_m else \
                            (cn, steps, "headline pass")
        sub_launch_s = rcn["t_dev_subsets_s"]
        n_sub_launches = max(rcn["subsets_launches"], 1)
        achieved = (rcn["alg_bytes_subsets"] / max(sub_launch_s, 1e-12)) / 1e9
        # counters of the same kernel on the same workload from separate rocprofv3 --pmc passes (they cannot be read from inside
        # this process): HBM-side traffic per launch, VALU instructions per test, VALU-busy share.  profiles/README.md
        pmc, pmc_src = None, None
        for cand in ("r06_%s_pmc_summary.json" % args.config, "r05_%s_pmc_summary.json" % args.config, "r04_%s_pmc_summary.json" % args.config, "r03_%s_pmc_summary.json" % args.config, "r02_%s_pmc_summary.json" % args.config):
            pmc_path = os.path.join(ROOT, "profiles", cand)
            if not args.p and not args.n and os.path.exists(pmc_path):
                js = json.load(open(pmc_path))
                hit = [k for k in js if isinstance(js[k], dict) and k.startswith(kname.split("<")[0]) and "evaluated_tests" in js[k]]
                if hit:
                    pmc, pmc_src = js[hit[0]], "profiles/" + cand
                    break
        traffic = pmc.get("fetch_bytes_per_launch") if pmc else None
        valu = None
        if pmc and "valu_wave_insts_per_test" in pmc:
            # resident wavefronts per SIMD the dominant kernel is compiled for (build remarks): the persistent discrete kernel runs ONE
            # (r04 printed 4 x the VALU-busy share for it), the long-list max_k 4-5 segment kernel three, the size-3 one four
            wps_built = 1 if kname == "dh_mi_target_kernel" else (3 if cfg["max_k"] > 3 else 4)
            wps = wps_built if (kname == "dh_mi_target_kernel" or "valu_mix_wave_insts_per_test" in pmc) else pmc.get("waves_per_simd", wps_built)
            ipt = pmc["valu_wave_insts_per_test"]
            tps = rcn["cond_tests_evaluated"] / max(sub_launch_s, 1e-12)
            # issue roof: 256 CUs x 4 SIMDs, a 64-lane instruction holds a SIMD >= 2 cycles (32 lanes / clock for 32-bit ops, 16 for Float64)
            peak_wi = 256 * 4 * 2.4e9 / 2
            valu = {"wave_insts_per_test": ipt, "busy_frac": min(1.0, pmc.get("active_inst_valu_frac_of_wave_cycles", 0.0) * wps),
                    "waves_per_simd": wps, "achieved_wave_insts_per_s": ipt * tps, "issue_peak_wave_insts_per_s": peak_wi,
                    "issue_frac_if_all_32bit": ipt * tps / peak_wi,
                    "note": "busy_frac = SQ_ACTIVE_INST_VALU x waves per SIMD / SQ_WAVE_CYCLES from the tracked PMC pass: the share of "
                            "cycles in which the SIMD's vector ALU is executing; Float64 and transcendental instructions hold it 4-16 "
                            "cycles, so the issue fraction computed at the 32-bit rate understates it", "source": pmc_src}
            mixd = pmc.get("valu_mix_wave_insts_per_test")
            if mixd:
                # roofline.valu_frac: the kernel's measured instruction mix (SQ_INSTS_VALU_* per evaluated test) priced at the issue
                # cost of each class measured on this device (profiles/r04_valu_rate.txt, cycles per wave64 instruction and SIMD at
                # the nominal 2.4 GHz) -> cycles a test needs if nothing but VALU issue limited it -> tests/s at that bound
                CYC = {"f64_add": 5.1, "f64_mul": 5.2, "f64_fma": 5.3, "f64_trans": 17.3, "f32_add": 3.0, "f32_mul": 3.0, "f32_fma": 3.0,
                       "f32_trans": 9.0, "int32": 3.0, "int64": 6.0, "cvt": 3.0, "other": 3.0}
                cyc = sum(mixd.get(k, 0.0) * c for k, c in CYC.items())
                bound_tps = 256 * 4 * 2.4e9 / max(cyc, 1e-9)
                valu.update({"mix_wave_insts_per_test": mixd, "issue_cycles_per_test_and_simd": cyc, "issue_bound_tests_per_s": bound_tps,
                             "valu_frac": tps / bound_tps, "cycle_costs_source": "profiles/r04_valu_rate.txt (tools/valu_rate.cpp)"})
        # what the counters of the tracked PMC pass name as the limiter.  Fabric traffic well below the algorithmic bytes means the
        # operands are cache-resident and HBM is NOT the bound: then "valu" when the vector ALUs are busy >= 60 % of the cycles,
        # else "latency" (dependent chains / issue of a few resident wavefronts: the discrete persistent kernel).  "hbm" only when
        # the traffic is there, or when no PMC summary of this configuration is tracked (the nominal label of SURVEY 8d)
        low_traffic = pmc is not None and (traffic or 0) < 0.5 * rcn["alg_bytes_subsets"] / n_sub_launches
        bound = ("valu" if valu and valu["busy_frac"] >= 0.6 else "latency") if low_traffic else "hbm"
        # r06: a kernel that reaches less than 5 % of the nominal HBM rate with its vector ALUs busy less than 60 % of the time is
        # waiting -- on dependent loads, on other wavefronts -- whatever its fabric traffic looks like next to the algorithmic bytes
        # (cfg2 / cfg3he printed "hbm" at 0.003: their spill / record writes are of the size of their tiny algorithmic bytes)
        if valu and achieved / HBM_PEAK_GBS < 0.05 and valu["busy_frac"] < 0.6:
            bound = "latency"
        roofline = {"bound": bound, "kernel": kname,
                    "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                    "traffic": traffic, "traffic_source": (pmc_src + " (FETCH_SIZE + WRITE_SIZE per launch)") if traffic else None,
                    "valu": valu, "valu_frac": valu.get("valu_frac") if valu else None, "measured_on": rsrc,
                    "note": "achieved / peak / frac: nominal HBM roofline with the algorithmic bytes of SURVEY 8d (fz: 4*C(k+2,2)+32 B per "
                            "test, discrete: (k+2)*n*b/8+32 B) over the HIP-event time of the kernel's launches (device rounds: one launch "
                            "in four, rotating slot, scaled to all launches).  bound = the resource the counters name: with `valu`, the "
                            "gathered entries are L2-resident (traffic << algorithmic bytes) and the Float64 division / square-root "
                            "sequences keep the vector ALUs busy",
                    "kernel_seconds_per_step": sub_launch_s / rsteps,
                    "step_seconds_of_that_pass": (one_m["ms_per_step"] / 1e3) if one_m else dt / steps,
                    "stage": {"achieved": cn["alg_bytes_subsets"] / max(cn["t_cond_s"], 1e-12) / 1e9,
                              "frac": cn["alg_bytes_subsets"] / max(cn["t_cond_s"], 1e-12) / 1e9 / HBM_PEAK_GBS,
                              "conditional_stage_s": cn["t_cond_s"] / steps,
                              "note": "headline pass (two concurrent chains): algorithmic bytes / wall time of the whole conditional stage"},
                    "alg_bytes_per_launch": rcn["alg_bytes_subsets"] / n_sub_launches,
                    "avg_launch_us": 1e6 * sub_launch_s / n_sub_launches, "launches": n_sub_launches,
                    "evaluated_tests_per_s_in_kernel": rcn["cond_tests_evaluated"] / max(sub_launch_s, 1e-12)}
        if pmc and cfg["test_name"] in ("mi", "mi_nz") and pmc.get("write_bytes_raw") and pmc.get("evaluated_tests"):
            # discrete kinds: bytes the kernel writes per byte of result it owes (a 32-byte record per evaluated test) -- spills and board
            # / record traffic show up here long before they show up in the time (tracked PMC pass of the same command)
            roofline["write_over_result_bytes"] = pmc["write_bytes_raw"] / (32.0 * pmc["evaluated_tests"])
        if cn.get("t_l0_mfma_s", 0.0) > 0.0:
            # level 0 of the discrete kinds on the matrix cores (mi_level0_mfma_kernel, v_mfma_scale_f32_32x32x64_f8f6f4): a binary Gram
            # product priced against the dense fp4 peak (MI355X_MICROARCH.md: ~10 PFLOP/s; 256 CUs x 4 SIMDs x 2 x 65 536 / 32 cycles at 2.4 GHz)
            roofline["level0"] = {"kernel": "mi_level0_mfma_kernel", "bound": "mfma", "unit": "TFLOP/s", "peak": 10000.0,
                                  "achieved": cn["l0_mfma_flops"] / cn["t_l0_mfma_s"] / 1e12,
                                  "frac": cn["l0_mfma_flops"] / cn["t_l0_mfma_s"] / 1e16,
                                  "kernel_seconds_per_step": cn["t_l0_mfma_s"] / steps,
                                  "note": "HIP events around the kernel's launch on the engine's stream; flops = 2 x tiles x 256 x 256 x 64 W "
                                          "(the samples padded to whole 64-bit words); profiles/r05_level0_matrix_loop.json"}
        if args.stream_columns:
            # variant S (recursive_pcor = 0).  B_fzS = what a test that shares nothing streams.  The kernels share: the correlations of a
            # job are computed once per job from its columns (fzs_gram_kernel) and the tests condition sub-matrices of
            # that matrix out of LDS, so `achieved` is a multiple of the HBM rate and the bound is the conditioning arithmetic
            # (FW_FZS_GRAM=0: the per-test streaming form, 1.8 x the nominal rate with X / Y in LDS and the accepted columns in L2)
            roofline["served_by"] = ("job-local Float64 correlation matrices in LDS (fw_fzs.hip: fzs_gram_kernel + fzs_subsets_seg_kernel<.., GRAM>); "
                                     "profiles/r03_fzs_micro*.json")
            if rcn.get("gram_jobs", 0) > 0:
                # the job-matrix form has its own algorithmic unit (fw_counters.gram_*): what must cross HBM is a job's columns ONCE
                # ((a + 2) n 4 bytes) plus 32 bytes per test record, and the contraction is 2 n C(a + 2, 2) flops per job; B_fzS (columns
                # per test) described a form that shares nothing and gave "fractions" of 5.8 (r03 review)
                jb = rcn["gram_alg_bytes"] + 32.0 * rcn["cond_tests_evaluated"]
                roofline.update({"bound": "valu", "unit": "GB/s", "achieved": jb / max(sub_launch_s, 1e-12) / 1e9,
                                 "frac": jb / max(sub_launch_s, 1e-12) / 1e9 / HBM_PEAK_GBS, "alg_bytes_per_launch": jb / n_sub_launches,
                                 "job_matrices": {"jobs": rcn["gram_jobs"], "column_bytes": rcn["gram_alg_bytes"], "gram_flops": rcn["gram_alg_flops"],
                                                  "gram_tflops_over_kernel_time": rcn["gram_alg_flops"] / max(sub_launch_s, 1e-12) / 1e12,
                                                  "per_test_streaming_unit_bytes": rcn["alg_bytes_subsets"],
                                                  "note": "achieved / frac: (columns of every job once + 32 B per evaluated test) over the HIP-event time of the "
                                                          "variant's kernels; the limiter is the Float64 conditioning arithmetic of the tests on the LDS-resident "
                                                          "matrix (bound = valu), not HBM"}})
            else:
                roofline["bound"] = "valu" if roofline["frac"] > 1.0 else "hbm"
        sub_launch_s = cn["t_dev_subsets_s"]  # the stage table below reports the headline pass
        cpu, cpu_skipped = None, None
        if world > 1:
            cpu_skipped = "the CPU baseline is timed at N = 1 only"
        elif not args.no_cpu_baseline:
            cpu = cpu_baseline(args, cfg, data, n, p, eng, main_m["level0"])

        def sched(m, f, r):
            return {"feed_forward": f, "round_size": r, "rounds": m["rounds"], "value": m["value"], "unit": "tests/s",
                    "ms_per_step": m["ms_per_step"], "time_to_network_s": m["ms_per_step"] / 1e3,
                    "edges": int(len(m["net"]["edge_src"])),
                    "tests_per_step": {"level0": m["level0"], "conditional_ref_equivalent": m["cond_ref"],
                                       "conditional_evaluated": m["cond_eval"]},
                    "exchange": m["exchange"], "simulated_world": m.get("simulated_world")}

        out = {"metric": "ci_tests_per_sec", "value": main_m["value"], "unit": "tests/s", "n_gpus": world, "steps": args.steps,
               "warmup": args.warmup, "ms_per_step": main_m["ms_per_step"], "higher_is_better": True, "scaling": "strong",
               "vs_baseline": None, "dtype": "f64" if cfg["test_name"] in ("fz", "fz_nz") else "i32",
               "data": "bundled table of the reference's tests" if cfg.get("label") else "synthetic",
               "config": {"workload": cfg.get("label") or "%s: fwsynth-v1 %d OTUs x %d samples, %s, max_k=%d, alpha=0.01" %
                                      (args.config, p, n, cfg["test_name"], cfg["max_k"]),
                          "counts_sha256": csum, "recursive_pcor": 0 if args.stream_columns else 1, "dense_cor": 0 if args.no_cor_matrix else 1, "feed_forward": ff, "round_size": R if ff else 0,
                          "sampled_targets": args.max_targets or None,
                          "parallelism": "targets of each round dealt by estimated work over %d GPU(s), one rank per GPU, backend %s" %
                                         (world, (dist.get_backend() if use_dist else "none"))},
               "time_to_network_s": dt / steps, "normalisation": norm_rec, "edges": int(len(net["edge_src"])), "rounds": main_m["rounds"],
               # the learned network of the last timed pass as rank 0 holds it: edge list and weights -- equal across --gpus N (profiles/tools/scale_node.sh)
               "network_sha256": hashlib.sha256(b"".join(np.ascontiguousarray(net[k]).tobytes() for k in ("edge_src", "edge_dst", "edge_weight"))).hexdigest(),
               "tests_per_step": {"level0": main_m["level0"], "conditional_ref_equivalent": main_m["cond_ref"],
                                  "conditional_evaluated": main_m["cond_eval"]},
               "exchange": main_m["exchange"], "simulated_world": main_m.get("simulated_world"),
               "stage_seconds_rank0": {"level0": cn["t_level0_s"] / steps, "level0_host": cn["t_level0_host_s"] / steps, "conditional": cn["t_cond_s"] / steps,
                                       "subsets_kernels_device": sub_launch_s / steps,
                                       "host_advance": cn["t_host_advance_s"] / steps, "host_build": cn["t_host_build_s"] / steps,
                                       "host_launch": cn["t_host_launch_s"] / steps, "host_wait_device": cn["t_host_wait_s"] / steps, "host_merge": cn["t_host_merge_s"] / steps,
                                       "subsets_calls": cn["subsets_calls"] / steps},
               "kernel_launches_per_step": launches / steps,
               "other_schedule": sched(other_m, *((0, 0) if ff else (1, auto_R))) if other_m else None,
               "host_seam": ({"note": "FW_HOST_HITON=1: HITON-PC state machines on the host, every pool round = one window of every "
                                      "in-flight (T, candidate, accepted) job through the fw_test_subsets_batch kernels -- the seam "
                                      "hiton.jl:100 would call; feed_forward=0", **sched(seam_m, 0, 0)} if seam_m else None),
               "roofline": roofline, "cpu_baseline": cpu}
        if cpu_skipped:
            out["cpu_baseline_note"] = cpu_skipped
        print(json.dumps(out))
        sys.stdout.flush()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    eng.close()


if __name__ == "__main__":
    main()
