"""ctypes binding of include/flashweave_amd.h and the Python mirror of the reference's operator interface."""
import ctypes as C
import os
from collections import namedtuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
FW_MI, FW_MI_NZ, FW_FZ, FW_FZ_NZ = 0, 1, 2, 3
FW_MAX_K = 7
_KINDS = {"mi": FW_MI, "mi_nz": FW_MI_NZ, "fz": FW_FZ, "fz_nz": FW_FZ_NZ}

TestResult = namedtuple("TestResult", "stat pval df suff_power")  # src/types.jl:140-145


class FlashWeaveError(RuntimeError):
    """Raised for every non-zero ABI status (the reference throws Julia exceptions, e.g. learning.jl:72)."""

    def __init__(self, code, msg):
        super().__init__("[fw %d] %s" % (code, msg))
        self.code = code


class _Params(C.Structure):
    _fields_ = [("kind", C.c_int32), ("n", C.c_int32), ("p", C.c_int32), ("device", C.c_int32),
                ("max_k", C.c_int32), ("hps", C.c_int32), ("fdr", C.c_int32), ("dense_rules", C.c_int32),
                ("n_obs_min", C.c_int64), ("max_tests", C.c_int64), ("alpha", C.c_double),
                ("recursive_pcor", C.c_int32), ("no_cor_mat", C.c_int32)]


class _TestResult(C.Structure):
    _fields_ = [("stat", C.c_double), ("pval", C.c_double), ("df", C.c_int32), ("suff_power", C.c_int32)]


class _SubsetsResult(C.Structure):
    _fields_ = [("stat", C.c_double), ("pval", C.c_double), ("df", C.c_int32), ("suff_power", C.c_int32),
                ("status", C.c_int32), ("n_zs", C.c_int32), ("zs", C.c_int32 * FW_MAX_K), ("reserved0", C.c_int32),
                ("num_tests", C.c_int64), ("frac", C.c_double)]


class _Counters(C.Structure):
    _fields_ = [("level0_tests", C.c_int64), ("cond_tests_ref", C.c_int64), ("cond_tests_evaluated", C.c_int64),
                ("subsets_calls", C.c_int64), ("kernel_launches", C.c_int64), ("subsets_launches", C.c_int64), ("t_level0_s", C.c_double), ("t_level0_host_s", C.c_double),
                ("t_cond_s", C.c_double), ("t_dev_subsets_s", C.c_double), ("t_host_advance_s", C.c_double),
                ("t_host_build_s", C.c_double), ("t_host_launch_s", C.c_double), ("t_host_wait_s", C.c_double), ("t_host_merge_s", C.c_double),
                ("alg_bytes_subsets", C.c_double), ("gram_jobs", C.c_int64), ("gram_alg_bytes", C.c_double), ("gram_alg_flops", C.c_double),
                ("l0_mfma_flops", C.c_double), ("t_l0_mfma_s", C.c_double)]


class _LearnOpts(C.Structure):
    _fields_ = [("feed_forward", C.c_int32), ("round_size", C.c_int32), ("rank", C.c_int32),
                ("world_size", C.c_int32), ("max_targets", C.c_int32), ("reserved0", C.c_int32)]


PREPARE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                         C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64))
EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p)


class _DevExchange(C.Structure):  # fw_dev_exchange
    _fields_ = [("user", C.c_void_p), ("prepare", PREPARE_FN), ("exchange", EXCHANGE_FN)]


ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int64, C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                           C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int64),
                           C.POINTER(C.POINTER(C.c_int32)), C.POINTER(C.POINTER(C.c_int32)),
                           C.POINTER(C.POINTER(C.c_double)), C.POINTER(C.POINTER(C.c_double)))

_LIB = None


def lib_path():
    # profiling builds of the same sources (e.g. -DFW_MI_TICKS, profiles/tools): honoured only under FW_KNOBS=1, like every FW_* knob
    alt = os.environ.get("FW_LIB_PATH") if os.environ.get("FW_KNOBS") == "1" else None
    return alt if alt else os.path.join(_HERE, "libflashweave_amd.so")


def load_library():
    """dlopen the in-tree library.  torch (if importable) is imported first so that a single HIP runtime
    (libamdhip64.so.7) ends up in the process."""
    global _LIB
    if _LIB is not None:
        return _LIB
    so = lib_path()
    if not os.path.exists(so):
        raise FlashWeaveError(-2, "libflashweave_amd.so is missing (run __graft_entry__.build()); there is no "
                                  "CPU fallback for the HIP path")
    try:
        import torch  # noqa: F401
    except Exception:
        pass
    L = C.CDLL(so)
    vp = C.c_void_p
    L.fw_abi_version.restype = C.c_int
    L.fw_params_default.argtypes = [C.POINTER(_Params), C.c_int32, C.c_int32, C.c_int32]
    L.fw_ctx_create.argtypes = [C.POINTER(_Params), C.POINTER(vp)]
    L.fw_ctx_destroy.argtypes = [vp]
    L.fw_last_error.restype = C.c_char_p
    L.fw_last_error.argtypes = [vp]
    L.fw_set_data_dense_f32.argtypes = [vp, vp]
    L.fw_set_data_csc_i32.argtypes = [vp, vp, vp, vp]
    L.fw_set_data_dense_i32.argtypes = [vp, vp]
    L.fw_get_levels.argtypes = [vp, vp, vp]
    L.fw_set_cor_mat.argtypes = [vp, vp]
    L.fw_compute_cor_mat.argtypes = [vp]
    L.fw_get_cor_mat.argtypes = [vp, vp]
    L.fw_level0.argtypes = [vp, C.POINTER(C.c_int64)]
    L.fw_level0_get.argtypes = [vp, vp, vp, vp, vp]
    L.fw_set_row_views.argtypes = [vp, C.c_int32]
    L.fw_normalize_counts.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_int32, vp, vp, vp, vp, vp, C.POINTER(C.c_int32),
                                      C.POINTER(C.c_int32)]
    L.fw_level0_sharded.argtypes = [vp, C.c_int32, C.c_int32, vp, vp, C.POINTER(C.c_int64)]
    L.fw_level0_sharded_dev.argtypes = [vp, C.c_int32, C.c_int32, C.POINTER(_DevExchange), C.POINTER(C.c_int64)]
    L.fw_use_cor_buffer.argtypes = [vp, vp, C.c_int64]
    L.fw_compute_cor_mat_rows.argtypes = [vp, C.c_int32, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    L.fw_cor_mat_ready.argtypes = [vp]
    L.fw_test_batch.argtypes = [vp, C.c_int64, vp, vp, vp, vp, vp]
    L.fw_test_subsets_batch.argtypes = [vp, C.c_int64, vp, vp, vp, vp, vp]
    L.fw_learn_network.argtypes = [vp, C.POINTER(_LearnOpts), vp, vp, C.POINTER(C.c_int64)]
    L.fw_learn_network_dev.argtypes = [vp, C.POINTER(_LearnOpts), C.POINTER(_DevExchange), C.POINTER(C.c_int64)]
    L.fw_network_get.argtypes = [vp, vp, vp, vp]
    L.fw_network_get_directed.argtypes = [vp, vp, vp, vp, vp]
    L.fw_get_counters.argtypes = [vp, C.POINTER(_Counters)]
    L.fw_reset_counters.argtypes = [vp]
    if hasattr(L, "fw_comm_init"):
        L.fw_comm_unique_id.argtypes = [vp]
        L.fw_comm_init.argtypes = [vp, vp, C.c_int32, C.c_int32]
        L.fw_comm_destroy.argtypes = [vp]
        L.fw_comm_stats.argtypes = [vp, vp, vp, vp, vp, vp]
        L.fw_level0_comm.argtypes = [vp, vp]
        L.fw_cor_mat_allgather_comm.argtypes = [vp, C.c_int64]
        L.fw_learn_network_comm.argtypes = [vp, vp, vp]
    if hasattr(L, "fw_selftest"):  # (absent from older builds loaded through FW_LIB_PATH for A/B profiling)
        L.fw_selftest.argtypes = [vp, C.c_int, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64)]
    L.fw_effective_n_obs_min.restype = C.c_int64
    L.fw_effective_n_obs_min.argtypes = [vp]
    _LIB = L
    return L


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def normalize_counts(counts, test_name, device=0):
    """Normalisation front-end on the device (fw_normalize_counts): -> (data, row_mask, col_mask) like preprocess.normalize,
    for every test_name ("fz": clr_adapt, "fz_nz": clr_nz, "mi": binary, "mi_nz": binned_nz_clr)."""
    L = load_library()
    raw = np.asarray(counts)
    if raw.ndim != 2:
        raise ValueError("normalize_counts: counts must be a samples x OTUs matrix")
    if not np.issubdtype(raw.dtype, np.integer):  # a silent cast would truncate relative abundances / floats to zero
        if not np.all(np.isfinite(raw)) or np.any(raw != np.floor(raw)):
            raise TypeError("normalize_counts: the device front-end takes integer counts (got non-integral %s values)" % raw.dtype)
    if raw.size and (raw.min() < 0 or raw.max() > np.iinfo(np.int32).max):
        raise ValueError("normalize_counts: counts must lie in [0, 2^31 - 1]")
    x = np.asfortranarray(raw.astype(np.int32))
    n, p = x.shape
    rm, cm = np.zeros(n, np.uint8), np.zeros(p, np.uint8)
    no, po = C.c_int32(0), C.c_int32(0)
    kind = _KINDS[test_name]
    of = np.zeros(n * p, np.float32) if kind in (FW_FZ, FW_FZ_NZ) else None
    oi = np.zeros(n * p, np.int32) if kind in (FW_MI, FW_MI_NZ) else None
    rc = L.fw_normalize_counts(device, kind, n, p, _ptr(x), _ptr(of), _ptr(oi), _ptr(rm), _ptr(cm), C.byref(no), C.byref(po))
    if rc != 0:
        raise FlashWeaveError(rc, L.fw_last_error(None).decode())
    out = (of if of is not None else oi)[:no.value * po.value].reshape((no.value, po.value), order="F")
    return out, rm.astype(bool), cm.astype(bool)


class Engine:
    """One engine context on one GPU (replaces make_test_object, src/misc.jl:34-45).

    test_name: "mi" | "mi_nz" | "fz" (src/types.jl:64-72).  Keyword defaults are learn_network's
    (src/learning.jl:466-473)."""

    def __init__(self, test_name, n, p, max_k=3, alpha=0.01, hps=5, n_obs_min=-1, max_tests=10_000_000, FDR=True,
                 device=0, dense_rules=False, recursive_pcor=True, dense_cor=True):
        self.L = load_library()
        self.test_name = test_name
        self.n, self.p = int(n), int(p)
        P = _Params()
        self.L.fw_params_default(C.byref(P), _KINDS[test_name], self.n, self.p)
        P.device, P.max_k, P.alpha, P.hps = device, max_k, alpha, hps
        P.n_obs_min, P.max_tests, P.fdr = n_obs_min, max_tests, int(FDR)
        P.recursive_pcor = int(bool(recursive_pcor))  # False: conditional fz tests stream the sample columns (no cor_mat, statfuns.jl:19-21)
        P.no_cor_mat = int(not dense_cor)  # dense_cor = False (learning.jl:42): no p x p matrix at all; needs recursive_pcor = False
        P.dense_rules = int(bool(dense_rules))  # Matrix (dense) table methods instead of the SparseMatrixCSC ones
        self.h = C.c_void_p()
        rc = self.L.fw_ctx_create(C.byref(P), C.byref(self.h))
        if rc != 0:
            raise FlashWeaveError(rc, self.L.fw_last_error(None).decode())
        self.max_k = max_k
        self._cb = None

    # -- plumbing ------------------------------------------------------------------------------------
    def _ck(self, rc):
        if rc != 0:
            raise FlashWeaveError(rc, self.L.fw_last_error(self.h).decode())

    def close(self):
        if getattr(self, "h", None):
            self.L.fw_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- data ----------------------------------------------------------------------------------------
    def set_data(self, data):
        """fz: dense Float32 n x p; mi / mi_nz: integer n x p (dense ndarray) or a (colptr, rowval, nzval) CSC triple
        with 0-based rows."""
        if self.test_name in ("fz", "fz_nz"):
            d = np.asfortranarray(np.asarray(data, dtype=np.float32))
            assert d.shape == (self.n, self.p)
            self._ck(self.L.fw_set_data_dense_f32(self.h, _ptr(d)))
        elif isinstance(data, tuple):
            colptr, rowval, nzval = (np.ascontiguousarray(data[0], dtype=np.int64),
                                     np.ascontiguousarray(data[1], dtype=np.int32),
                                     np.ascontiguousarray(data[2], dtype=np.int32))
            self._ck(self.L.fw_set_data_csc_i32(self.h, _ptr(colptr), _ptr(rowval), _ptr(nzval)))
        else:
            d = np.asfortranarray(np.asarray(data, dtype=np.int32))
            assert d.shape == (self.n, self.p)
            self._ck(self.L.fw_set_data_dense_i32(self.h, _ptr(d)))

    def set_cor_mat(self, cor_mat):
        cm = np.asfortranarray(np.asarray(cor_mat, dtype=np.float32))
        assert cm.shape == (self.p, self.p)
        self._ck(self.L.fw_set_cor_mat(self.h, _ptr(cm)))

    def cor(self):
        """cor(data_dense) -> Float32 p x p, on the MFMA units (src/learning.jl:44)."""
        self._ck(self.L.fw_compute_cor_mat(self.h))
        return self.cor_mat()

    def compute_cor(self):
        """Device-only form of cor(): the matrix stays resident, nothing is copied back."""
        self._ck(self.L.fw_compute_cor_mat(self.h))

    def cor_mat(self):
        out = np.zeros((self.p, self.p), dtype=np.float32, order="F")
        self._ck(self.L.fw_get_cor_mat(self.h, _ptr(out)))
        return out

    def set_row_views(self, on=True):
        """mi_nz + dense_rules: test_subsets on the (T, candidate) row views hiton.jl uses (include/flashweave_amd.h)."""
        self._ck(self.L.fw_set_row_views(self.h, int(bool(on))))

    def levels(self):
        lv, mv = np.zeros(self.p, np.int32), np.zeros(self.p, np.int32)
        self._ck(self.L.fw_get_levels(self.h, _ptr(lv), _ptr(mv)))
        return lv, mv

    @property
    def n_obs_min(self):
        return int(self.L.fw_effective_n_obs_min(self.h))

    # -- level 0 -------------------------------------------------------------------------------------
    def level0(self, rank=0, world_size=1, allgather=None):
        """Runs level 0 and keeps the neighbour lists in the context (no copy to Python); returns the entry count.
        world_size > 1: this rank screens its share of the pair tiles (discrete kinds) and the significant pairs are
        exchanged through `allgather` (flashweave.jl_amd/dist.py)."""
        nnz = C.c_int64(0)
        if world_size > 1:
            cb = ALLGATHER_FN(allgather)
            self._cb0 = cb
            self._ck(self.L.fw_level0_sharded(self.h, rank, world_size, C.cast(cb, C.c_void_p), None, C.byref(nnz)))
        else:
            self._ck(self.L.fw_level0(self.h, C.byref(nnz)))
        return nnz.value

    def level0_dev(self, rank, world_size, exchange):
        """Level 0 of a target-sharded run with the exchange kept in device memory (fw_level0_sharded_dev): `exchange` is a
        (prepare, exchange) pair of Python callables (dist.make_dev_exchange)."""
        nnz = C.c_int64(0)
        x = _DevExchange(None, PREPARE_FN(exchange[0]), EXCHANGE_FN(exchange[1]))
        self._xdev = x
        self._ck(self.L.fw_level0_sharded_dev(self.h, rank, world_size, C.byref(x), C.byref(nnz)))
        return nnz.value

    # -- library-side collectives (fw_comm_*: RCCL on a communicator the library owns) -------------------
    @staticmethod
    def comm_unique_id():
        """Rank 0: the 128-byte rendezvous id (ncclGetUniqueId); ship it to every rank, then comm_init everywhere."""
        L = load_library()
        buf = (C.c_uint8 * 128)()
        rc = L.fw_comm_unique_id(buf)
        if rc:
            raise FlashWeaveError(rc, (L.fw_last_error(None) or b"").decode())
        return bytes(buf)

    def comm_init(self, id128, rank, world_size):
        buf = (C.c_uint8 * 128).from_buffer_copy(bytes(id128))
        self._ck(self.L.fw_comm_init(self.h, buf, int(rank), int(world_size)))

    def comm_destroy(self):
        self._ck(self.L.fw_comm_destroy(self.h))

    def comm_stats(self):
        a, b, c_, d = C.c_int64(0), C.c_int64(0), C.c_int64(0), C.c_int64(0)
        s_ = C.c_double(0)
        self._ck(self.L.fw_comm_stats(self.h, C.byref(a), C.byref(b), C.byref(c_), C.byref(d), C.byref(s_)))
        return dict(calls=a.value, collectives=b.value, entries=c_.value, bytes=d.value, seconds=s_.value)

    def level0_comm(self):
        """fw_level0_comm: level 0 with this rank's share of the pair tiles (discrete kinds), significant pairs all-gathered by the library."""
        nnz = C.c_int64(0)
        self._ck(self.L.fw_level0_comm(self.h, C.byref(nnz)))
        return nnz.value

    def cor_allgather_comm(self, rows_per_rank):
        self._ck(self.L.fw_cor_mat_allgather_comm(self.h, int(rows_per_rank)))

    def lgl_comm(self, feed_forward=True, round_size=1, max_targets=0, edge_dict=True):
        """fw_learn_network_comm: LGL of a target-sharded run, the per-round exchange issued by the library (rank / world_size are the
        communicator's)."""
        opts = _LearnOpts(int(feed_forward), int(round_size), 0, 1, int(max_targets), 0)
        ne = C.c_int64(0)
        self._ck(self.L.fw_learn_network_comm(self.h, C.byref(opts), C.byref(ne)))
        return self._network(ne.value, edge_dict)

    # -- row-block sharding of cor() ------------------------------------------------------------------
    def use_cor_buffer(self, device_ptr, capacity_floats):
        """Keep the p x p matrix in caller-owned device memory (a torch tensor's data_ptr()): fw_use_cor_buffer."""
        self._ck(self.L.fw_use_cor_buffer(self.h, C.c_void_p(device_ptr), int(capacity_floats)))

    def compute_cor_rows(self, rank, world_size):
        """This rank's row block of the matrix -> (row0, rows_per_rank); gather the blocks, then cor_ready()."""
        r0, rp = C.c_int64(0), C.c_int64(0)
        self._ck(self.L.fw_compute_cor_mat_rows(self.h, rank, world_size, C.byref(r0), C.byref(rp)))
        return r0.value, rp.value

    def cor_ready(self):
        self._ck(self.L.fw_cor_mat_ready(self.h))

    def pw_univar_neighbors(self):
        """pw_univar_neighbors (src/tests.jl:436-532) -> CSR dict(off, idx, stat, pval)."""
        nnz = C.c_int64(0)
        self._ck(self.L.fw_level0(self.h, C.byref(nnz)))
        return self.pw_univar_neighbors_get()

    def pw_univar_neighbors_get(self):
        """The neighbour lists of the last level-0 run (fw_level0_get) -> CSR dict(off, idx, stat, pval)."""
        off = np.zeros(self.p + 1, np.int64)
        self._ck(self.L.fw_level0_get(self.h, _ptr(off), None, None, None))
        nnz = C.c_int64(int(off[-1]))
        k = max(nnz.value, 1)
        idx, stat, pv = np.zeros(k, np.int32), np.zeros(k, np.float64), np.zeros(k, np.float64)
        self._ck(self.L.fw_level0_get(self.h, _ptr(off), _ptr(idx), _ptr(stat), _ptr(pv)))
        return dict(off=off, idx=idx[:nnz.value], stat=stat[:nnz.value], pval=pv[:nnz.value])

    # -- single tests --------------------------------------------------------------------------------
    def test_batch(self, X, Y, Zs_list):
        """Batch of test(X, Y, Zs, ...) (src/tests.jl:28,108,184,250)."""
        m = len(X)
        Xa, Ya = np.asarray(X, np.int32), np.asarray(Y, np.int32)
        zoff = np.zeros(m + 1, np.int64)
        for i, z in enumerate(Zs_list):
            zoff[i + 1] = zoff[i] + len(z)
        zflat = np.array([v for z in Zs_list for v in z] or [0], dtype=np.int32)
        out = (_TestResult * m)()
        self._ck(self.L.fw_test_batch(self.h, m, _ptr(Xa), _ptr(Ya), _ptr(zoff), _ptr(zflat), out))
        return [TestResult(o.stat, o.pval, o.df, bool(o.suff_power)) for o in out]

    def test(self, X, Y, Zs=()):
        return self.test_batch([X], [Y], [tuple(Zs)])[0]

    def test_subsets_batch(self, T, cand, accepted_list):
        """Batch of test_subsets(T, candidate, accepted, ...) (src/tests.jl:281-346)."""
        m = len(T)
        Ta, Ca = np.asarray(T, np.int32), np.asarray(cand, np.int32)
        off = np.zeros(m + 1, np.int64)
        for i, a in enumerate(accepted_list):
            off[i + 1] = off[i] + len(a)
        flat = np.array([v for a in accepted_list for v in a] or [0], dtype=np.int32)
        out = (_SubsetsResult * m)()
        self._ck(self.L.fw_test_subsets_batch(self.h, m, _ptr(Ta), _ptr(Ca), _ptr(off), _ptr(flat), out))
        return [dict(status=o.status, stat=o.stat, pval=o.pval, df=o.df, suff_power=bool(o.suff_power),
                     Zs=tuple(o.zs[:o.n_zs]), num_tests=o.num_tests, frac=o.frac) for o in out]

    def test_subsets(self, T, cand, accepted):
        return self.test_subsets_batch([T], [cand], [list(accepted)])[0]

    def selftest(self, which=1, cases=1 << 28, seed=1):
        """fw_selftest: device-side bit comparison of a hand-written arithmetic sequence with the compiler's (1 = Float64 division)."""
        bad = C.c_uint64(0)
        self._ck(self.L.fw_selftest(self.h, which, cases, seed, C.byref(bad)))
        return int(bad.value)

    # -- LGL -----------------------------------------------------------------------------------------
    def lgl(self, feed_forward=True, round_size=1, rank=0, world_size=1, max_targets=0, allgather=None, edge_dict=True, dev_exchange=None):
        """LGL minus normalisation (src/learning.jl:203-279).  Returns dict(edges={(i,j): w}, directed=CSR);
        edge_dict=False leaves the edges as the three arrays fw_network_get fills (edge_src, edge_dst, edge_weight) and
        skips the Python dictionary (48 000 tuples cost ~8 ms at cfg3)."""
        opts = _LearnOpts(int(feed_forward), int(round_size), int(rank), int(world_size), int(max_targets), 0)
        ne = C.c_int64(0)
        cb = None
        if allgather is not None:
            cb = ALLGATHER_FN(allgather)
            self._cb = cb
        if dev_exchange is not None:  # (prepare, exchange) of dist.make_dev_exchange: the library packs / unpacks, Python runs the collective
            x = _DevExchange(None, PREPARE_FN(dev_exchange[0]), EXCHANGE_FN(dev_exchange[1]))
            self._xdev_lgl = x
            self._ck(self.L.fw_learn_network_dev(self.h, C.byref(opts), C.byref(x), C.byref(ne)))
        else:
            self._ck(self.L.fw_learn_network(self.h, C.byref(opts), C.cast(cb, C.c_void_p) if cb else None, None, C.byref(ne)))
        return self._network(ne.value, edge_dict)

    def _network(self, n_edges, edge_dict):
        ne = C.c_int64(n_edges)
        k = max(ne.value, 1)
        src, dst, w = np.zeros(k, np.int32), np.zeros(k, np.int32), np.zeros(k, np.float64)
        self._ck(self.L.fw_network_get(self.h, _ptr(src), _ptr(dst), _ptr(w)))
        off = np.zeros(self.p + 1, np.int64)
        self._ck(self.L.fw_network_get_directed(self.h, _ptr(off), None, None, None))
        kk = max(int(off[-1]), 1)
        idx, pw, pp = np.zeros(kk, np.int32), np.zeros(kk, np.float64), np.zeros(kk, np.float64)
        self._ck(self.L.fw_network_get_directed(self.h, _ptr(off), _ptr(idx), _ptr(pw), _ptr(pp)))
        m = ne.value
        out = dict(edge_src=src[:m], edge_dst=dst[:m], edge_weight=w[:m], pc_off=off, pc_idx=idx[:off[-1]],
                   pc_weight=pw[:off[-1]], pc_pval=pp[:off[-1]])
        if edge_dict:
            out["edges"] = dict(zip(zip(src[:m].tolist(), dst[:m].tolist()), w[:m].tolist()))  # python ints / floats
        return out

    def counters(self):
        cn = _Counters()
        self._ck(self.L.fw_get_counters(self.h, C.byref(cn)))
        return {f: getattr(cn, f) for f, _ in _Counters._fields_}

    def reset_counters(self):
        self._ck(self.L.fw_reset_counters(self.h))
