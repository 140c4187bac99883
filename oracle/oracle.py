"""ctypes front-end of the CPU oracle (oracle/fw_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg as the checker / the reported CPU baseline.  The product package
(flashweave.jl_amd) never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class Result(C.Structure):
    _fields_ = [("stat", C.c_double), ("pval", C.c_double), ("df", C.c_int64),
                ("suff_power", C.c_int32), ("pad", C.c_int32)]


class Params(C.Structure):
    _fields_ = [("alpha", C.c_double), ("hps", C.c_int), ("n_obs_min", C.c_int64), ("max_k", C.c_int),
                ("max_tests", C.c_int64), ("FDR", C.c_int), ("feed_forward", C.c_int),
                ("round_size", C.c_int), ("max_targets", C.c_int), ("target_stride", C.c_int),
                ("max_seconds", C.c_double), ("target_offset", C.c_int), ("deadline", C.c_double)]


def build(force=False):
    so = os.path.join(_HERE, "libfw_oracle.so")
    src = os.path.join(_HERE, "fw_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "libfw_oracle.so")
        if not os.path.exists(so):
            build()
        L = C.CDLL(so)
        vp = C.c_void_p
        L.fwo_create_discrete_sparse.restype = vp
        L.fwo_create_discrete_sparse.argtypes = [C.c_int, C.c_int, vp, vp, vp, C.c_int, C.c_int]
        L.fwo_create_discrete_dense.restype = vp
        L.fwo_create_discrete_dense.argtypes = [C.c_int, C.c_int, vp, C.c_int, C.c_int]
        L.fwo_create_fz.restype = vp
        L.fwo_create_fz.argtypes = [C.c_int, C.c_int, vp, vp]
        L.fwo_create_fz_nz.restype = vp
        L.fwo_create_fz_nz.argtypes = [C.c_int, C.c_int, vp, C.c_int]
        L.fwo_destroy.argtypes = [vp]
        L.fwo_L.argtypes = [vp]
        L.fwo_get_levels.argtypes = [vp, vp, vp]
        L.fwo_test.argtypes = [vp, C.c_int, C.c_int, vp, C.c_int, C.c_int, C.c_int64, C.POINTER(Result)]
        L.fwo_test_subsets.restype = C.c_int
        L.fwo_test_subsets.argtypes = [vp, C.c_int, C.c_int, vp, C.c_int, C.c_int, C.c_double, C.c_int, C.c_int64,
                                       C.c_int64, C.POINTER(Result), vp, C.POINTER(C.c_int),
                                       C.POINTER(C.c_int64), C.POINTER(C.c_double)]
        L.fwo_contingency_table.restype = C.c_int
        L.fwo_contingency_table.argtypes = [vp, C.c_int, C.c_int, vp, C.c_int, vp, C.c_int64]
        L.fwo_mutual_information.restype = C.c_double
        L.fwo_mutual_information.argtypes = [vp, C.c_int, C.c_int, C.c_int]
        L.fwo_mi_pval.restype = C.c_double
        L.fwo_mi_pval.argtypes = [C.c_double, C.c_int64, C.c_int64]
        L.fwo_chisq_ccdf.restype = C.c_double
        L.fwo_chisq_ccdf.argtypes = [C.c_int64, C.c_double]
        L.fwo_fz_pval.restype = C.c_double
        L.fwo_fz_pval.argtypes = [C.c_double, C.c_int64, C.c_int64]
        L.fwo_pcor_rec.restype = C.c_double
        L.fwo_pcor_rec.argtypes = [vp, C.c_int, C.c_int, vp, C.c_int]
        L.fwo_pcor.restype = C.c_double
        L.fwo_pcor.argtypes = [vp, C.c_int, C.c_int, vp, C.c_int]
        L.fwo_fz_set_data.argtypes = [vp, vp, C.c_int]
        L.fwo_fz_nz_set_stream.argtypes = [vp, C.c_int]
        L.fwo_benjamini_hochberg.argtypes = [vp, C.c_int64, C.c_double, C.c_int64]
        L.fwo_level0.restype = vp
        L.fwo_level0.argtypes = [vp, C.c_double, C.c_int, C.c_int64, C.c_int, C.c_int]
        L.fwo_nbrs_free.argtypes = [vp]
        L.fwo_nbrs_from_csr.restype = vp
        L.fwo_nbrs_from_csr.argtypes = [C.c_int, vp, vp, vp, vp, C.c_int64]
        L.fwo_level0_sample.restype = C.c_int64
        L.fwo_level0_sample.argtypes = [vp, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_double, C.POINTER(C.c_double)]
        L.fwo_level0_rows.restype = C.c_int64
        L.fwo_level0_rows.argtypes = [vp, C.c_double, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int64, vp, vp, vp, vp,
                                      C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        L.fwo_nbrs_total.restype = C.c_int64
        L.fwo_nbrs_total.argtypes = [vp, C.c_int]
        L.fwo_nbrs_ntests.restype = C.c_int64
        L.fwo_nbrs_ntests.argtypes = [vp]
        L.fwo_nbrs_copy.argtypes = [vp, C.c_int, vp, vp, vp, vp]
        L.fwo_auto_n_obs_min.restype = C.c_int64
        L.fwo_auto_n_obs_min.argtypes = [vp, C.c_int64, C.c_int, C.c_int]
        L.fwo_learn.restype = vp
        L.fwo_learn.argtypes = [vp, C.POINTER(Params), vp]
        L.fwo_learn_mt.restype = vp
        L.fwo_learn_mt.argtypes = [vp, C.c_int, C.POINTER(Params), vp]
        L.fwo_network_free.argtypes = [vp]
        L.fwo_network_nedges.restype = C.c_int64
        L.fwo_network_nedges.argtypes = [vp]
        L.fwo_network_copy.argtypes = [vp, vp, vp, vp]
        L.fwo_network_stats.argtypes = [vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_double),
                                        C.POINTER(C.c_double), C.POINTER(C.c_int)]
        L.fwo_network_pc_total.restype = C.c_int64
        L.fwo_network_pc_total.argtypes = [vp, C.c_int]
        L.fwo_network_pc_copy.argtypes = [vp, C.c_int, vp, vp, vp, vp]
        L.fwo_cor.argtypes = [vp, C.c_int, C.c_int, vp, vp]
        _LIB = L
    return _LIB


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def dense_to_csc(mat):
    """n x p integer matrix -> (colptr int64, rowval int32 0-based, nzval int32), like Julia's sparse()."""
    mat = np.asarray(mat)
    n, p = mat.shape
    colptr = np.zeros(p + 1, dtype=np.int64)
    rows, vals = [], []
    for v in range(p):
        nzr = np.nonzero(mat[:, v])[0]
        rows.append(nzr.astype(np.int32))
        vals.append(mat[nzr, v].astype(np.int32))
        colptr[v + 1] = colptr[v] + len(nzr)
    rowval = np.concatenate(rows) if rows else np.zeros(0, np.int32)
    nzval = np.concatenate(vals) if vals else np.zeros(0, np.int32)
    return colptr, np.ascontiguousarray(rowval), np.ascontiguousarray(nzval)


def cor(data, out="f32"):
    """Pearson matrix of an n x p matrix, Float64 accumulation (learning.jl:42-45 with prec=64)."""
    d = np.asfortranarray(np.asarray(data, dtype=np.float64))
    n, p = d.shape
    o32 = np.zeros((p, p), dtype=np.float32, order="F")
    o64 = np.zeros((p, p), dtype=np.float64, order="F")
    lib().fwo_cor(_ptr(d), n, p, _ptr(o32), _ptr(o64))
    return o32 if out == "f32" else o64


class Oracle:
    """One oracle context.  kind in {'mi', 'mi_nz', 'fz'}."""

    def __init__(self, kind, data=None, sparse=True, max_k=3, cor_mat=None, n_obs=None, csc=None, shape=None):
        """csc = (colptr, rowval, nzval) + shape = (n, p): a prebuilt CSC triple (shared by several contexts, e.g. one
        per thread of bench.py's multi-core leg; a context only holds scratch of its own)."""
        self.kind = kind
        self.L = lib()
        self._keep = []
        self._post = []  # set-up calls to replay on the per-thread contexts of learn(threads > 1)
        if kind in ("mi", "mi_nz"):
            if csc is not None:
                self.n, self.p = shape
            else:
                mat = np.asarray(data)
                self.n, self.p = mat.shape
            nz = 1 if kind == "mi_nz" else 0
            if sparse:
                colptr, rowval, nzval = csc if csc is not None else dense_to_csc(mat)
                self._keep += [colptr, rowval, nzval]
                self._mk = lambda: self.L.fwo_create_discrete_sparse(self.n, self.p, _ptr(colptr), _ptr(rowval),
                                                                      _ptr(nzval), nz, max_k)
                self.h = self._mk()
            else:
                d = np.asfortranarray(mat.astype(np.int32))
                self._keep.append(d)
                self._mk = lambda: self.L.fwo_create_discrete_dense(self.n, self.p, _ptr(d), nz, max_k)
                self.h = self._mk()
        elif kind == "fz":
            cm = np.asarray(cor_mat)
            self.p = cm.shape[0]
            self.n = int(n_obs)
            if cm.dtype == np.float32:
                cm = np.asfortranarray(cm)
                self._keep.append(cm)
                self._mk = lambda: self.L.fwo_create_fz(self.n, self.p, _ptr(cm), None)
                self.h = self._mk()
            else:
                cm = np.asfortranarray(cm.astype(np.float64))
                self._keep.append(cm)
                self._mk = lambda: self.L.fwo_create_fz(self.n, self.p, None, _ptr(cm))
                self.h = self._mk()
        elif kind == "fz_nz":
            arr = np.asarray(data)
            self.n, self.p = arr.shape
            is_f32 = 1 if arr.dtype == np.float32 else 0
            d = np.asfortranarray(arr.astype(np.float64))
            self._keep.append(d)
            self._mk = lambda: self.L.fwo_create_fz_nz(self.n, self.p, _ptr(d), is_f32)
            self.h = self._mk()
        else:
            raise ValueError(kind)

    def close(self):
        if self.h:
            self.L.fwo_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def levels(self):
        lv = np.zeros(self.p, np.int32)
        mv = np.zeros(self.p, np.int32)
        self.L.fwo_get_levels(self.h, _ptr(lv), _ptr(mv))
        return lv, mv

    def auto_n_obs_min(self, n_obs_min=-1, hps=5, max_k=3):
        return int(self.L.fwo_auto_n_obs_min(self.h, n_obs_min, hps, max_k))

    def test(self, X, Y, Zs=(), hps=5, n_obs_min=0):
        z = np.asarray(Zs, dtype=np.int32)
        r = Result()
        self.L.fwo_test(self.h, X, Y, _ptr(z), len(z), hps, n_obs_min, C.byref(r))
        return (r.stat, r.pval, int(r.df), bool(r.suff_power))

    def set_fz_data(self, data, stream=True):
        """fz: attach the normalised n x p matrix; stream=True: conditional tests use pcor (no cor_mat, statfuns.jl:19-21)."""
        d = np.asfortranarray(np.asarray(data, dtype=np.float64))
        self._keep.append(d)
        self._post.append(lambda h: self.L.fwo_fz_set_data(h, _ptr(d), int(stream)))
        self._post[-1](self.h)

    def set_fz_nz_stream(self, stream=True):
        """fz_nz: conditional tests through pcor on the row view (recursive_pcor = false: FzTestCond without cor_mat, tests.jl:253)."""
        self._post.append(lambda h: self.L.fwo_fz_nz_set_stream(h, int(stream)))
        self._post[-1](self.h)

    def pcor(self, X, Y, Zs):
        z = np.asarray(Zs, dtype=np.int32)
        return float(self.L.fwo_pcor(self.h, X, Y, _ptr(z), len(z)))

    def pcor_rec(self, X, Y, Zs):
        z = np.asarray(Zs, dtype=np.int32)
        return float(self.L.fwo_pcor_rec(self.h, X, Y, _ptr(z), len(z)))

    def contingency_table(self, X, Y, Zs=(), nslots=None):
        z = np.asarray(Zs, dtype=np.int32)
        Lv = self.L.fwo_L(self.h)
        if nslots is None:
            nslots = max(1, Lv ** len(z))
        out = np.zeros(Lv * Lv * nslots, dtype=np.int64)
        lz = self.L.fwo_contingency_table(self.h, X, Y, _ptr(z), len(z), _ptr(out), nslots)
        return out.reshape((Lv, Lv, nslots), order="F"), lz

    def test_subsets(self, X, Y, Z_total, max_k=3, alpha=0.01, hps=5, n_obs_min=0, max_tests=10_000_000):
        zt = np.asarray(Z_total, dtype=np.int32)
        r = Result()
        zs = np.zeros(16, np.int32)
        nzs = C.c_int(0)
        nt = C.c_int64(0)
        fr = C.c_double(0)
        st = self.L.fwo_test_subsets(self.h, X, Y, _ptr(zt), len(zt), max_k, alpha, hps, n_obs_min, max_tests,
                                     C.byref(r), _ptr(zs), C.byref(nzs), C.byref(nt), C.byref(fr))
        return dict(status=st, stat=r.stat, pval=r.pval, df=int(r.df), suff_power=bool(r.suff_power),
                    Zs=tuple(int(v) for v in zs[:nzs.value]), num_tests=int(nt.value), frac=fr.value)

    def level0(self, alpha=0.01, hps=5, n_obs_min=0, FDR=True, correct_reliable_only=True):
        h = self.L.fwo_level0(self.h, alpha, hps, n_obs_min, int(FDR), int(correct_reliable_only))
        tot = self.L.fwo_nbrs_total(h, self.p)
        off = np.zeros(self.p + 1, np.int32)
        idx = np.zeros(max(tot, 1), np.int32)
        stat = np.zeros(max(tot, 1), np.float64)
        pval = np.zeros(max(tot, 1), np.float64)
        self.L.fwo_nbrs_copy(h, self.p, _ptr(off), _ptr(idx), _ptr(stat), _ptr(pval))
        nt = self.L.fwo_nbrs_ntests(h)
        self.L.fwo_nbrs_free(h)
        return dict(off=off, idx=idx[:tot], stat=stat[:tot], pval=pval[:tot], n_tests=int(nt))

    def level0_sample(self, hps=5, n_obs_min=0, x_start=0, x_stride=1, max_seconds=0.0):
        """bench.py cpu_baseline: time the level-0 pair tests of every x_stride-th row -> (n_tests, seconds)."""
        sec = C.c_double(0)
        nt = self.L.fwo_level0_sample(self.h, hps, n_obs_min, x_start, x_stride, max_seconds, C.byref(sec))
        return int(nt), sec.value

    def level0_rows(self, alpha=0.01, hps=5, n_obs_min=0, x_start=0, x_stride=1, max_rows=1 << 30, cap=1 << 22):
        """Raw level-0 results of sampled rows (every Y > X): the pairs with raw p < alpha -> dict(X, Y, stat, pval, n_tests, m)."""
        x = np.zeros(cap, np.int32)
        y = np.zeros(cap, np.int32)
        st = np.zeros(cap, np.float64)
        pv = np.zeros(cap, np.float64)
        nt, m = C.c_int64(0), C.c_int64(0)
        k = self.L.fwo_level0_rows(self.h, alpha, hps, n_obs_min, x_start, x_stride, max_rows, cap, _ptr(x), _ptr(y), _ptr(st), _ptr(pv),
                                   C.byref(nt), C.byref(m))
        if k < 0:
            raise ValueError("level0_rows: more than %d significant pairs in the sample" % cap)
        return dict(X=x[:k], Y=y[:k], stat=st[:k], pval=pv[:k], n_tests=int(nt.value), m=int(m.value))

    def learn(self, max_k=3, alpha=0.01, hps=5, n_obs_min=-1, max_tests=10_000_000, FDR=True, feed_forward=True,
              round_size=1, max_targets=0, target_stride=1, max_seconds=0.0, target_offset=0, nbrs=None, threads=1):
        """nbrs = dict(off, idx, stat, pval[, n_tests]): level-0 neighbour lists computed elsewhere (bench.py's
        cpu_baseline at sizes where a full CPU level-0 pass does not fit the bounded sample).
        threads > 1 (needs round_size > 1 or feed_forward = False): the targets between two whitelist snapshots run on a pool
        of threads, one context each over this context's arrays (fwo_learn_mt); same results as threads = 1."""
        P = Params(alpha, hps, n_obs_min, max_k, max_tests, int(FDR), int(feed_forward), round_size, max_targets,
                   target_stride, max_seconds, target_offset, 0.0)
        nb = None
        if nbrs is not None:
            o = np.ascontiguousarray(nbrs["off"], dtype=np.int64)
            i = np.ascontiguousarray(nbrs["idx"], dtype=np.int32)
            s_ = np.ascontiguousarray(nbrs["stat"], dtype=np.float64)
            q = np.ascontiguousarray(nbrs["pval"], dtype=np.float64)
            nb = self.L.fwo_nbrs_from_csr(self.p, _ptr(o), _ptr(i), _ptr(s_), _ptr(q), int(nbrs.get("n_tests", 0)))
        extra = []
        try:
            if threads > 1:
                extra = [self._mk() for _ in range(threads - 1)]
                for h in extra:
                    for f in self._post:
                        f(h)
                hs = (C.c_void_p * threads)(self.h, *extra)
                g = self.L.fwo_learn_mt(hs, threads, C.byref(P), nb)
            else:
                g = self.L.fwo_learn(self.h, C.byref(P), nb)
        finally:
            for h in extra:
                self.L.fwo_destroy(h)
            if nb:
                self.L.fwo_nbrs_free(nb)
        ne = self.L.fwo_network_nedges(g)
        if ne < 0:
            self.L.fwo_network_free(g)
            raise ValueError("Dataset has an insufficient number of observations (n_obs_min)")
        src = np.zeros(max(ne, 1), np.int32)
        dst = np.zeros(max(ne, 1), np.int32)
        w = np.zeros(max(ne, 1), np.float64)
        self.L.fwo_network_copy(g, _ptr(src), _ptr(dst), _ptr(w))
        n0, n1 = C.c_int64(0), C.c_int64(0)
        t0, t1 = C.c_double(0), C.c_double(0)
        ntg = C.c_int(0)
        self.L.fwo_network_stats(g, C.byref(n0), C.byref(n1), C.byref(t0), C.byref(t1), C.byref(ntg))
        tot = self.L.fwo_network_pc_total(g, self.p)
        off = np.zeros(self.p + 1, np.int32)
        idx = np.zeros(max(tot, 1), np.int32)
        st = np.zeros(max(tot, 1), np.float64)
        pv = np.zeros(max(tot, 1), np.float64)
        self.L.fwo_network_pc_copy(g, self.p, _ptr(off), _ptr(idx), _ptr(st), _ptr(pv))
        self.L.fwo_network_free(g)
        edges = {(int(a), int(b)): float(x) for a, b, x in zip(src[:ne], dst[:ne], w[:ne])}
        return dict(edges=edges, n_level0_tests=int(n0.value), n_cond_tests=int(n1.value), t_level0=t0.value,
                    t_cond=t1.value, n_targets=int(ntg.value), pc_off=off, pc_idx=idx[:tot], pc_weight=st[:tot],
                    pc_pval=pv[:tot])


def mutual_information(ctab):
    t = np.asfortranarray(np.asarray(ctab, dtype=np.int64))
    if t.ndim == 2:
        return float(lib().fwo_mutual_information(_ptr(t), t.shape[0], t.shape[1], 0))
    return float(lib().fwo_mutual_information(_ptr(t), t.shape[0], t.shape[1], t.shape[2]))


def mi_pval(mi, df, n_obs):
    return float(lib().fwo_mi_pval(mi, df, n_obs))


def fz_pval(stat, n, len_z):
    return float(lib().fwo_fz_pval(stat, n, len_z))


def benjamini_hochberg(pvals, alpha=0.01, m=None):
    pv = np.array(pvals, dtype=np.float64)
    lib().fwo_benjamini_hochberg(_ptr(pv), len(pv), alpha, len(pv) if m is None else m)
    return pv
