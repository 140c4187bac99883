"""Host-side normalisation front-end: the three modes the BASELINE configs use.

Mirrors preprocess_data (reference src/preprocessing.jl:412-563) for a plain count matrix without meta
variables: variance / zero-read filters (:367-409), then
  "fz"    -> clr_adapt      adaptive pseudo-counts + centred log-ratio (:133-214)      -> Float32 dense
  "mi"    -> binary         presence/absence, columns with exactly 2 levels (:475-490) -> int {0,1}
  "mi_nz" -> binned_nz_clr  non-zero CLR, 2-bin median discretisation of the non-zeros (:492-521) -> int {0,1,2}
Computation is in Float64 like the reference (clrnorm converts to Matrix{Float64}, :330-341); `prec` only
selects the output type of the continuous mode (convert_to_target_prec, misc.jl:54-62).
"""
import numpy as np


def filter_by_variance(data):
    """preprocessing.jl:367-409: drop zero-variance columns, then all-zero rows."""
    col_mask = np.var(data, axis=0) > 0.0
    data = data[:, col_mask]
    row_mask = data.sum(axis=1) > 0
    return data[row_mask, :], row_mask, col_mask


def adaptive_clr(X):
    """adaptive_pseudocount! + clr!(pseudo_count=0) (preprocessing.jl:157-214)."""
    X = np.array(X, dtype=np.float64)
    max_depth_index = int(np.argmax(X.sum(axis=1)))
    s1 = X[max_depth_index]
    min_abund = X[X != 0].min()
    base_pcount = 1.0 if min_abund >= 1 else min_abund / 10
    k = int((s1 == 0).sum())
    Nprod1 = float(np.log(s1[s1 != 0]).sum())
    P = X.shape[1]
    pseudo = np.zeros(X.shape[0])
    for i in range(X.shape[0]):
        s2 = X[i]
        nz = int((s2 == 0).sum())
        Nprod2 = float(np.log(s2[s2 != 0]).sum())
        if not (nz < P and k < P):
            raise ValueError("samples with all zero abundances are not allowed")
        pseudo[i] = np.exp((1.0 / (nz - P)) * ((k - P) * np.log(base_pcount) + Nprod1 - Nprod2))
    keep = pseudo != 0
    X, pseudo = X[keep], pseudo[keep]
    for i in range(X.shape[0]):
        row = X[i]
        row[row == 0] = pseudo[i]
    gmean = np.exp(np.log(X).mean(axis=1, keepdims=True))  # StatsBase.geomean
    return np.log(X / gmean), keep


def clr_nz(X):
    """clr!(ignore_zeros=true) (preprocessing.jl:192-207): log(x / geomean of the row's non-zeros), zeros stay 0."""
    X = np.array(X, dtype=np.float64)
    out = np.zeros_like(X)
    for i in range(X.shape[0]):
        m = X[i] != 0
        if m.any():
            g = np.exp(np.log(X[i, m]).mean())
            out[i, m] = np.log(X[i, m] / g)
    return out


def _tiedrank(x):
    from scipy.stats import rankdata
    return rankdata(x, method="average")


def discretize_nz(col, nz_mask, n_bins=3):
    """discretize_nz + discretize (preprocessing.jl:238-291), disc_method = "median", rank_method = "tied"."""
    out = np.zeros(col.shape[0], dtype=np.int64)
    if nz_mask.any():
        r = _tiedrank(col[nz_mask])
        r = r / r.max()
        step = (1.0 / (n_bins - 1)) + 1e-5
        out[nz_mask] = np.floor(r / step).astype(np.int64) + 1
    return out


def normalize(counts, test_name, prec=32):
    """-> (data, row_mask, col_mask).  row/col masks refer to the input matrix."""
    counts = np.asarray(counts)
    data, row_mask, col_mask = filter_by_variance(counts.astype(np.float64))
    cols = np.nonzero(col_mask)[0]
    if test_name == "fz":
        out, keep = adaptive_clr(data)
        rows = np.nonzero(row_mask)[0]
        row_mask = np.zeros_like(row_mask)
        row_mask[rows[keep]] = True
        return out.astype(np.float32 if prec == 32 else np.float64), row_mask, col_mask
    if test_name == "mi":
        b = np.sign(data).astype(np.int64)
        lv = np.array([len(np.unique(b[:, j])) for j in range(b.shape[1])])
        km = lv == 2
        cm = np.zeros_like(col_mask)
        cm[cols[km]] = True
        return b[:, km], row_mask, cm
    if test_name == "mi_nz":
        nzm = data != 0
        c = clr_nz(data)
        d = np.stack([discretize_nz(c[:, j], nzm[:, j]) for j in range(c.shape[1])], axis=1)
        km = np.array([len(np.unique(d[d[:, j] != 0, j])) == 2 for j in range(d.shape[1])])
        cm = np.zeros_like(col_mask)
        cm[cols[km]] = True
        return d[:, km], row_mask, cm
    if test_name == "fz_nz":  # clr_nz (preprocessing.jl:335-342): zeros stay zeros (= absences)
        out = clr_nz(data)
        return out.astype(np.float32 if prec == 32 else np.float64), row_mask, col_mask
    raise ValueError("unsupported test_name %r" % (test_name,))
