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


def adaptive_clr(counts):
    """clr_adapt for a dense count matrix (what preprocessing.jl:157-214 computes): every sample's zeros are replaced by ONE fill value
    chosen so that the filled sample is as far from its own geometric mean as the deepest sample is when ITS zeros hold the floor
    value -- in logs,  fill_i = exp( [ (z* - w) log(floor) + L* - L_i ] / (z_i - w) )  with z = zeros of the sample, L = sum of the logs
    of its non-zero counts, w = the table's width, * = the sample with the largest total -- followed by the centred log-ratio.
    Returns (matrix, kept-samples mask); a sample whose fill value underflows to zero is dropped."""
    M = np.array(counts, dtype=np.float64)
    n_samples, width = M.shape
    present = M != 0
    zeros = width - present.sum(axis=1)
    if (zeros >= width).any():
        raise ValueError("samples with all zero abundances are not allowed")
    log_mass = np.array([np.log(M[i, present[i]]).sum() for i in range(n_samples)])
    deepest = int(np.argmax(M.sum(axis=1)))
    smallest = M[present].min()
    floor = 1.0 if smallest >= 1 else smallest / 10
    anchor = (zeros[deepest] - width) * np.log(floor) + log_mass[deepest]
    fill = np.exp((1.0 / (zeros - width)) * (anchor - log_mass))
    kept = fill != 0
    M, present, fill = M[kept], present[kept], fill[kept]
    M = np.where(present, M, fill[:, None])
    centre = np.exp(np.log(M).mean(axis=1, keepdims=True))  # geometric mean of the filled sample
    return np.log(M / centre), kept


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


# ---- meta variables (preprocessing.jl:42-117 one-hot, :293-316 discretize_meta!, :527-555) ---------------------------------
def _is_number(v):
    return isinstance(v, (int, float, np.integer, np.floating)) and not isinstance(v, bool)


def onehot(meta, header=None):
    """onehot(X, vnames) (preprocessing.jl:42-117): a string factor with more than two categories becomes one 0/1 dummy
    column per category (sorted, named <var>_<category>); a string factor with one or two categories becomes the integers
    1, 2 (factors_to_ints); numeric columns pass through.  -> (Float64 matrix, header)"""
    meta = np.asarray(meta, dtype=object)
    cols, names = [], []
    for j in range(meta.shape[1]):
        x = meta[:, j]
        name = header[j] if header is not None else ""
        if _is_number(x[0]):
            cols.append(np.array([float(v) for v in x]))
            names.append(name)
            continue
        cats = sorted(set(x))
        if len(cats) > 2:
            for cat in cats:
                cols.append(np.array([1.0 if v == cat else 0.0 for v in x]))
                names.append("%s_%s" % (name, cat) if name else "")
        else:
            fmap = {c: float(i + 1) for i, c in enumerate(cats)}
            cols.append(np.array([fmap[v] for v in x]))
            names.append(name)
    return np.stack(cols, axis=1), names


def is_continuous_vec(x):
    """iscontinuous(x_vec) (preprocessing.jl:295-302)."""
    if np.allclose(np.round(x), x):
        return bool(x.max() > 1 or len(np.unique(x)) > 2)
    return True


def discretize(x, n_bins):
    """discretize(x_vec, n_bins), disc_method = "median", rank_method = "tied" (preprocessing.jl:238-265)."""
    if len(x) == 0:
        return x
    r = _tiedrank(x)
    r = r / r.max()
    step = (1.0 / n_bins) + 1e-5
    return np.floor(r / step)


def normalize_with_meta(counts, test_name, meta, prec=32, header=None, meta_header=None, make_onehot=True, normalizer=None):
    """preprocess_data with a meta_mask (preprocessing.jl:412-563): OTU columns are normalised as in normalize(); meta
    variables are one-hot encoded, follow the row filters, are discretised into 2 bins for the discrete tests when they look
    continuous, are shifted by +1 for "fz_nz" if they hold zeros (zeros mean "absent" there), lose zero-variance columns
    and are appended.  -> dict(data, header, meta_mask, row_mask)"""
    if make_onehot:
        md, mh = onehot(meta, meta_header)
    else:
        md, mh = np.asarray(meta, dtype=np.float64), list(meta_header or [""] * np.asarray(meta).shape[1])
    # normalizer: the OTU part on the device (engine.normalize_counts); the handful of meta columns stay here
    data, row_mask, col_mask = normalizer(counts, test_name) if normalizer is not None else normalize(counts, test_name, prec=prec)
    md = md[row_mask]
    if test_name in ("mi", "mi_nz"):
        for j in range(md.shape[1]):
            if is_continuous_vec(md[:, j]):
                md[:, j] = discretize(md[:, j], 2)
    if test_name == "fz_nz":
        for j in range(md.shape[1]):
            if (md[:, j] == 0).any():
                md[:, j] += 1
    keep = np.var(md, axis=0) > 0.0
    md, mh = md[:, keep], [h for h, k in zip(mh, keep) if k]
    out = np.concatenate([data, md.astype(data.dtype)], axis=1)
    hdr = None
    if header is not None:
        hdr = [h for h, k in zip(header, col_mask) if k] + mh
    return dict(data=out, header=hdr, meta_header=mh, meta_mask=np.r_[np.zeros(data.shape[1], bool), np.ones(md.shape[1], bool)],
                row_mask=row_mask)


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
