"""`learn_network`: the reference's user entry point (src/learning.jl:466-598) for the modes this engine covers, as a thin
composition of the normalisation front-end, the device engine and the host driver.  Keyword names and defaults
follow the reference; unsupported options raise instead of being silently ignored."""
import numpy as np

from . import io as fio
from . import preprocess as pre
from .engine import Engine, normalize_counts


class FWResult(dict):
    """Edge list + bookkeeping (the reference's FWResult, src/types.jl:172-200, reduced to plain data)."""

    def save(self, path):
        """save_network (src/io.jl:300-336): .edgelist or .gml by extension."""
        fio.save_network(path, self["edges"], self["variable_ids"], self["meta_variable_mask"])


def default_round_size(p):
    """Targets per feed-forward round when the caller does not choose: 1024 * ceil(p / 10240), about ten rounds per pass -- the
    schedule bench.py reports.  Rounds of this size run as device-resident rounds (fz) / one persistent launch (discrete)."""
    return 1024 * ((p + 10239) // 10240)


def _integral(a):
    return np.issubdtype(a.dtype, np.integer) or (np.issubdtype(a.dtype, np.floating) and np.all(np.isfinite(a)) and
                                                   np.all(a == np.floor(a)) and a.size and a.max() < 2**31 and a.min() >= 0)


def learn_network(data, sensitive=True, heterogeneous=False, max_k=3, alpha=0.01, feed_forward=True, normalize=True,
                  header=None, hps=5, FDR=True, n_obs_min=-1, max_tests=10_000_000, prec=32, round_size=None, device=0,
                  meta_data=None, meta_header=None, make_onehot=True, recursive_pcor=True, dense_cor=True, device_normalize=True, **unsupported):
    """data: samples x OTUs count matrix (or an already normalised matrix with normalize=False).
    meta_data: optional samples x meta-variables table (numbers and / or string factors), handled like the reference's
    meta_data_path input: one-hot encoding, discretisation for the discrete tests, +1 shift for fz_nz (preprocess.py).
    round_size: targets per feed-forward round.  None (default) = default_round_size(p): about ten rounds per pass, every one of
    them on the device (the benchmarked configuration); the whitelists refresh once per round.  1 = the reference's deterministic
    `single_il` schedule (what the golden networks were generated with): every round is one target and runs through the host
    job pool -- exact reproduction of the reference's edge lists, and far slower.  0 = one round (parallel="single").
    recursive_pcor / dense_cor (sensitive mode, learning.jl:42,127): recursive_pcor=False takes the conditional tests from the data
    instead of the Pearson matrix; dense_cor=False (needs recursive_pcor=False) never builds the p x p matrix at all -- level 0
    multiplies and screens the centred columns tile by tile (same network as dense_cor=True, memory bounded by the data).
    device_normalize: normalise integer count tables on the device (fw_normalize_counts; all four modes); False, or a table of
    non-integral abundances, takes the host front-end (preprocess.py)."""
    if unsupported:
        raise TypeError("learn_network: unsupported options %s (see DESIGN.md section 7)" % sorted(unsupported))
    import time
    test_name = ("fz" if sensitive else "mi") + ("_nz" if heterogeneous else "")  # src/learning.jl:480-483
    data = np.asarray(data)
    if header is None:
        header = ["X%d" % (i + 1) for i in range(data.shape[1])]
    meta_mask = None
    if meta_data is not None and not normalize:
        # the reference appends the meta columns as they are and keeps their mask (learning.jl:500-520); here an already
        # normalised matrix must already hold them -- silently dropping the argument would lose the mask
        raise ValueError("learn_network: meta_data with normalize=False is not supported: append the prepared meta columns to "
                         "`data` yourself (preprocess.normalize_with_meta) or pass normalize=True")
    t_norm0 = time.perf_counter()
    on_device = bool(normalize and device_normalize and prec == 32 and _integral(data))
    if on_device and test_name == "mi_nz" and data.shape[0] > 16384:
        on_device = False  # binned_nz_clr sorts a column's non-zeros in LDS (fw_norm.hip: at most 16 384 samples): host front-end
    dev_norm = (lambda c, t: normalize_counts(c, t, device=device)) if on_device else None
    if normalize and meta_data is not None:
        r = pre.normalize_with_meta(data, test_name, meta_data, prec=prec, header=header, meta_header=meta_header,
                                    make_onehot=make_onehot, normalizer=dev_norm)
        mat, header, meta_mask = r["data"], r["header"], [bool(v) for v in r["meta_mask"]]
    elif normalize:
        mat, row_mask, col_mask = dev_norm(data, test_name) if on_device else pre.normalize(data, test_name, prec=prec)
        header = [h for h, k in zip(header, col_mask) if k]
    else:
        mat = data
    t_norm = time.perf_counter() - t_norm0
    n, p = mat.shape
    if round_size is None:
        round_size = default_round_size(p)
    eng = Engine(test_name, n, p, max_k=max_k, alpha=alpha, hps=hps, n_obs_min=n_obs_min, max_tests=max_tests, FDR=FDR,
                 device=device, recursive_pcor=recursive_pcor, dense_cor=dense_cor)
    try:
        eng.set_data(mat)
        if test_name == "fz" and dense_cor:
            eng.compute_cor()
        net = eng.lgl(feed_forward=feed_forward, round_size=round_size)
        counters = eng.counters()
    finally:
        eng.close()
    counters["t_normalize_s"] = t_norm
    counters["normalized_on_device"] = on_device
    return FWResult(edges=net["edges"], variable_ids=header, meta_variable_mask=meta_mask or [False] * len(header),
                    parameters=dict(sensitive=sensitive, heterogeneous=heterogeneous, max_k=max_k, alpha=alpha,
                                    feed_forward=feed_forward, test_name=test_name, round_size=round_size), counters=counters)
