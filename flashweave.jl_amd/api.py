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
    """Targets per feed-forward round when the caller does not choose.

    p <= 512: 1 -- the reference's deterministic `single_il` schedule (interleaved.jl:62-183; what its golden networks were
    generated with), so the default output of a small problem IS the reference's network.  Larger problems: rounds of
    min(1024 * ceil(p / 10240), max(64, ceil(p / 8))) targets -- eight to ten rounds per pass, so that whitelists exist at every
    size (one round would silently turn feed_forward off), each of them on the device (fz: device-resident rounds, discrete: one
    persistent launch).  At the benchmark sizes this is the schedule bench.py reports (cfg3 1024, cfg4 5120, cfg5 10 240).  The result records which schedule ran
    (`parameters["schedule"]`): rounds deviate from single_il in when the whitelists refresh, not in the tests themselves."""
    if p <= 512:
        return 1
    return min(1024 * ((p + 10239) // 10240), max(64, (p + 7) // 8))


def _integral(a):
    """Count table the device front-end accepts: integral values in 0 .. 2^31 - 1 (fw_normalize_counts takes Int32); anything else
    -- relative abundances, negative entries, counts beyond Int32 -- goes to the host front-end (preprocess.py)."""
    if not a.size:
        return False
    if np.issubdtype(a.dtype, np.integer):
        return bool(a.min() >= 0 and a.max() < 2**31)
    return bool(np.issubdtype(a.dtype, np.floating) and np.all(np.isfinite(a)) and np.all(a == np.floor(a)) and
                a.max() < 2**31 and a.min() >= 0)


def learn_network(data, sensitive=True, heterogeneous=False, max_k=3, alpha=0.01, feed_forward=True, normalize=True,
                  header=None, hps=5, FDR=True, n_obs_min=-1, max_tests=10_000_000, prec=32, round_size=None, device=0,
                  meta_data=None, meta_header=None, make_onehot=True, recursive_pcor=True, dense_cor=True, device_normalize=True, **unsupported):
    """data: samples x OTUs count matrix (or an already normalised matrix with normalize=False).
    meta_data: optional samples x meta-variables table (numbers and / or string factors), handled like the reference's
    meta_data_path input: one-hot encoding, discretisation for the discrete tests, +1 shift for fz_nz (preprocess.py).
    round_size: targets per feed-forward round.  None (default) = default_round_size(p): the reference's `single_il` schedule
    (1) up to 512 variables, eight to ten device rounds per pass beyond (the benchmarked configuration; the whitelists refresh
    once per round).  1 = `single_il` at any size (what the golden networks were generated with): every round is one target
    and runs through the host job pool -- exact reproduction of the reference's edge lists, and far slower on large tables.
    0 = one round (parallel="single").
    recursive_pcor / dense_cor (sensitive mode, learning.jl:42,127): recursive_pcor=False takes the conditional tests from the data
    instead of the Pearson matrix; dense_cor=False never builds the p x p matrix at all -- level 0 multiplies and screens the
    centred columns tile by tile (same network as dense_cor=True, memory bounded by the data).  As in the reference, dense_cor
    only matters for the plain "fz" test (the other tests never build a matrix: the flag is ignored there); without a matrix the
    conditional tests can only come from the data, so dense_cor=False implies recursive_pcor=False (a warning says so when the
    caller left recursive_pcor at its default).
    device_normalize: normalise integer count tables on the device (fw_normalize_counts; all four modes); False, or a table of
    non-integral abundances, takes the host front-end (preprocess.py)."""
    if unsupported:
        raise TypeError("learn_network: unsupported options %s (see DESIGN.md section 7)" % sorted(unsupported))
    import time
    test_name = ("fz" if sensitive else "mi") + ("_nz" if heterogeneous else "")  # src/learning.jl:480-483
    if test_name != "fz":
        dense_cor = True  # (learning.jl:42: only the plain fz test ever builds a matrix; the engine takes the flag for fz alone)
    elif not dense_cor and recursive_pcor:
        import warnings
        warnings.warn("learn_network: dense_cor=False leaves no correlation matrix for recursive partial correlations; "
                      "running with recursive_pcor=False (conditional tests from the data)", stacklevel=2)
        recursive_pcor = False
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
                                    feed_forward=feed_forward, test_name=test_name, round_size=round_size,
                                    recursive_pcor=recursive_pcor, dense_cor=dense_cor,
                                    schedule=("single_il (one target per round: the reference's deterministic schedule)" if round_size == 1
                                              else "one round (parallel=\"single\": no whitelists)" if (round_size == 0 or not feed_forward or round_size >= p)
                                              else "rounds of %d targets (whitelists refresh once per round; deviates from single_il)" % round_size)),
                    counters=counters)
