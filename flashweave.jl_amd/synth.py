"""fwsynth-v1: the synthetic count-matrix generator the benchmark configs are defined on.
The reference ships no generator; this file IS the contract.  Poisson-lognormal counts with AR(1) latent
chains inside blocks of 25 OTUs, a per-sample depth effect, 3 global latent factors (profile "confounded"),
and optionally 4 habitats with block-wise structural absences (HE configs).  Chunked by column block."""
import hashlib

import numpy as np

BLOCK, RHO, SIGMA, MU_SD, DEPTH_SD = 25, 0.9, 2.0, 1.0, 0.3


def generate(p, n, seed, profile="confounded", mode="S", habitats=0, n_meta=0, fac_frac=None):
    """-> counts int32 (n, p) [+ meta int32 (n, n_meta)].  mode 'S' (Fisher-z configs) / 'F' (discrete)."""
    n_fac = 0 if profile == "chain" else 3
    if fac_frac is None:
        fac_frac = min(0.3, 600.0 / p)
    fac_sd = 0.7 if mode == "S" else 2.0
    g_rng = np.random.Generator(np.random.PCG64([seed, 2**31 - 1]))
    d = g_rng.normal(0.0, DEPTH_SD, n)
    g = g_rng.standard_normal((n, n_fac))
    h = g_rng.integers(0, habitats, n) if habitats else None
    counts = np.zeros((n, p), dtype=np.int32)
    nblocks = (p + BLOCK - 1) // BLOCK
    for b in range(nblocks):
        w = min(BLOCK, p - b * BLOCK)
        rng = np.random.Generator(np.random.PCG64([seed, b]))
        eps = rng.standard_normal((n, w))
        mu = rng.normal(0.0, MU_SD, w)
        a = rng.normal(0.0, fac_sd, (n_fac, w)) * (rng.random((n_fac, w)) < fac_frac)
        present = None
        if habitats:
            present = rng.random(habitats) < 0.5
            if not present.any():
                present[0] = True
        z = np.empty((n, w))
        z[:, 0] = eps[:, 0]
        c = np.sqrt(1.0 - RHO * RHO)
        for j in range(1, w):
            z[:, j] = RHO * z[:, j - 1] + c * eps[:, j]
        lam = np.exp(mu[None, :] + SIGMA * z + g @ a + d[:, None])
        if habitats:
            lam[~present[h], :] = 0.0
        # saturate instead of wrapping: a heavy-tailed draw can exceed the int32 count range (seen once in ~600 seeds)
        counts[:, b * BLOCK:b * BLOCK + w] = np.minimum(rng.poisson(lam), 2**31 - 1).astype(np.int32)
    if n_meta:
        assert habitats == 4 and n_meta == 20
        meta = np.zeros((n, 20), dtype=np.int32)
        for q in range(4):
            meta[:, q] = h == q
        col = 4
        for q in range(4):
            for r in range(q + 1, 4):
                meta[:, col] = (h == q) | (h == r)
                col += 1
        for q in range(10):
            meta[:, col + q] = g_rng.random(n) < 0.5
        return counts, meta
    return counts


def checksum(counts):
    return hashlib.sha256(np.ascontiguousarray(counts, dtype=np.int32).tobytes()).hexdigest()


# BASELINE.json configs; seeds = 20260928 + config index
CONFIGS = {
    "cfg2": dict(p=1000, n=500, seed=20260930, mode="F", test_name="mi", max_k=3),
    "cfg3": dict(p=10000, n=2000, seed=20260931, mode="S", test_name="fz", max_k=3),
    "cfg4": dict(p=50000, n=5000, seed=20260932, mode="F", test_name="mi_nz", max_k=3, habitats=4, n_meta=20),
    "cfg5": dict(p=100000, n=10000, seed=20260933, mode="S", test_name="fz", max_k=5),
    # not a BASELINE config: FlashWeaveHE-S (fz_nz) on cfg3's shape with habitat-wise structural absences and the
    # 20 meta variables of the HE generator -- the bench / profile line of the zero-ignoring Fisher-z path
    "cfg3he": dict(p=3000, n=2000, seed=20260934, mode="S", test_name="fz_nz", max_k=3, habitats=4, n_meta=20),
}
