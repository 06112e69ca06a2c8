"""Synthetic *-shaped interaction data.

The reference's dataset loaders download MovieLens / text8 (cymf/dataset/movielens.py:31-40,
cymf/dataset/text8.py:33-48); there is no network here, so benchmarks and tests use data of
the same shape: user activity ~ lognormal(sigma=1), item popularity ~ Zipf(s=1) over a random
item permutation, (u, i) pairs sampled, de-duplicated, value 1.0.
"""
import numpy as np
from scipy import sparse

# name -> (U, I, target nnz, K, seed)   
CONFIGS = {
    "C1": (943, 1682, 44853, 20, 100),            # ml-100k-shaped
    "C2": (6040, 3706, 465977, 64, 101),          # ml-1m-shaped
    "C3": (1000000, 100000, 100000000, 128, 102), # 1M x 100k, headline
    "C4": (138493, 26744, 20000263, 64, 103),     # ml-20m-shaped (WMF)
    "C5": (71290, 71290, 30000000, 100, 104),     # text8-shaped co-occurrence (GloVe)
}


def implicit_matrix(U, I, nnz, seed, zipf_s=1.0, sigma=1.0, oversample=1.08, max_rounds=40):
    """CSR float64 (U, I) with ~nnz ones (exactly nnz when reachable), sorted indices."""
    rng = np.random.default_rng(seed)
    act = rng.lognormal(mean=0.0, sigma=sigma, size=U)
    act /= act.sum()
    pop = 1.0 / np.arange(1, I + 1, dtype=np.float64) ** zipf_s
    pop /= pop.sum()
    perm = rng.permutation(I)
    cdf_i = np.cumsum(pop)
    cdf_i[-1] = 1.0
    cdf_u = np.cumsum(act)
    cdf_u[-1] = 1.0
    keys = np.empty(0, dtype=np.int64)
    need = nnz
    for _ in range(max_rounds):
        m = int(need * oversample) + 16
        u = np.searchsorted(cdf_u, rng.random(m), side="right").astype(np.int64)
        i = perm[np.searchsorted(cdf_i, rng.random(m), side="right")].astype(np.int64)
        keys = np.unique(np.concatenate([keys, u * I + i]))
        if len(keys) >= nnz:
            break
        need = nnz - len(keys)
        oversample = max(oversample, 2.0)
    if len(keys) > nnz:  # drop a random subset to hit the target exactly
        drop = rng.choice(len(keys), size=len(keys) - nnz, replace=False)
        mask = np.ones(len(keys), dtype=bool)
        mask[drop] = False
        keys = keys[mask]
    rows = (keys // I).astype(np.int32)
    cols = (keys % I).astype(np.int32)
    indptr = np.zeros(U + 1, dtype=np.int64)
    indptr[1:] = np.bincount(rows, minlength=U)
    indptr = np.cumsum(indptr).astype(np.int32 if len(keys) < 2**31 else np.int64)
    X = sparse.csr_matrix((np.ones(len(keys), dtype=np.float64), cols, indptr), shape=(U, I))
    X.has_sorted_indices = True
    return X


def implicit_matrix_large(U, I, nnz, seed, zipf_s=1.0, sigma=1.0, oversample=1.06):
    """Same distribution as implicit_matrix, organised for 1e7..1e9 interactions: per-user counts
    are drawn first (multinomial over the lognormal activities), so the (user, item) keys are
    generated already grouped by user and one sort of the int64 keys de-duplicates them.
    Returns ~nnz interactions (a little fewer than requested after de-duplication is possible)."""
    rng = np.random.default_rng(seed)
    act = rng.lognormal(mean=0.0, sigma=sigma, size=U)
    act /= act.sum()
    pop = 1.0 / np.arange(1, I + 1, dtype=np.float64) ** zipf_s
    cdf_i = np.cumsum(pop / pop.sum())
    cdf_i[-1] = 1.0
    perm = rng.permutation(I).astype(np.int64)
    keys = np.empty(0, dtype=np.int64)
    need = nnz
    for _ in range(8):
        m = int(need * oversample) + 1024
        n_u = np.minimum(rng.multinomial(m, act), I)
        new = np.repeat(np.arange(U, dtype=np.int64) * I, n_u)
        block = 1 << 24
        for b in range(0, len(new), block):     # bounded temporaries
            e = min(b + block, len(new))
            new[b:e] += perm[np.searchsorted(cdf_i, rng.random(e - b), side="right")]
        keys = np.concatenate([keys, new]) if len(keys) else new
        del new
        keys.sort()
        keep = np.ones(len(keys), dtype=bool)
        keep[1:] = keys[1:] != keys[:-1]
        keys = keys[keep]
        if len(keys) >= nnz:
            break
        need = nnz - len(keys)
        oversample = max(oversample, 1.6)
    if len(keys) > nnz:
        drop = rng.choice(len(keys), size=len(keys) - nnz, replace=False)
        keep = np.ones(len(keys), dtype=bool)
        keep[drop] = False
        keys = keys[keep]
    rows = (keys // I).astype(np.int32)
    cols = (keys % I).astype(np.int32)
    del keys
    indptr = np.zeros(U + 1, dtype=np.int64)
    indptr[1:] = np.bincount(rows, minlength=U)
    indptr = np.cumsum(indptr)
    return rows, cols, indptr


def config_matrix(name):
    U, I, nnz, K, seed = CONFIGS[name]
    return implicit_matrix(U, I, nnz, seed), K


def cooccurrence_matrix(V, nnz, seed):
    """text8-shaped co-occurrence: pattern as above, counts ~ lognormal clipped to [0.1, 1e4]."""
    X = implicit_matrix(V, V, nnz, seed)
    rng = np.random.default_rng(seed + 7)
    X.data[:] = np.clip(rng.lognormal(mean=0.5, sigma=1.5, size=X.nnz), 0.1, 1e4)
    return X
