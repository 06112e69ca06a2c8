"""Host-side pieces every trainer shares: input coercion, the reference's initialisation and
shuffle (global legacy numpy state), progress bar, early-stopping bookkeeping."""
import threading

import numpy as np
from scipy import sparse

try:  # the reference shows a tqdm bar (cymf/bpr.pyx:159); optional here
    from tqdm import tqdm as _tqdm
except Exception:  # pragma: no cover
    _tqdm = None


def coerce_csr(X):
    """cymf/bpr.pyx:78-87 / cymf/wmf.pyx:69-78: None or unknown type -> ValueError."""
    if X is None:
        raise ValueError()
    if sparse.isspmatrix(X) or isinstance(X, sparse.sparray if hasattr(sparse, "sparray") else ()):
        X = sparse.csr_matrix(X)
    elif isinstance(X, np.ndarray):
        X = sparse.csr_matrix(X)
    else:
        raise ValueError()
    return X.astype(np.float64)


# The reference draws initial factors and the shuffle from numpy's GLOBAL legacy state.  fit() holds this lock while it
# does, so that several ranks running as threads of one process (tests with dist.Comm.local_group) each get the
# sequence a process of their own would get.
GLOBAL_RNG_LOCK = threading.RLock()


def init_factors(model, U, I, K):
    """cymf/bpr.pyx:97-101: seed 4321 only when W is None; H drawn from the continuing state."""
    if model.W is None:
        np.random.seed(4321)
        model.W = np.random.uniform(low=-0.1, high=0.1, size=(U, K)) / K
    if model.H is None:
        model.H = np.random.uniform(low=-0.1, high=0.1, size=(I, K)) / K
    # training is in place on C-contiguous float64 (cymf/bpr.pyx:127-128)
    model.W = np.ascontiguousarray(model.W, dtype=np.float64)
    model.H = np.ascontiguousarray(model.H, dtype=np.float64)


def reference_shuffle(*arrays):
    """sklearn.utils.shuffle(*arrays) with random_state=None (cymf/bpr.pyx:104): one permutation
    drawn with np.random.shuffle(arange(n)) from the GLOBAL legacy state."""
    idx = np.arange(len(arrays[0]))
    np.random.shuffle(idx)
    return tuple(np.asarray(a)[idx] for a in arrays)


def membership_pattern(X):
    """CSR pattern with explicit zeros dropped and sorted indices = the device form of the
    reference's `vector<set<int>> user_positives` (cymf/bpr.pyx:146-147)."""
    P = X.copy()
    P.eliminate_zeros()
    P.sum_duplicates()
    P.sort_indices()
    return P.indptr.astype(np.int32), P.indices.astype(np.int32)


class Progress:
    def __init__(self, total, verbose, ncols=120):
        self.bar = _tqdm(total=total, leave=True, ncols=ncols, disable=not verbose) if _tqdm else None

    def step(self, desc):
        if self.bar is not None:
            self.bar.set_description(desc)
            self.bar.update(1)

    def close(self):
        if self.bar is not None:
            self.bar.close()


class EpochChunks:
    """How many epochs the next call into the library runs.  Every call ends with a device synchronisation and the
    losses' way back; on a small problem (an ml-1m-sized BPR epoch is 0.3 ms of kernels) that is several times the
    work.  With a validation evaluator every epoch is its own call (the factors are downloaded after each), and so it is
    in a sharded job (the cut is timed, and every rank has to make the same number of calls); otherwise
    the calls double while they stay under `target` seconds, so that a progress bar still moves about every tenth of
    a second.  The epochs themselves are the same whichever way they are cut (one stream of negatives, one order)."""

    def __init__(self, total, every_epoch, target=0.05, cap=256):
        self.left, self.every, self.target, self.cap, self.n = int(total), bool(every_epoch), target, cap, 1
        self._t0 = None

    def _every(self, total):   # (tests: an epoch per call)
        self.left, self.every, self.target, self.cap, self.n, self._t0 = int(total), True, 0.0, 1, 1, None

    def __iter__(self):
        return self

    def __next__(self):
        import time
        now = time.perf_counter()
        if self._t0 is not None and not self.every and now - self._t0 < self.target:
            self.n = min(self.n * 2, self.cap)
        if self.left <= 0:
            raise StopIteration
        n = 1 if self.every else min(self.n, self.left)
        self.left -= n
        self._t0 = time.perf_counter()
        return n

    def stop(self):
        self.left = 0


class EarlyStopping:
    """cymf/bpr.pyx:173-183, including its quirks: break needs count > 10; the best snapshot is
    refreshed whenever the validation DCG@5 does not get worse than the best seen."""

    def __init__(self, model):
        self.model = model
        self.count = 0
        self.W_best = model.W.copy()
        self.H_best = model.H.copy()

    def update(self, valid_dcg):
        m = self.model
        if m.early_stopping and m.valid_dcg > valid_dcg and self.count > 10:
            return True
        elif m.early_stopping and m.valid_dcg > valid_dcg:
            self.count += 1
        else:
            self.count = 0
            m.valid_dcg = valid_dcg
            self.W_best = m.W.copy()
            self.H_best = m.H.copy()
        return False

    def finish(self):
        m = self.model
        if m.valid_evaluator and m.early_stopping:   # cymf/bpr.pyx:188-190
            m.W = self.W_best.copy()
            m.H = self.H_best.copy()


def pick_mode(mode, num_threads):
    """num_threads == 1 is the reference's only deterministic setting -> exact
    sequential-order mode; anything else is its HOGWILD regime -> throughput mode."""
    if mode is None:
        return "exact" if num_threads == 1 else "throughput"
    if mode not in ("exact", "throughput"):
        raise ValueError(f"mode must be 'exact' or 'throughput', got {mode!r}")
    return mode


def pick_dtype(dtype, mode):
    """Device arithmetic.  Default: float64 in exact mode (the parity path: the reference is
    float64 throughout, and the level launches are latency-bound, so double costs nothing there),
    float32 in throughput mode (the HBM-bound path)."""
    if dtype is None:
        return "float64" if mode == "exact" else "float32"
    if dtype not in ("float32", "float64"):
        raise ValueError(f"dtype must be 'float32' or 'float64', got {dtype!r}")
    return dtype
