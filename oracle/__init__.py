"""oracle -- CPU restatement of the reference's hot path.  TEST INFRASTRUCTURE ONLY.

Importers allowed: tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg.
The product package (cymf_amd/) must never import this module.

`oracle.lib()` returns a ctypes handle on oracle/libcymf_oracle.so (built from
oracle/cymf_oracle.c with gcc); the helpers below wrap it with numpy arrays.
Each C function cites the reference lines it restates.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libcymf_oracle.so")
_SRC = os.path.join(_HERE, "cymf_oracle.c")
_lib = None

OPT_IDS = {"sgd": 0, "adagrad": 1, "adam": 2}


# the CPU baseline is built `-O3 -fopenmp -ffp-contract=off` (no FMA contraction: the reference's
# x86-64 build has none)
CFLAGS = ["-O3", "-ffp-contract=off", "-fopenmp", "-std=c99", "-fPIC", "-shared", "-Wall"]
# SURVEY.md section 5 (race detection / sanitizers): the same source under AddressSanitizer + UBSan, for tests only
SAN_FLAGS = ["-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
             "-ffp-contract=off", "-fopenmp", "-std=c99", "-fPIC", "-shared", "-Wall"]
_SAN_SO = os.path.join(_HERE, "libcymf_oracle_san.so")


def _build_one(so, flags, force):
    stamp = so + ".flags"
    want = " ".join(flags)
    have = open(stamp).read() if os.path.exists(stamp) else None
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(_SRC) or have != want:
        subprocess.check_call(["gcc"] + flags + [_SRC, "-o", so, "-lm"])
        with open(stamp, "w") as f:
            f.write(want)
    return so


def build(force: bool = False) -> str:
    return _build_one(_SO, CFLAGS, force)


def build_sanitized(force: bool = False) -> str:
    """libcymf_oracle_san.so: cymf_oracle.c under -fsanitize=address,undefined.  Load it in a process started with
    LD_PRELOAD=<libasan> and CYMF_ORACLE_SO pointing at it (tests/test_oracle_sanitized.py)."""
    return _build_one(_SAN_SO, SAN_FLAGS, force)


def lib():
    global _lib
    if _lib is None:
        so = os.environ.get("CYMF_ORACLE_SO")      # the sanitizer build, tests only
        if not so:
            so = build()
        L = C.CDLL(so)
        vp, i32, i64, u32, u64, f64 = C.c_void_p, C.c_int32, C.c_int64, C.c_uint32, C.c_uint64, C.c_double
        L.orc_rng_fill_uniform.argtypes = [u32, u64, i64, i64, vp]
        L.orc_rng_fill_uniform.restype = None
        L.orc_rng_fill_raw.argtypes = [u32, i64, vp]
        L.orc_rng_fill_raw.restype = None
        L.orc_bpr_create.argtypes = [i32, i32, i32, C.c_int, f64, f64, u32, vp, vp]
        L.orc_bpr_create.restype = vp
        L.orc_bpr_destroy.argtypes = [vp]
        L.orc_bpr_skipped.argtypes = [vp]
        L.orc_bpr_skipped.restype = i64
        L.orc_bpr_epoch.argtypes = [vp, vp, vp, i64, vp, vp, vp]
        L.orc_bpr_epoch.restype = f64
        L.orc_bpr_apply.argtypes = [vp, vp, vp, vp, i64]
        L.orc_bpr_apply.restype = f64
        L.orc_bpr_epoch_hogwild.argtypes = [vp, vp, vp, i64, vp, vp, C.c_int, vp]
        L.orc_bpr_epoch_hogwild.restype = f64
        L.orc_max_threads.argtypes = []
        L.orc_max_threads.restype = C.c_int
        L.orc_relmf_create.argtypes = [i32, i32, i32, C.c_int, f64, f64, f64, u32, vp, vp]
        L.orc_relmf_create.restype = vp
        L.orc_relmf_destroy.argtypes = [vp]
        L.orc_relmf_epoch.argtypes = [vp, vp, vp, vp]
        L.orc_relmf_epoch.restype = f64
        L.orc_glove_create.argtypes = [i32, i32, i32, f64, f64, f64, vp, vp, vp, vp]
        L.orc_glove_create.restype = vp
        L.orc_glove_destroy.argtypes = [vp]
        L.orc_glove_epoch.argtypes = [vp, vp, vp, vp, i64]
        L.orc_glove_epoch.restype = f64
        L.orc_wmf_half_sweep.argtypes = [i32, i32, i32, vp, vp, vp, vp, f64, f64]
        L.orc_wmf_half_sweep.restype = None
        for fn in (L.orc_dcg_at_k, L.orc_recall_at_k, L.orc_ap_at_k):
            fn.argtypes = [vp, C.c_int, C.c_int]
            fn.restype = f64
        _lib = L
    return _lib


def max_threads():
    return int(lib().orc_max_threads())


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _c(a, dtype):
    a = np.ascontiguousarray(a, dtype=dtype)
    return a


# --------------------------------------------------------------------------- RNG
def uniform_stream(seed: int, rng_range: int, n: int, skip: int = 0) -> np.ndarray:
    """Draws [skip, skip+n) of UniformGenerator(0, rng_range, seed) (cymf/math.pyx:12-18)."""
    out = np.empty(n, dtype=np.int64)
    lib().orc_rng_fill_uniform(seed, rng_range, n, skip, _p(out))
    return out


def raw_stream(seed: int, n: int) -> np.ndarray:
    out = np.empty(n, dtype=np.uint32)
    lib().orc_rng_fill_raw(seed, n, _p(out))
    return out


# --------------------------------------------------------------------------- BPR
class Bpr:
    """Sequential fp64 BPR trainer on borrowed W/H (in place), cymf/bpr.pyx:117-171."""

    def __init__(self, W, H, optimizer="adam", lr=0.001, wd=0.01, neg_seed=1234):
        assert W.dtype == np.float64 and H.dtype == np.float64
        assert W.flags.c_contiguous and H.flags.c_contiguous
        self.W, self.H = W, H
        self.h = lib().orc_bpr_create(W.shape[0], H.shape[0], W.shape[1], OPT_IDS[optimizer], lr, wd,
                                      neg_seed, _p(W), _p(H))
        assert self.h

    def epoch(self, users, positives, indptr, indices, want_negatives=False):
        users, positives = _c(users, np.int32), _c(positives, np.int32)
        indptr, indices = _c(indptr, np.int32), _c(indices, np.int32)
        neg = np.empty(len(users), dtype=np.int32) if want_negatives else None
        loss = lib().orc_bpr_epoch(self.h, _p(users), _p(positives), len(users), _p(indptr), _p(indices), _p(neg))
        return (loss, neg) if want_negatives else loss

    def epoch_hogwild(self, users, positives, indptr, indices, n_threads):
        """All-thread lock-free epoch (cpu_baseline leg); returns (mean loss, performed)."""
        users, positives = _c(users, np.int32), _c(positives, np.int32)
        indptr, indices = _c(indptr, np.int32), _c(indices, np.int32)
        done = C.c_int64(0)
        loss = lib().orc_bpr_epoch_hogwild(self.h, _p(users), _p(positives), len(users), _p(indptr), _p(indices),
                                           int(n_threads), C.byref(done))
        return loss, done.value

    def apply(self, u, i, j):
        u, i, j = _c(u, np.int32), _c(i, np.int32), _c(j, np.int32)
        return lib().orc_bpr_apply(self.h, _p(u), _p(i), _p(j), len(u))

    @property
    def skipped(self):
        return lib().orc_bpr_skipped(self.h)

    def close(self):
        if self.h:
            lib().orc_bpr_destroy(self.h)
            self.h = None

    __del__ = close


# --------------------------------------------------------------------------- RelMF
class RelMf:
    def __init__(self, W, H, optimizer="adam", lr=0.001, wd=0.01, clip=0.1, seed=1234):
        assert W.dtype == np.float64 and H.dtype == np.float64
        self.W, self.H = W, H
        self.h = lib().orc_relmf_create(W.shape[0], H.shape[0], W.shape[1], OPT_IDS[optimizer], lr, wd,
                                        clip, seed, _p(W), _p(H))

    def epoch(self, X, propensities, want_draws=False):
        X = _c(X, np.float64)
        p = _c(propensities, np.float64)
        d = np.empty(X.size, dtype=np.int64) if want_draws else None
        loss = lib().orc_relmf_epoch(self.h, _p(X), _p(p), _p(d))
        return (loss, d) if want_draws else loss

    def close(self):
        if self.h:
            lib().orc_relmf_destroy(self.h)
            self.h = None

    __del__ = close


# --------------------------------------------------------------------------- GloVe
class Glove:
    def __init__(self, W, bW, H, bH, lr=0.01, x_max=10.0, alpha=0.75):
        for a in (W, bW, H, bH):
            assert a.dtype == np.float64 and a.flags.c_contiguous
        self.W, self.bW, self.H, self.bH = W, bW, H, bH
        self.h = lib().orc_glove_create(W.shape[0], H.shape[0], W.shape[1], lr, x_max, alpha,
                                        _p(W), _p(bW), _p(H), _p(bH))

    def epoch(self, central, context, counts):
        central, context = _c(central, np.int32), _c(context, np.int32)
        counts = _c(counts, np.float64)
        return lib().orc_glove_epoch(self.h, _p(central), _p(context), _p(counts), len(central))

    def close(self):
        if self.h:
            lib().orc_glove_destroy(self.h)
            self.h = None

    __del__ = close


# --------------------------------------------------------------------------- WMF
def wmf_half_sweep(indptr, indices, X, Y, weight, weight_decay):
    """In-place update of X (rows,K) given fixed Y (cols,K): cymf/wmf.pyx:136-174."""
    assert X.dtype == np.float64 and Y.dtype == np.float64 and X.flags.c_contiguous and Y.flags.c_contiguous
    indptr, indices = _c(indptr, np.int32), _c(indices, np.int32)
    lib().orc_wmf_half_sweep(X.shape[0], Y.shape[0], X.shape[1], _p(indptr), _p(indices), _p(X), _p(Y),
                             weight, weight_decay)


def wmf_fit(Xcsr, W, H, num_epochs, weight=10.0, weight_decay=0.01):
    """cymf/wmf.pyx:110-112: user half-sweep then item half-sweep per epoch."""
    Xt = Xcsr.T.tocsr()
    for _ in range(num_epochs):
        wmf_half_sweep(Xcsr.indptr, Xcsr.indices, W, H, weight, weight_decay)
        wmf_half_sweep(Xt.indptr, Xt.indices, H, W, weight, weight_decay)


# --------------------------------------------------------------------------- metrics
def dcg_at_k(y, k):
    y = _c(y, np.int32)
    return lib().orc_dcg_at_k(_p(y), len(y), k)


def recall_at_k(y, k):
    y = _c(y, np.int32)
    return lib().orc_recall_at_k(_p(y), len(y), k)


def ap_at_k(y, k):
    y = _c(y, np.int32)
    return lib().orc_ap_at_k(_p(y), len(y), k)


# --------------------------------------------------------------------------- host-side pieces the fits share
def reference_init(U, I, K):
    """np.random.seed(4321); W, H ~ U(-0.1,0.1)/K (cymf/bpr.pyx:97-101). Leaves the global
    numpy state where the reference leaves it, so a following shuffle matches bpr.pyx:104."""
    np.random.seed(4321)
    W = np.random.uniform(low=-0.1, high=0.1, size=(U, K)) / K
    H = np.random.uniform(low=-0.1, high=0.1, size=(I, K)) / K
    return W, H


def reference_shuffle(*arrays):
    """sklearn.utils.shuffle(*arrays) with random_state=None == one np.random.shuffle of arange
    drawn from the global legacy state (cymf/bpr.pyx:104)."""
    idx = np.arange(len(arrays[0]))
    np.random.shuffle(idx)
    return tuple(a[idx] for a in arrays)


def bpr_fit(Xcsr, K, optimizer, lr, wd, num_epochs, W=None, H=None):
    """End-to-end restatement of BPR.fit at num_threads=1 (cymf/bpr.pyx:68-171)."""
    Xcsr = Xcsr.tocsr().astype(np.float64)
    U, I = Xcsr.shape
    pat = Xcsr.copy()          # membership pattern: explicit zeros dropped, indices sorted
    pat.eliminate_zeros()
    pat.sort_indices()
    if W is None or H is None:
        W0, H0 = reference_init(U, I, K)
        W = W0 if W is None else W
        H = H0 if H is None else H
    users, positives = reference_shuffle(*Xcsr.nonzero())
    m = Bpr(W, H, optimizer, lr, wd)
    losses = [m.epoch(users, positives, pat.indptr, pat.indices) for _ in range(num_epochs)]
    m.close()
    return W, H, losses


# --------------------------------------------------------------------------- ExpoMF (numpy restatement)
def expomf_fit(Xcsr, W, H, num_epochs, lam_y=1.0, weight_decay=0.01):
    """cymf/expomf.pyx:105-207 in numpy, in place on W and H: E-step (:141-144, with the reference's
    sqrt(lam_y / 2.0 * pi)), user solves, item solves with the updated W (:146-147, np.linalg.solve is
    the dgesv of solvep), mu (:149).  PARITY UNPINNED: expomf.pyx does not build here (cblas)."""
    from scipy import sparse
    X = sparse.csr_matrix(Xcsr).astype(np.float64)
    X.eliminate_zeros()
    U, I = X.shape
    K = W.shape[1]
    mu = np.ones(I) * 0.01
    rows, cols = X.nonzero()

    def als(P, E, Xf, Yf):
        ridge = (weight_decay / lam_y) * np.eye(K)
        for i in range(P.shape[0]):
            idx = P.indices[P.indptr[i]:P.indptr[i + 1]]
            if len(idx) == 0:
                Xf[i] = 0.0
                continue
            b = (Yf[idx] * (E[i, idx] * lam_y)[:, None]).sum(axis=0)
            A = ridge + (Yf * (E[i] * lam_y)[:, None]).T @ Yf
            Xf[i] = np.linalg.solve(A, b)

    Xt = X.T.tocsr()
    for _ in range(num_epochs):
        n_ui = np.sqrt(lam_y / 2.0 * np.pi) * np.exp(-lam_y * np.dot(W, H.T) ** 2 / 2.0)
        E = (n_ui + 1e-8) / (n_ui + 1e-8 + (1 - mu) / mu)
        E[rows, cols] = 1.0
        als(X, E, W, H)
        als(Xt, E.T, H, W)
        mu = (1.0 + E.sum(axis=0) - 1.0) / (1.0 + 1.0 + U - 2.0)
    return mu
