"""Sampled-negative ranking evaluation (cymf/evaluator.pyx:34-149): the harness that produces
DCG/Recall/MAP@k between epochs, on the GPU (csrc/eval.hip) behind the reference's class surface.
Candidate sampling consumes the same mt19937 index stream as the reference's UniformGenerator.
Quirks kept: negatives may repeat, the mean runs over ALL users including those
without test items, the IPS metrics index the propensities by candidate position."""
import ctypes as C

import numpy as np
from scipy import sparse

from . import _lib

_METRIC_ROW = {"DCG": 0, "Recall": 1, "MAP": 2}


class Evaluator(object):
    def __init__(self, X, X_train=None, metrics=["DCG", "Recall", "MAP"], k=5, num_negatives=100, unbiased=False,
                 device=0):
        self.X = sparse.csr_matrix(X)
        self.user_positives = self.X.copy()
        if X_train is not None:
            self.user_positives = self.user_positives + sparse.csr_matrix(X_train)
        self.X = self.X.astype(np.float64)
        self.user_positives = sparse.csr_matrix(self.user_positives).astype(np.float64)
        self.propensity_scores = np.maximum(np.asarray(sparse.csr_matrix(X).mean(axis=0)).flatten(), 1e-4)
        for m in metrics:
            if m not in _METRIC_ROW:
                raise KeyError(m)
        self.metrics = metrics
        self.k = k
        self.num_negatives = num_negatives
        self.unbiased = unbiased
        self.device = device
        self._h = None

    def _handle(self):
        if self._h is None:
            L = _lib.lib()
            U, I = self.X.shape
            allp = self.user_positives.copy()
            allp.sum_duplicates()
            allp.sort_indices()
            h = C.c_void_p()
            prop = _lib.f64c(self.propensity_scores)
            _lib.check(L.cymf_eval_create(C.byref(h), U, I, _lib.ptr(_lib.i32c(self.X.indptr)), _lib.ptr(_lib.i32c(self.X.indices)),
                                          _lib.ptr(_lib.i32c(allp.indptr)), _lib.ptr(_lib.i32c(allp.indices)),
                                          _lib.ptr(prop), prop.size, self.device))
            self._h = h
            _lib.track(self)
        return self._h

    def negatives(self, seed=1234):
        """(evaluated users, their sampled negatives [n, num_negatives], stream draws consumed)."""
        L, h = _lib.lib(), self._handle()
        n = C.c_int32(0)
        _lib.check(L.cymf_eval_num_users(h, C.byref(n)))
        users = np.empty(n.value, dtype=np.int32)
        neg = np.empty((n.value, self.num_negatives), dtype=np.int32)
        used = C.c_int64(0)
        _lib.check(L.cymf_eval_negatives(h, int(seed), int(self.num_negatives), _lib.ptr(users), _lib.ptr(neg), C.byref(used)))
        return users, neg, used.value

    def evaluate(self, W, H, seed=1234):
        _W = _lib.f64c(np.asarray(W).astype(np.float64, copy=False))
        _H = _lib.f64c(np.asarray(H).astype(np.float64, copy=False))
        U, I = self.X.shape
        if _W.ndim != 2 or _H.ndim != 2 or _W.shape[0] != U or _H.shape[0] != I or _W.shape[1] != _H.shape[1]:
            raise ValueError(f"W {_W.shape} / H {_H.shape} do not fit the {U} x {I} evaluation matrix")
        ks = [self.k] if isinstance(self.k, int) else list(self.k)
        self.k = ks
        ks32 = np.asarray(ks, dtype=np.int32)
        kmax = max(int(ks32.max()), 1)
        disc = np.ones(kmax)
        disc[1:] = np.log2(np.arange(1, kmax) + 1.0)      # cymf/metrics.pyx:36-41
        out = np.zeros((3, len(ks), U))
        _lib.check(_lib.lib().cymf_eval_run(self._handle(), _lib.ptr(_W), _lib.ptr(_H), _W.shape[1], int(seed),
                                            int(self.num_negatives), _lib.ptr(ks32), len(ks), _lib.ptr(disc),
                                            1 if self.unbiased else 0, _lib.ptr(out)))
        return {f"{m}@{k}": out[_METRIC_ROW[m], ki].mean() for ki, k in enumerate(ks) for m in self.metrics}

    def close(self):
        if getattr(self, "_h", None):
            _lib.lib().cymf_eval_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class AverageOverAllEvaluator(Evaluator):
    def __init__(self, X, X_train=None, metrics=["DCG", "Recall", "MAP"], k=5, num_negatives=100, device=0):
        super().__init__(X, X_train, metrics, k, num_negatives, unbiased=False, device=device)


AoaEvaluator = AverageOverAllEvaluator


class UnbiasedEvaluator(Evaluator):
    def __init__(self, X, X_train=None, metrics=["DCG", "Recall", "MAP"], k=5, num_negatives=100, device=0):
        super().__init__(X, X_train, metrics, k, num_negatives, unbiased=True, device=device)
