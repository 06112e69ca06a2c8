"""Sampled-negative ranking evaluation (cymf/evaluator.pyx:34-149): the harness that produces
DCG/Recall/MAP@k between epochs.  Host numpy, like the reference's Python-level loop; its
candidate sampling consumes the same mt19937 index stream, generated on the GPU through
cymf_rng_fill_uniform.  Quirks kept: negatives may repeat, ties are broken by
argsort()[::-1], the mean runs over ALL users including those without test items."""
import numpy as np
from scipy import sparse

from . import _lib, metrics as M


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
        self.metrics = metrics
        self.k = k
        self.num_negatives = num_negatives
        self.unbiased = unbiased
        self.device = device

    def _stream(self, seed, n_items):
        """Generator over UniformGenerator(0, I, seed) draws (cymf/evaluator.pyx:82), fetched in blocks."""
        block = 1 << 16
        pos = 0
        while True:
            chunk = _lib.rng_fill_uniform(seed, n_items, block, skip=pos, device=self.device)
            pos += block
            for v in chunk:
                yield int(v)

    def evaluate(self, W, H, seed=1234):
        _W = np.asarray(W, dtype=np.float64)
        _H = np.asarray(H, dtype=np.float64)
        U, I = self.X.shape
        ks = [self.k] if isinstance(self.k, int) else list(self.k)
        self.k = ks
        buff = {f"{m}@{k}": np.zeros(U) for k in ks for m in self.metrics}
        indptr, indices = self.X.indptr, self.X.indices
        all_indptr, all_indices = self.user_positives.indptr, self.user_positives.indices
        gen = self._stream(seed, I)
        for user in range(U):
            if indptr[user] == indptr[user + 1]:
                continue
            items = list(indices[indptr[user]:indptr[user + 1]])
            feedbacks = [1] * len(items)
            positives = set(all_indices[all_indptr[user]:all_indptr[user + 1]].tolist())
            for _ in range(self.num_negatives):
                item = next(gen)
                while item in positives:
                    item = next(gen)
                items.append(item)
                feedbacks.append(0)
            order = np.dot(_H[np.array(items)], _W[user]).argsort()[::-1]
            y = np.array(feedbacks, dtype=np.int32)[order]
            if self.unbiased:
                # the reference indexes propensities by rank position, not by item id (evaluator.pyx:116)
                p = self.propensity_scores[order]
            for k in ks:
                for m in self.metrics:
                    if self.unbiased:
                        fn = {"DCG": M.dcg_at_k_with_ips, "Recall": M.recall_at_k_with_ips,
                              "MAP": M.average_precision_at_k_with_ips}[m]
                        buff[f"{m}@{k}"][user] = fn(y, p, k)
                    else:
                        fn = {"DCG": M.dcg_at_k, "Recall": M.recall_at_k, "MAP": M.average_precision_at_k}[m]
                        buff[f"{m}@{k}"][user] = fn(y, k)
        return {key: val.mean() for key, val in buff.items()}


class AverageOverAllEvaluator(Evaluator):
    def __init__(self, X, X_train=None, metrics=["DCG", "Recall", "MAP"], k=5, num_negatives=100, device=0):
        super().__init__(X, X_train, metrics, k, num_negatives, unbiased=False, device=device)


AoaEvaluator = AverageOverAllEvaluator


class UnbiasedEvaluator(Evaluator):
    def __init__(self, X, X_train=None, metrics=["DCG", "Recall", "MAP"], k=5, num_negatives=100, device=0):
        super().__init__(X, X_train, metrics, k, num_negatives, unbiased=True, device=device)
