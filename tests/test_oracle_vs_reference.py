"""The oracle against the REFERENCE ITSELF, compiled in place by oracle/build_ref.py into oracle/_ref/
(build container only: /root/reference does not travel, so this module skips where oracle/_ref is absent).
Bit-for-bit (np.array_equal) on fresh seeded inputs that are not among the committed fixtures."""
import os
import sys

import numpy as np
import pytest
from scipy import sparse

import oracle
from conftest import ROOT

_REF = os.path.join(ROOT, "oracle", "_ref")
if not os.path.isdir(os.path.join(_REF, "cymf")) or not any(f.startswith("bpr.") for f in os.listdir(os.path.join(_REF, "cymf"))):
    pytest.skip("oracle/_ref not built (python oracle/build_ref.py needs /root/reference)", allow_module_level=True)
sys.path.insert(0, _REF)
try:
    from cymf.bpr import BPR as RefBPR
    from cymf.glove import GloVe as RefGloVe
    from cymf.relmf import RelMF as RefRelMF
    from cymf import metrics as ref_metrics
except Exception as e:  # pragma: no cover
    pytest.skip(f"compiled reference not importable: {e}", allow_module_level=True)


def _X(U, I, n, seed):
    rs = np.random.RandomState(seed)
    X = sparse.csr_matrix((np.ones(n), (rs.randint(0, U, n), rs.randint(0, I, n))), shape=(U, I))
    X.data[:] = 1
    return X


@pytest.mark.parametrize("opt", ["sgd", "adagrad", "adam"])
@pytest.mark.parametrize("K", [5, 64, 100])
def test_bpr_bit_exact(opt, K):
    X = _X(150, 220, 4000, K)
    m = RefBPR(K, 0.03, opt, 0.02)
    m.fit(X, num_epochs=2, num_threads=1, verbose=False)
    W, H, _ = oracle.bpr_fit(X, K, opt, 0.03, 0.02, 2)
    assert np.array_equal(W, m.W) and np.array_equal(H, m.H)


def test_bpr_dense_input_and_unsorted_indices():
    # ndarray input (cymf/bpr.pyx:83-84) and a CSR whose rows are stored unsorted: X.nonzero() order is the storage order
    rs = np.random.RandomState(0)
    D = (rs.rand(40, 50) < 0.2).astype(np.float64)
    m = RefBPR(7, 0.05, "sgd", 0.01)
    m.fit(D, num_epochs=2, num_threads=1, verbose=False)
    W, H, _ = oracle.bpr_fit(sparse.csr_matrix(D), 7, "sgd", 0.05, 0.01, 2)
    assert np.array_equal(W, m.W) and np.array_equal(H, m.H)
    X = sparse.csr_matrix(D)
    for r in range(X.shape[0]):                       # reverse every row's storage order
        s, e = X.indptr[r], X.indptr[r + 1]
        X.indices[s:e] = X.indices[s:e][::-1].copy()
    X.has_sorted_indices = False
    m = RefBPR(7, 0.05, "sgd", 0.01)
    m.fit(X.copy(), num_epochs=2, num_threads=1, verbose=False)
    W, H, _ = oracle.bpr_fit(X.copy(), 7, "sgd", 0.05, 0.01, 2)
    assert np.array_equal(W, m.W) and np.array_equal(H, m.H)


@pytest.mark.parametrize("opt", ["sgd", "adagrad", "adam"])
def test_relmf_bit_exact(opt):
    rs = np.random.RandomState(7)
    X = (rs.rand(25, 33) < 0.15).astype(np.float64)
    m = RefRelMF(6, 0.2, 0.04, opt, 0.02)
    m.fit(X, num_epochs=2, num_threads=1)
    W, H = oracle.reference_init(25, 33, 6)
    prop = np.maximum(X.mean(axis=0) / X.mean(axis=0).max(), 1e-5) ** 0.5
    om = oracle.RelMf(W, H, opt, 0.04, 0.02, 0.2)
    for _ in range(2):
        om.epoch(X, prop)
    assert np.array_equal(W, m.W) and np.array_equal(H, m.H)


def test_glove_bit_exact():
    rs = np.random.RandomState(9)
    V, K = 70, 12
    X = sparse.csr_matrix((rs.lognormal(0, 1, 900), (rs.randint(0, V, 900), rs.randint(0, V, 900))), shape=(V, V))
    np.random.seed(123)
    g = RefGloVe(K, 0.04, 0.6, 5.0)
    g.fit(X, 3, 1)
    np.random.seed(123)
    W = np.random.uniform(-0.5, 0.5, (V, K)) / K
    b = np.random.uniform(-0.5, 0.5, (V,)) / K
    _W = np.random.uniform(-0.5, 0.5, (V, K)) / K
    _b = np.random.uniform(-0.5, 0.5, (V,)) / K
    ce, cx = X.nonzero()
    ce, cx, cnt = oracle.reference_shuffle(ce, cx, X.data)
    om = oracle.Glove(W, b, _W, _b, 0.04, 5.0, 0.6)
    for _ in range(3):
        om.epoch(ce, cx, cnt)
    assert np.array_equal((W + _W) / 2.0, g.W) and np.array_equal(b, g.bias)


def test_metrics_bit_exact():
    rs = np.random.RandomState(2)
    for _ in range(50):
        y = (rs.rand(rs.randint(1, 120)) < 0.1).astype(np.int32)
        for k in (1, 5, 20):
            assert oracle.dcg_at_k(y, k) == ref_metrics.dcg_at_k(y, k)
            assert oracle.recall_at_k(y, k) == ref_metrics.recall_at_k(y, k)
            assert oracle.ap_at_k(y, k) == ref_metrics.average_precision_at_k(y, k)
