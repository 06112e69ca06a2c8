"""The oracle against the REFERENCE ITSELF, compiled in place by oracle/build_ref.py -- into a directory
OUTSIDE this repository (build_ref.ref_dir()), because everything under the repository root ships to the GPU
box and the reference must not travel in any form.  Build container only: where /root/reference is absent every
test here skips, nothing is imported at collection time, and the committed fixtures keep the pin.
Bit-for-bit (np.array_equal) on fresh seeded inputs that are not among the committed fixtures."""
import glob
import os
import types

import numpy as np
import pytest
from scipy import sparse

import oracle
from conftest import ROOT
from oracle import build_ref


@pytest.fixture(scope="module")
def ref():
    """The compiled reference's classes; imported here (never at module level) and only in the build container."""
    if not build_ref.available():
        pytest.skip("/root/reference absent (GPU box): the compiled reference never travels; fixtures keep the pin")
    try:
        build_ref.build(verbose=False)
        mods = build_ref.import_ref()
    except Exception as e:  # pragma: no cover
        pytest.skip(f"compiled reference not buildable here: {e}")
    if mods is None:  # pragma: no cover
        pytest.skip("compiled reference not importable")
    return types.SimpleNamespace(BPR=mods["bpr"].BPR, GloVe=mods["glove"].GloVe, RelMF=mods["relmf"].RelMF,
                                 metrics=mods["metrics"], read_text=mods["glove"].read_text, dir=build_ref.ref_dir())


def test_no_compiled_reference_under_the_repository_root():
    """gpurun ships every file under the root (git-ignored ones too): a compiled reference module there would
    travel to the GPU box.  oracle/_ref/ was round 1's location and must stay empty/absent."""
    hits = glob.glob(os.path.join(ROOT, "oracle", "_ref", "**", "*.so"), recursive=True)
    hits += [p for p in glob.glob(os.path.join(ROOT, "**", "cymf", "*.cpython-*.so"), recursive=True)]
    assert not hits, hits
    assert os.path.commonpath([build_ref.ref_dir(), ROOT]) != ROOT


def _X(U, I, n, seed):
    rs = np.random.RandomState(seed)
    X = sparse.csr_matrix((np.ones(n), (rs.randint(0, U, n), rs.randint(0, I, n))), shape=(U, I))
    X.data[:] = 1
    return X


@pytest.mark.parametrize("opt", ["sgd", "adagrad", "adam"])
@pytest.mark.parametrize("K", [5, 64, 100])
def test_bpr_bit_exact(ref, opt, K):
    X = _X(150, 220, 4000, K)
    m = ref.BPR(K, 0.03, opt, 0.02)
    m.fit(X, num_epochs=2, num_threads=1, verbose=False)
    W, H, _ = oracle.bpr_fit(X, K, opt, 0.03, 0.02, 2)
    assert np.array_equal(W, m.W) and np.array_equal(H, m.H)


def test_bpr_dense_input_and_unsorted_indices(ref):
    # ndarray input (cymf/bpr.pyx:83-84) and a CSR whose rows are stored unsorted: X.nonzero() order is the storage order
    rs = np.random.RandomState(0)
    D = (rs.rand(40, 50) < 0.2).astype(np.float64)
    m = ref.BPR(7, 0.05, "sgd", 0.01)
    m.fit(D, num_epochs=2, num_threads=1, verbose=False)
    W, H, _ = oracle.bpr_fit(sparse.csr_matrix(D), 7, "sgd", 0.05, 0.01, 2)
    assert np.array_equal(W, m.W) and np.array_equal(H, m.H)
    X = sparse.csr_matrix(D)
    for r in range(X.shape[0]):                       # reverse every row's storage order
        s, e = X.indptr[r], X.indptr[r + 1]
        X.indices[s:e] = X.indices[s:e][::-1].copy()
    X.has_sorted_indices = False
    m = ref.BPR(7, 0.05, "sgd", 0.01)
    m.fit(X.copy(), num_epochs=2, num_threads=1, verbose=False)
    W, H, _ = oracle.bpr_fit(X.copy(), 7, "sgd", 0.05, 0.01, 2)
    assert np.array_equal(W, m.W) and np.array_equal(H, m.H)


@pytest.mark.parametrize("opt", ["sgd", "adagrad", "adam"])
def test_relmf_bit_exact(ref, opt):
    rs = np.random.RandomState(7)
    X = (rs.rand(25, 33) < 0.15).astype(np.float64)
    m = ref.RelMF(6, 0.2, 0.04, opt, 0.02)
    m.fit(X, num_epochs=2, num_threads=1)
    W, H = oracle.reference_init(25, 33, 6)
    prop = np.maximum(X.mean(axis=0) / X.mean(axis=0).max(), 1e-5) ** 0.5
    om = oracle.RelMf(W, H, opt, 0.04, 0.02, 0.2)
    for _ in range(2):
        om.epoch(X, prop)
    assert np.array_equal(W, m.W) and np.array_equal(H, m.H)


def test_glove_bit_exact(ref):
    rs = np.random.RandomState(9)
    V, K = 70, 12
    X = sparse.csr_matrix((rs.lognormal(0, 1, 900), (rs.randint(0, V, 900), rs.randint(0, V, 900))), shape=(V, V))
    np.random.seed(123)
    g = ref.GloVe(K, 0.04, 0.6, 5.0)
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


def test_metrics_bit_exact(ref):
    rs = np.random.RandomState(2)
    for _ in range(50):
        y = (rs.rand(rs.randint(1, 120)) < 0.1).astype(np.int32)
        for k in (1, 5, 20):
            assert oracle.dcg_at_k(y, k) == ref.metrics.dcg_at_k(y, k)
            assert oracle.recall_at_k(y, k) == ref.metrics.recall_at_k(y, k)
            assert oracle.ap_at_k(y, k) == ref.metrics.average_precision_at_k(y, k)


@pytest.mark.parametrize("min_count,window", [(1, 3), (2, 10), (3, 5)])
def test_read_text_matches_the_compiled_reference(ref, tmp_path, min_count, window):
    """cymf_amd.glove.read_text (vectorised host code) against cymf.glove.read_text (cymf/glove.pyx:183-241) on a multi-line
    file: same vocabulary in the same order, same co-occurrence matrix.  The reference sums 1/distance in hash-map order,
    ours in sorted COO order, so the values agree to rounding (measured 1.4e-14), not bit for bit.  Every word stands at
    least once inside a line (a word seen only at line boundaries raises KeyError in both, tested in test_host_logic)."""
    from cymf_amd.glove import read_text
    rs = np.random.RandomState(11)
    vocab = [f"w{i}" for i in range(40)]
    p = (1.0 / np.arange(1, 41)); p /= p.sum()
    lines = []
    for n in (60, 1, 45, 80, 30):
        lines.append(" ".join(["w0"] + list(rs.choice(vocab, size=n, p=p)) + ["w0"]))
    f = tmp_path / "corpus.txt"
    f.write_text("\n".join(lines))
    Mr, i2w_r = ref.read_text(str(f), min_count, window)
    M, i2w = read_text(str(f), min_count, window)
    assert dict(i2w_r) == dict(i2w)
    assert M.shape == Mr.shape and M.nnz == Mr.nnz and M.nnz > 300
    d = abs(M - Mr)
    assert d.nnz == 0 or d.max() <= 1e-12 * max(1.0, abs(Mr).max())
