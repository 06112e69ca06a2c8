"""BASELINE.json's full sizes, through size-independent properties (the oracle cannot train 100M
triplets in seconds, but it can produce the index stream and check invariants):
  C3  BPR K=128, 1M users x 100k items, 100M interactions (headline config)
  C4  WMF K=64, ml-20m-shaped
  C5  GloVe K=100, text8-shaped co-occurrence
"""
import numpy as np
import pytest

import oracle
from cymf_amd import synthetic
from cymf_amd.bpr import BprTrainer
from cymf_amd.glove import GloveTrainer
from cymf_amd.wmf import WmfTrainer

pytestmark = pytest.mark.gpu


def test_c3_bpr_full_size_properties():
    U, I, nnz, K, seed = synthetic.CONFIGS["C3"]
    rows, cols, indptr = synthetic.implicit_matrix_large(U, I, nnz, seed)
    N = len(rows)
    assert N == nnz
    perm = np.random.default_rng(4321).permutation(N)
    users, pos = rows[perm], cols[perm]
    rs = np.random.RandomState(4321)
    W0 = rs.uniform(-0.1, 0.1, size=(U, K)) / K
    H0 = rs.uniform(-0.1, 0.1, size=(I, K)) / K
    W32, H32 = W0.astype(np.float32).astype(np.float64), H0.astype(np.float32).astype(np.float64)
    pair_keys = rows.astype(np.int64) * I + cols          # sorted: rows ascending, cols ascending inside a row

    # (1) zero learning rate: one epoch is the identity on W and H (float32 round trip only)
    t = BprTrainer(U, I, K, "sgd", 0.0, 0.01, mode="throughput", steps_per_epoch=25)
    t.set_data(users, pos, indptr.astype(np.int32), cols)
    t.upload(W0, H0)
    t.epochs(1)
    W, H = np.empty_like(W0), np.empty_like(H0)
    t.download(W, H)
    assert np.array_equal(W, W32) and np.array_equal(H, H32)
    # (2) every one of the 100M negatives is the draw of the global stream at its position (bit-exact,
    #     second epoch included); every skipped draw is a (user, draw) pair present in X, and a 4M-position
    #     sample of the kept ones is absent from X
    n_skipped = 0
    rs2 = np.random.RandomState(1)
    for ep in range(2):
        if ep:
            t.epochs(1)
        got = t.last_negatives()
        draws = oracle.uniform_stream(1234, I, N, skip=ep * N).astype(np.int32)
        skipped_at = np.nonzero(got < 0)[0]
        kept = got >= 0
        assert np.array_equal(got[kept], draws[kept]), ep
        q = users[skipped_at].astype(np.int64) * I + draws[skipped_at]
        assert np.array_equal(pair_keys[np.minimum(np.searchsorted(pair_keys, q), N - 1)], q), ep
        sample = rs2.randint(0, N, 4_000_000)
        sample = sample[kept[sample]]
        q = users[sample].astype(np.int64) * I + draws[sample]
        assert not (pair_keys[np.minimum(np.searchsorted(pair_keys, q), N - 1)] == q).any(), ep
        n_skipped += len(skipped_at)
        del got, draws, kept
    performed, skipped = t.stats()
    assert skipped == n_skipped and performed + skipped == 2 * N
    t.close()

    # (3) training: two epochs, finite factors, loss falls from ~log 2, hot items stay bounded
    t = BprTrainer(U, I, K, "sgd", 0.05, 0.01, mode="throughput", steps_per_epoch=25)
    t.set_data(users, pos, indptr.astype(np.int32), cols)
    t.upload(W0, H0)
    losses = t.epochs(2)
    t.download(W, H)
    t.close()
    assert np.isfinite(W).all() and np.isfinite(H).all()
    assert losses[1] < 0.8 * losses[0] < 0.8 * 0.7
    assert np.abs(H).max() < 10.0


def test_c4_wmf_full_size_normal_equations():
    U, I, nnz, K, seed = synthetic.CONFIGS["C4"]
    rows, cols, indptr = synthetic.implicit_matrix_large(U, I, nnz, seed)
    from scipy import sparse
    X = sparse.csr_matrix((np.ones(len(rows), dtype=np.float32), cols, indptr), shape=(U, I))
    Xt = X.T.tocsr()
    w, lam = 10.0, 0.01
    W0, H0 = oracle.reference_init(U, I, K)
    t = WmfTrainer(U, I, K, w, lam, dtype="float32")
    t.set_data(X.indptr, X.indices, Xt.indptr, Xt.indices)
    t.upload(W0, H0)
    t.half_sweep(0)
    W1, H1 = np.empty_like(W0), np.empty_like(H0)
    t.download(W1, H1)
    t.half_sweep(1)
    W2, H2 = np.empty_like(W0), np.empty_like(H0)
    t.download(W2, H2)
    t.close()
    assert np.array_equal(W1, W2) and np.isfinite(H2).all()
    # user rows solve their normal equations against H0; item rows against W1 (sampled, float64 on the host)
    def residual(Xc, Y, Xsolved, idx):
        Y = Y.astype(np.float32).astype(np.float64)
        G = Y.T @ Y + lam * np.eye(K)
        worst = 0.0
        for r in idx:
            s = Xc.indices[Xc.indptr[r]:Xc.indptr[r + 1]]
            if len(s) == 0:
                assert (Xsolved[r] == 0).all()
                continue
            A = G + (w - 1) * Y[s].T @ Y[s]
            b = w * Y[s].sum(axis=0)
            worst = max(worst, np.linalg.norm(A @ Xsolved[r] - b) / np.linalg.norm(b))
        return worst
    rs = np.random.RandomState(0)
    heavy_items = np.argsort(-np.diff(Xt.indptr))[:3]                  # the longest rows (10^4..10^5 entries)
    assert residual(X, H0, W1, rs.randint(0, U, 200)) < 1e-4
    assert residual(Xt, W1, H2, np.concatenate([rs.randint(0, I, 200), heavy_items])) < 1e-4


def test_c5_glove_full_size_trains():
    V, _, nnz, K, seed = synthetic.CONFIGS["C5"]
    X = synthetic.cooccurrence_matrix(V, nnz, seed)
    ce, cx = X.nonzero()
    rs = np.random.RandomState(3)
    p = rs.permutation(len(ce))
    ce, cx, cnt = ce[p], cx[p], X.data[p]
    W = rs.uniform(-0.5, 0.5, (V, K)) / K
    b = rs.uniform(-0.5, 0.5, (V,)) / K
    Wc = rs.uniform(-0.5, 0.5, (V, K)) / K
    bc = rs.uniform(-0.5, 0.5, (V,)) / K
    t = GloveTrainer(V, V, K, 0.05, 10.0, 0.75, dtype="float32", mode="throughput")
    t.set_data(ce, cx, cnt)
    t.upload(W, b, Wc, bc)
    losses = t.epochs(3) / len(ce)
    t.download(W, b, Wc, bc)
    t.close()
    assert np.isfinite(W).all() and np.isfinite(Wc).all() and np.isfinite(b).all()
    assert losses[2] < losses[1] < losses[0]
    # with (near) zero parameters the loss is 0.5 f(c) log(c)^2 per pair: already the first online
    # epoch must beat that level, the third must be far below it
    f = np.minimum((cnt / 10.0) ** 0.75, 1.0)
    L0 = float(np.mean(0.5 * f * np.log(cnt) ** 2))
    assert losses[0] < L0 and losses[2] < 0.5 * L0
