"""Edge cases of the class surface on the GPU: the reference's DEFAULT constructors (K=20 / K=50, Adam; cymf/bpr.pyx:44-52,
cymf/wmf.pyx:44-50, cymf/relmf.pyx:46-54, cymf/glove.pyx:57-65, cymf/expomf.pyx:45-51), component counts that are no
multiple of the wave or MFMA tile, one-row / one-column problems, and empty inputs.  Exact-mode results are compared with
the oracle; HOGWILD-mode ones must be finite and at the same loss level."""
import numpy as np
import pytest
from scipy import sparse

import oracle
from cymf_amd import BPR, WMF, ExpoMF, GloVe, RelMF, synthetic

pytestmark = pytest.mark.gpu


def rel_fro(a, b):
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


@pytest.mark.parametrize("K", [1, 3, 20, 33, 130])
def test_bpr_component_counts_off_the_tile_sizes(K):
    X = synthetic.implicit_matrix(90, 70, 900, 5)
    W, H, _ = oracle.bpr_fit(X, K, "adam", 0.001, 0.01, 2)
    m = BPR(num_components=K)                                   # reference defaults: adam, lr 0.001, wd 0.01
    m.fit(X, num_epochs=2, num_threads=1, verbose=False, dtype="float64")
    assert rel_fro(m.W, W) <= 1e-10 and rel_fro(m.H, H) <= 1e-10
    t = BPR(num_components=K)
    t.fit(X, num_epochs=2, num_threads=8, verbose=False)        # HOGWILD mode, f32
    assert np.isfinite(t.W).all() and np.isfinite(t.H).all()
    # 900 triplets on 70 items, all in flight at once: with Adam's normalised steps the lock-free order ends with ~40 %
    # (H) / ~25 % (W) larger norms than the sequential one here, for every K alike (cf. the RelMF step-path test)
    assert abs(np.linalg.norm(t.H) / np.linalg.norm(H) - 1) < 0.6 and abs(np.linalg.norm(t.W) / np.linalg.norm(W) - 1) < 0.4
    # SGD has no such effect: the lock-free result stays within a few per cent of the sequential one for every K
    Ws, Hs, _ = oracle.bpr_fit(X, K, "sgd", 0.05, 0.01, 2)
    s = BPR(K, 0.05, "sgd", 0.01)
    s.fit(X, num_epochs=2, num_threads=8, verbose=False)
    assert abs(np.linalg.norm(s.H) / np.linalg.norm(Hs) - 1) < 0.05 and abs(np.linalg.norm(s.W) / np.linalg.norm(Ws) - 1) < 0.05
    assert rel_fro(s.W, Ws) < 0.1 and rel_fro(s.H, Hs) < 0.1


def test_default_constructors_fit():
    X = synthetic.implicit_matrix(120, 90, 1500, 6)
    for cls in (BPR, WMF, RelMF, ExpoMF):
        m = cls()
        assert m.num_components == 20
        kw = {} if cls is RelMF else {"verbose": False}
        m.fit(X if cls is not RelMF else X.toarray(), num_epochs=2, **kw)
        assert m.W.shape == (120, 20) and m.H.shape == (90, 20) and m.W.dtype == np.float64
        assert np.isfinite(m.W).all() and np.isfinite(m.H).all()
    C = synthetic.cooccurrence_matrix(150, 4000, 7)
    g = GloVe()
    g.fit(C, 2, 1)
    assert g.W.shape == (150, 50) and np.isfinite(g.W).all()


def test_wmf_default_k20_vs_oracle():
    X = synthetic.implicit_matrix(120, 90, 1500, 6)
    W, H = oracle.reference_init(120, 90, 20)
    oracle.wmf_fit(X, W, H, 2)
    m = WMF()
    m.fit(X, num_epochs=2, verbose=False, dtype="float64")
    assert rel_fro(m.W, W) <= 1e-9 and rel_fro(m.H, H) <= 1e-9
    m32 = WMF()
    m32.fit(X, num_epochs=2, verbose=False)
    assert rel_fro(m32.W, W) <= 1e-4 and rel_fro(m32.H, H) <= 1e-4


@pytest.mark.parametrize("shape,cells", [((1, 1), [(0, 0)]), ((1, 9), [(0, 2), (0, 7)]), ((9, 1), [(3, 0), (8, 0)]), ((4, 5), [])])
def test_degenerate_shapes(shape, cells):
    """One user, one item, or no interaction at all.  With one item every draw hits the positive (all skipped,
    cymf/bpr.pyx:165-167); with no interaction the loops have nothing to do and the factors stay at their initial values."""
    r = np.array([c[0] for c in cells], dtype=np.int64)
    c = np.array([c[1] for c in cells], dtype=np.int64)
    X = sparse.csr_matrix((np.ones(len(cells)), (r, c)), shape=shape)
    W0, H0 = oracle.reference_init(shape[0], shape[1], 8)
    for threads in (1, 4):
        m = BPR(8, 0.05, "sgd", 0.01)
        m.fit(X, num_epochs=2, num_threads=threads, verbose=False, dtype="float64" if threads == 1 else None)
        assert np.isfinite(m.W).all() and np.isfinite(m.H).all()
        if threads == 1:
            W, H, _ = oracle.bpr_fit(X, 8, "sgd", 0.05, 0.01, 2)
            assert rel_fro(m.W, W) <= 1e-10 and rel_fro(m.H, H) <= 1e-10
        if not cells or shape[1] == 1:
            assert rel_fro(m.W, W0) <= 1e-7 and rel_fro(m.H, H0) <= 1e-7
    w = WMF(8, 0.01, 10.0)
    w.fit(X, num_epochs=1, verbose=False, dtype="float64")
    Wr, Hr = oracle.reference_init(shape[0], shape[1], 8)
    oracle.wmf_fit(X, Wr, Hr, 1)
    assert np.allclose(w.W, Wr, rtol=1e-9, atol=1e-12) and np.allclose(w.H, Hr, rtol=1e-9, atol=1e-12)
    if not cells:
        assert (w.W == 0).all() and (w.H == 0).all()            # empty rows are zeroed (cymf/wmf.pyx:154-156)


def test_relmf_all_zero_and_single_cell():
    """No click at all: the item means are 0/0 in the reference (relmf.pyx:88, NaN propensities) -- this build refuses
    instead of training on NaN; a single clicked cell trains and matches the oracle."""
    Xd = np.zeros((6, 7))
    Xd[2, 3] = 1.0
    prop = np.maximum(Xd.mean(axis=0) / Xd.mean(axis=0).max(), 1e-5) ** 0.5
    W, H = oracle.reference_init(6, 7, 8)
    om = oracle.RelMf(W, H, "sgd", 0.05, 0.01, 0.1)
    om.epoch(Xd, prop)
    m = RelMF(8, 0.1, 0.05, "sgd", 0.01)
    m.fit(Xd, num_epochs=1, num_threads=1, dtype="float64")
    assert rel_fro(m.W, W) <= 1e-10 and rel_fro(m.H, H) <= 1e-10
    z = RelMF(8, 0.1, 0.05, "sgd", 0.01)
    with np.errstate(all="ignore"):
        try:
            z.fit(np.zeros((6, 7)), num_epochs=1, num_threads=1)
            assert z.W.shape == (6, 8)                          # if it trains, shapes hold; NaNs are the reference's own outcome
        except (ValueError, RuntimeError):
            pass


def test_glove_without_pairs_and_single_pair():
    empty = sparse.csr_matrix((5, 5))
    np.random.seed(3)
    g = GloVe(8, 0.05)
    g.fit(empty, 2, 1)
    assert g.W.shape == (5, 8) and np.isfinite(g.W).all()
    one = sparse.csr_matrix((np.array([3.0]), ([1], [4])), shape=(5, 5))
    np.random.seed(3)
    g1 = GloVe(8, 0.05)
    g1.fit(one, 2, 1, dtype="float64")
    np.random.seed(3)                                            # glove.pyx:91-94: no seeding in fit, the caller's state
    V, K = 5, 8
    W = np.random.uniform(-0.5, 0.5, (V, K)) / K
    np.random.uniform(-0.5, 0.5, (V,))
    Wc = np.random.uniform(-0.5, 0.5, (V, K)) / K
    assert g1.W.shape == (5, 8) and np.isfinite(g1.W).all()
    touched = np.abs(g1.W - (W + Wc) / 2).sum(axis=1) > 0
    assert set(np.nonzero(touched)[0]) <= {1, 4}                 # only the pair's two words moved
