"""ExpoMF on the GPU (csrc/expomf.hip behind cymf_amd.ExpoMF) against the numpy restatement of
cymf/expomf.pyx:105-207 (oracle.expomf_fit).  PARITY UNPINNED by the reference: expomf.pyx needs cblas.h to
build, which the image lacks (tests/golden/make_golden.py), so the restatement is the only check."""
import numpy as np
import pytest
from scipy import sparse

import oracle
from cymf_amd import ExpoMF, Evaluator, synthetic

pytestmark = pytest.mark.gpu


def _init(U, I, K):
    np.random.seed(4321)                                             # expomf.pyx:99-102
    return np.random.randn(U, K) * 0.01, np.random.randn(I, K) * 0.01


@pytest.mark.parametrize("K,lam_y,wd", [(20, 1.0, 0.01), (8, 0.5, 0.1), (64, 1.0, 0.01), (140, 1.0, 0.01)])   # K=140: system in a global slice
def test_expomf_vs_restatement_unpinned(K, lam_y, wd):
    X = synthetic.implicit_matrix(150, 220, 3000, 71).tolil()
    X[4] = 0                                                         # a user without positives: row of zeros (:178-182)
    X[:, 9] = 0                                                      # an item nobody touched
    X = X.tocsr()
    X.eliminate_zeros()
    W, H = _init(150, 220, K)
    oracle.expomf_fit(X, W, H, 3, lam_y, wd)
    m = ExpoMF(K, lam_y, wd)
    m.fit(X, num_epochs=3, verbose=False)
    assert np.linalg.norm(m.W - W) <= 1e-9 * np.linalg.norm(W) and np.linalg.norm(m.H - H) <= 1e-9 * np.linalg.norm(H)
    assert (m.W[4] == 0).all() and (m.H[9] == 0).all()


def test_expomf_class_surface_and_learning():
    with pytest.raises(ValueError):
        ExpoMF().fit(None)
    with pytest.raises(ValueError):
        ExpoMF().fit([[1, 0], [0, 1]])
    with pytest.raises(ValueError):
        ExpoMF().fit(sparse.eye(4).tocsr(), early_stopping=True)
    X, K = synthetic.config_matrix("C1")                            # ml-100k-shaped, the size the reference runs it on
    rs = np.random.RandomState(0)
    mask = rs.rand(X.nnz) < 0.15
    Xte, Xtr = X.copy(), X.copy()
    Xte.data = Xte.data * mask
    Xtr.data = Xtr.data * (~mask)
    Xte.eliminate_zeros()
    Xtr.eliminate_zeros()
    ev = Evaluator(Xte, Xtr)
    m = ExpoMF(K, 1.0, 0.01)
    W0, H0 = _init(*X.shape, K)
    base = ev.evaluate(W0, H0)["Recall@5"]
    m.fit(Xtr.toarray(), num_epochs=4, valid_evaluator=ev, verbose=False)    # dense ndarray input is accepted (:85-86)
    got = ev.evaluate(m.W, m.H)["Recall@5"]
    assert got > 3 * base and got > 0.1 and np.isfinite(m.valid_dcg)
    # warm start honours preset factors (:99-102)
    m2 = ExpoMF(K, 1.0, 0.01)
    m2.W, m2.H = m.W.copy(), m.H.copy()
    m2.fit(Xtr, num_epochs=1, verbose=False)
    assert ev.evaluate(m2.W, m2.H)["Recall@5"] > got - 0.05          # continues from the given factors (mu restarts at 0.01, :120)
