"""Lock-free (throughput) mode held to the REFERENCE'S OWN order.

The reference trains in the shuffled order (cymf/bpr.pyx:104,162-169); the lock-free kernels bucket the triplets by positive
item inside `steps_per_epoch` windows of that order and run them concurrently.  Here fit(num_threads != 1) with its DEFAULT
steps_per_epoch is compared with the sequential oracle in the shuffled order (oracle.bpr_fit, pinned bit for bit to the
compiled reference) on C1- (ml-100k-) and C2- (ml-1m-) shaped data: held-out Recall@5 within 0.01, final loss within 3 %,
factor norms within 10 % (5 % for SGD / AdaGrad).  The table behind the default is tools/order_fidelity.py -> DESIGN.md 4."""
import numpy as np
import pytest

import oracle
from cymf_amd import BPR, synthetic
from cymf_amd.evaluator import Evaluator

pytestmark = pytest.mark.gpu


def _split(X, seed):
    rs = np.random.RandomState(seed)
    mask = rs.rand(X.nnz) < 0.15
    Xte, Xtr = X.copy(), X.copy()
    Xte.data = Xte.data * mask
    Xtr.data = Xtr.data * (~mask)
    Xte.eliminate_zeros()
    Xtr.eliminate_zeros()
    return Xtr, Xte


@pytest.mark.parametrize("config,opt,lr", [("C1", "sgd", 0.05), ("C1", "adagrad", 0.05), ("C1", "adam", 0.01),
                                           ("C2", "sgd", 0.05), ("C2", "adagrad", 0.05), ("C2", "adam", 0.002)])
def test_default_steps_follow_the_reference_order(config, opt, lr):
    X, K = synthetic.config_matrix(config)
    Xtr, Xte = _split(X, 3)
    ev = Evaluator(Xte, Xtr)
    epochs = 30
    W, H, losses = oracle.bpr_fit(Xtr, K, opt, lr, 0.01, epochs)
    ref = ev.evaluate(W, H)
    m = BPR(K, lr, opt, 0.01)
    m.fit(Xtr, num_epochs=epochs, num_threads=8, verbose=False)
    got = ev.evaluate(m.W, m.H)
    assert m.steps_per_epoch_ >= 16
    assert abs(got["Recall@5"] - ref["Recall@5"]) < 0.01, (got["Recall@5"], ref["Recall@5"])
    assert abs(m.losses[-1] / losses[-1] - 1) < 0.03, (m.losses[-1], losses[-1])
    bar = 0.10 if opt == "adam" else 0.05
    nW, nH = np.linalg.norm(m.W) / np.linalg.norm(W) - 1, np.linalg.norm(m.H) / np.linalg.norm(H) - 1
    assert abs(nW) < bar and abs(nH) < bar, (nW, nH)
    assert ref["Recall@5"] > 0.2                      # the comparison is between trained models
