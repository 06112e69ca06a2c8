"""Evaluator harness (cymf_amd/evaluator.py, candidate sampling on the device index stream)
against an independent restatement of cymf/evaluator.pyx:57-139 driven by the oracle's stream and
metrics.  The reference evaluator itself does not compile here -> parity unpinned for the
end-to-end numbers; its metric functions are pinned (tests/golden/metrics.npz)."""
import numpy as np
import pytest

import oracle
from cymf_amd import BPR, Evaluator, UnbiasedEvaluator, synthetic

pytestmark = pytest.mark.gpu


def _restated(Xte, Xtr, W, H, k=5, num_negatives=100, seed=1234):
    U, I = Xte.shape
    allpos = (Xte + Xtr).tocsr()
    stream = iter(oracle.uniform_stream(seed, I, U * num_negatives * 2 + 1000).tolist())
    out = {m: np.zeros(U) for m in ("DCG", "Recall", "MAP")}
    for u in range(U):
        te = Xte.indices[Xte.indptr[u]:Xte.indptr[u + 1]]
        if len(te) == 0:
            continue
        pos = set(allpos.indices[allpos.indptr[u]:allpos.indptr[u + 1]].tolist())
        items, fb = list(te), [1] * len(te)
        for _ in range(num_negatives):
            it = next(stream)
            while it in pos:
                it = next(stream)
            items.append(it)
            fb.append(0)
        order = np.dot(H[np.array(items)], W[u]).argsort()[::-1]
        y = np.array(fb, dtype=np.int32)[order]
        out["DCG"][u] = oracle.dcg_at_k(y, k)
        out["Recall"][u] = oracle.recall_at_k(y, k)
        out["MAP"][u] = oracle.ap_at_k(y, k)
    return {f"{m}@{k}": v.mean() for m, v in out.items()}


def _split(X, seed):
    rs = np.random.RandomState(seed)
    mask = rs.rand(X.nnz) < 0.15
    Xte, Xtr = X.copy(), X.copy()
    Xte.data = Xte.data * mask
    Xtr.data = Xtr.data * (~mask)
    Xte.eliminate_zeros()
    Xtr.eliminate_zeros()
    return Xtr, Xte


def test_evaluator_matches_restatement():
    X = synthetic.implicit_matrix(400, 600, 12000, 41)
    Xtr, Xte = _split(X, 0)
    rs = np.random.RandomState(1)
    W, H = rs.normal(size=(400, 16)), rs.normal(size=(600, 16))
    got = Evaluator(Xte, Xtr).evaluate(W, H)
    want = _restated(Xte, Xtr, W, H)
    for key in want:
        assert got[key] == pytest.approx(want[key], rel=1e-12), key
    # another seed, several k
    got2 = Evaluator(Xte, Xtr, k=[1, 5]).evaluate(W, H, seed=7)
    assert got2["Recall@5"] == pytest.approx(_restated(Xte, Xtr, W, H, seed=7)["Recall@5"], rel=1e-12)
    assert set(got2) == {f"{m}@{k}" for m in ("DCG", "Recall", "MAP") for k in (1, 5)}
    assert np.isfinite(list(UnbiasedEvaluator(Xte, Xtr).evaluate(W, H).values())).all()


def test_bpr_recall_matches_sequential_reference_and_early_stopping():
    """ml-100k-shaped data (the real file cannot be downloaded): Recall@5 of the GPU fit in exact
    mode equals the oracle's, HOGWILD mode lands within run-to-run noise, and the
    valid_evaluator / early_stopping hook works as cymf/bpr.pyx:173-190."""
    X, K = synthetic.config_matrix("C1")
    Xtr, Xte = _split(X, 3)
    ev = Evaluator(Xte, Xtr)
    W, H, _ = oracle.bpr_fit(Xtr, K, "adam", 0.01, 0.01, 30)        # README.md:62-63 settings
    ref = ev.evaluate(W, H)
    m = BPR(K, 0.01, "adam", 0.01)
    m.fit(Xtr, num_epochs=30, num_threads=1, verbose=False)
    got = ev.evaluate(m.W, m.H)
    assert got["Recall@5"] == pytest.approx(ref["Recall@5"], abs=1e-9)
    mt = BPR(K, 0.01, "adam", 0.01)
    mt.fit(Xtr, num_epochs=30, num_threads=8, verbose=False)
    hog = ev.evaluate(mt.W, mt.H)
    assert abs(hog["Recall@5"] - ref["Recall@5"]) < 0.02
    assert ref["Recall@5"] > 3 * ev.evaluate(*oracle.reference_init(943, 1682, K))["Recall@5"] or ref["Recall@5"] > 0.1
    # evaluator hook + early stopping
    me = BPR(K, 0.01, "adam", 0.01)
    me.fit(Xtr, num_epochs=8, num_threads=1, valid_evaluator=ev, early_stopping=True, verbose=False)
    assert np.isfinite(me.valid_dcg) and me.valid_dcg > 0
    assert ev.evaluate(me.W, me.H)["DCG@5"] == pytest.approx(me.valid_dcg, rel=1e-12)
