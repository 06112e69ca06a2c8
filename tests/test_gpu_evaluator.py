"""Evaluator harness (cymf_amd/evaluator.py -> csrc/eval.hip: candidate walk on the device index
stream, fp64 scoring, top-k ranking and metrics on the GPU) against an independent restatement of
cymf/evaluator.pyx:57-139 driven by the oracle's stream and the numpy metric restatements.  The
reference evaluator itself does not compile here -> parity unpinned for the end-to-end numbers;
its metric functions are pinned (tests/golden/metrics.npz)."""
import os

import numpy as np
import pytest
from scipy import sparse

import oracle
from cymf_amd import BPR, Evaluator, UnbiasedEvaluator, metrics as M, synthetic

pytestmark = pytest.mark.gpu


def _restated(Xte, Xtr, W, H, k=5, num_negatives=100, seed=1234, unbiased=False, want_negatives=False):
    Xte = sparse.csr_matrix(Xte)
    U, I = Xte.shape
    allpos = (Xte + sparse.csr_matrix(Xtr)).tocsr() if Xtr is not None else Xte
    prop = np.maximum(np.asarray(Xte.mean(axis=0)).flatten(), 1e-4)
    stream = oracle.uniform_stream(seed, I, U * num_negatives * 8 + 100000).tolist()
    cur = 0
    ks = [k] if isinstance(k, int) else list(k)
    out = {f"{m}@{kk}": np.zeros(U) for kk in ks for m in ("DCG", "Recall", "MAP")}
    negs = []
    for u in range(U):
        te = Xte.indices[Xte.indptr[u]:Xte.indptr[u + 1]]
        if len(te) == 0:
            continue
        pos = set(allpos.indices[allpos.indptr[u]:allpos.indptr[u + 1]].tolist())
        items, fb = list(te), [1] * len(te)
        for _ in range(num_negatives):
            it = stream[cur]; cur += 1
            while it in pos:
                it = stream[cur]; cur += 1
            items.append(it)
            fb.append(0)
        negs.append(items[len(te):])
        if W is None:
            continue
        order = np.dot(H[np.array(items)], W[u]).argsort()[::-1]
        y = np.array(fb, dtype=np.int32)[order]
        for kk in ks:
            if unbiased:
                p = prop[order]
                out[f"DCG@{kk}"][u] = M.dcg_at_k_with_ips(y, p, kk)
                out[f"Recall@{kk}"][u] = M.recall_at_k_with_ips(y, p, kk)
                out[f"MAP@{kk}"][u] = M.average_precision_at_k_with_ips(y, p, kk)
            else:
                out[f"DCG@{kk}"][u] = oracle.dcg_at_k(y, kk)
                out[f"Recall@{kk}"][u] = oracle.recall_at_k(y, kk)
                out[f"MAP@{kk}"][u] = oracle.ap_at_k(y, kk)
    if want_negatives:
        return np.array(negs, dtype=np.int32).reshape(len(negs), num_negatives), cur
    return {key: v.mean() for key, v in out.items()}


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
    # IPS variants (propensities indexed by candidate position, evaluator.pyx:92), several k
    gotu = UnbiasedEvaluator(Xte, Xtr, k=[3, 10]).evaluate(W, H)
    wantu = _restated(Xte, Xtr, W, H, k=[3, 10], unbiased=True)
    for key in wantu:
        assert gotu[key] == pytest.approx(wantu[key], rel=1e-12), key
    # k > 64 (cymf/evaluator.pyx:29 takes any k): the ranked list moves from the lanes into LDS; k beyond the
    # candidate count ranks everything
    gotw = Evaluator(Xte, Xtr, k=[5, 65, 100, 5000]).evaluate(W, H)
    wantw = _restated(Xte, Xtr, W, H, k=[5, 65, 100, 5000])
    for key in wantw:
        assert gotw[key] == pytest.approx(wantw[key], rel=1e-12), key
    assert gotw["Recall@5"] == pytest.approx(want["Recall@5"], rel=1e-12) and gotw["Recall@5000"] == pytest.approx((np.diff(Xte.indptr) > 0).mean(), rel=1e-12)   # every held-out item ranked; users without one count 0 (evaluator.pyx:134-136)
    gotwu = UnbiasedEvaluator(Xte, Xtr, k=[80]).evaluate(W, H)
    wantwu = _restated(Xte, Xtr, W, H, k=[80], unbiased=True)
    for key in wantwu:
        assert gotwu[key] == pytest.approx(wantwu[key], rel=1e-12), key
    with pytest.raises(ValueError):
        Evaluator(Xte, Xtr).evaluate(W[:-1], H)


@pytest.mark.parametrize("num_negatives,block", [(100, None), (7, None), (150, "257")])
def test_candidate_walk_is_the_sequential_stream(num_negatives, block, monkeypatch):
    """The negatives of every user equal the reference's sequential redraw loop (evaluator.pyx:80-88),
    bit for bit, including users with a third of the catalogue as positives (many redraws), users
    without held-out items (skipped, consume nothing) and draw blocks that end mid-user."""
    if block:
        monkeypatch.setenv("CYMF_EVAL_BLOCK", block)
    rs = np.random.RandomState(5)
    U, I = 300, 500
    dense = (rs.rand(U, I) < 0.02).astype(np.float64)
    dense[::7] = (rs.rand(len(dense[::7]), I) < 0.35)          # heavy users: many rejected draws
    Xall = sparse.csr_matrix(dense)
    Xtr, Xte = _split(Xall, 1)
    Xte = sparse.lil_matrix(Xte)
    Xte[5:40:3] = 0                                              # users without test items
    Xte = sparse.csr_matrix(Xte)
    Xte.eliminate_zeros()
    ev = Evaluator(Xte, Xtr, num_negatives=num_negatives)
    users, neg, used = ev.negatives(seed=99)
    want, want_used = _restated(Xte, Xtr, None, None, num_negatives=num_negatives, seed=99, want_negatives=True)
    assert np.array_equal(users, np.nonzero(np.diff(Xte.indptr))[0])
    assert np.array_equal(neg, want)
    assert used == want_used
    # few candidates: k larger than the list (numpy slices y[:k] short, metrics.pyx:33)
    W, H = rs.normal(size=(U, 8)), rs.normal(size=(I, 8))
    if num_negatives == 7:
        got = Evaluator(Xte, Xtr, num_negatives=2, k=[5, 20]).evaluate(W, H, seed=3)
        ref = _restated(Xte, Xtr, W, H, k=[5, 20], num_negatives=2, seed=3)
        for key in ref:
            assert got[key] == pytest.approx(ref[key], rel=1e-12), key


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
    assert abs(hog["Recall@5"] - ref["Recall@5"]) < 0.015    # (within 0.01 at the bars of tests/test_gpu_order_fidelity.py's split)
    assert ref["Recall@5"] > 3 * ev.evaluate(*oracle.reference_init(943, 1682, K))["Recall@5"] or ref["Recall@5"] > 0.1
    # evaluator hook + early stopping
    me = BPR(K, 0.01, "adam", 0.01)
    me.fit(Xtr, num_epochs=8, num_threads=1, valid_evaluator=ev, early_stopping=True, verbose=False)
    assert np.isfinite(me.valid_dcg) and me.valid_dcg > 0
    assert ev.evaluate(me.W, me.H)["DCG@5"] == pytest.approx(me.valid_dcg, rel=1e-12)
