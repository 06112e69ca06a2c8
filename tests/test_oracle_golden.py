"""The oracle (oracle/cymf_oracle.c) against the golden vectors the REFERENCE produced
(tests/golden/make_golden.py).  CPU only.  Bit-exact for the index stream and -- because the
oracle keeps the reference's operation order in fp64 -- for BPR / RelMF / GloVe factors too."""
import numpy as np
import pytest

import oracle
from conftest import csr_from_golden, golden, rel_fro


def test_raw_mt19937_known_answers():
    # std::mt19937(1234) raw words
    raw = oracle.raw_stream(1234, 6)
    assert raw.tolist() == [822569775, 2137449171, 2671936806, 3512589365, 1880026316, 2629000564]
    # equals numpy's legacy MT19937 stream (same generator, same init_genrand seeding)
    n = 5000
    ref = np.random.RandomState(1234).randint(0, 2**32, size=n, dtype=np.uint64).astype(np.uint32)
    assert np.array_equal(oracle.raw_stream(1234, n), ref)


def test_index_stream_known_answers_survey():
    assert oracle.uniform_stream(1234, 1682, 12).tolist() == [322, 837, 1046, 1375, 736, 1029, 1320, 1297, 1311, 1447, 458, 253]
    assert oracle.uniform_stream(1234, 100000, 12).tolist() == [19151, 49766, 62210, 81783, 43772, 61211, 78535, 77135, 77997, 86066, 27259, 15063]


def test_index_stream_vs_libstdcxx_fixture():
    g = golden("index_stream")
    for key in g.files:
        if key.startswith("seed99"):
            seed, rng_range = 99, 100000
        else:
            seed, rng_range = 1234, int(key[1:])
        want = g[key]
        got = oracle.uniform_stream(seed, rng_range, len(want))
        assert np.array_equal(got, want), key
    # skip= continues the same stream
    want = g["r100000"]
    assert np.array_equal(oracle.uniform_stream(1234, 100000, 96, skip=4000), want[4000:])


@pytest.mark.parametrize("name", ["bpr_60x80", "bpr_c1", "bpr_300x500_k128", "bpr_300x500_k64"])
def test_bpr_vs_reference_fixture(name):
    g = golden(name)
    X = csr_from_golden(g)
    K, lr, wd = int(g["K"]), float(g["lr"]), float(g["wd"])
    keys = [k for k in g.files if k.startswith("W_")]
    assert keys
    for key in keys:
        _, opt, ep = key.split("_")
        W, H, losses = oracle.bpr_fit(X, K, opt, lr, wd, int(ep))
        gW, gH = g[key], g["H_" + key[2:]]
        assert np.array_equal(W[:gW.shape[0]], gW), key
        assert np.array_equal(H[:gH.shape[0]], gH), key
        if "Wn_" + key[2:] in g.files:
            assert abs(np.linalg.norm(W) - float(g["Wn_" + key[2:]])) <= 1e-13 * np.linalg.norm(W)
            np.testing.assert_allclose(H.sum(axis=0), g["Hcs_" + key[2:]], rtol=0, atol=1e-13)
        assert all(np.isfinite(losses))


def test_bpr_survey_check_value():
    # SURVEY.md 8c re-creation check value
    from scipy import sparse
    rng = np.random.default_rng(0)
    U, I = 943, 1682
    r, c = rng.integers(0, U, 55000), rng.integers(0, I, 55000)
    X = sparse.csr_matrix((np.ones(55000), (r, c)), shape=(U, I))
    X.data[:] = 1
    assert X.nnz == 54052
    W, H, _ = oracle.bpr_fit(X, 20, "sgd", 0.01, 0.01, 6)
    assert f"{W.sum():.12e}" == "-3.341947290633e-01"
    assert f"{H.sum():.12e}" == "-2.542518775539e-01"
    g = golden("bpr_survey_check")
    assert W.sum() == float(g["Wsum"]) and H.sum() == float(g["Hsum"])


def test_bpr_skip_rule_and_stream_continuation():
    g = golden("bpr_60x80")
    X = csr_from_golden(g)
    W, H = oracle.reference_init(60, 80, 8)
    users, positives = oracle.reference_shuffle(*X.nonzero())
    m = oracle.Bpr(W, H, "sgd", 0.05, 0.01)
    _, n1 = m.epoch(users, positives, X.indptr, X.indices, want_negatives=True)
    _, n2 = m.epoch(users, positives, X.indptr, X.indices, want_negatives=True)
    N = len(users)
    # one draw per triplet incl. skipped ones; the stream is never reseeded (bpr.pyx:141,165-167)
    assert np.array_equal(np.concatenate([n1, n2]), oracle.uniform_stream(1234, 80, 2 * N))
    dense = X.toarray() != 0
    assert m.skipped == int(dense[users, n1].sum() + dense[users, n2].sum()) > 0


@pytest.mark.parametrize("K", [16, 100])
def test_glove_vs_reference_fixture(K):
    g = golden("glove_120")
    X = csr_from_golden(g, data=g["data"])
    V = int(g["V"])
    np.random.seed(int(g["np_seed"]))
    W = np.random.uniform(-0.5, 0.5, (V, K)) / K
    b = np.random.uniform(-0.5, 0.5, (V,)) / K
    _W = np.random.uniform(-0.5, 0.5, (V, K)) / K
    _b = np.random.uniform(-0.5, 0.5, (V,)) / K
    ce, cx = X.nonzero()
    ce, cx, cnt = oracle.reference_shuffle(ce, cx, X.data)
    m = oracle.Glove(W, b, _W, _b, float(g["lr"]), float(g["x_max"]), float(g["alpha"]))
    for _ in range(2):
        m.epoch(ce, cx, cnt)
    assert np.array_equal((W + _W) / 2.0, g[f"W_k{K}"])
    assert np.array_equal(b, g[f"bias_k{K}"])


def test_relmf_vs_reference_fixture():
    g = golden("relmf_30x40")
    X = g["X"]
    U, I = X.shape
    K = int(g["K"])
    prop = np.maximum(X.mean(axis=0) / X.mean(axis=0).max(), 1e-5) ** 0.5   # relmf.pyx:88
    for opt in ("sgd", "adagrad", "adam"):
        for ep in (1, 2):
            W, H = oracle.reference_init(U, I, K)
            m = oracle.RelMf(W, H, opt, float(g["lr"]), float(g["wd"]), float(g["clip"]))
            for _ in range(ep):
                m.epoch(X, prop)
            assert np.array_equal(W, g[f"W_{opt}_{ep}"]), (opt, ep)
            assert np.array_equal(H, g[f"H_{opt}_{ep}"]), (opt, ep)


def test_relmf_draws_are_the_uniform_stream():
    g = golden("relmf_30x40")
    X = g["X"]
    U, I = X.shape
    W, H = oracle.reference_init(U, I, 5)
    m = oracle.RelMf(W, H, "sgd", 0.05, 0.01, 0.1)
    _, d = m.epoch(X, np.ones(I), want_draws=True)
    assert np.array_equal(d, oracle.uniform_stream(1234, U * I, U * I))


def test_metrics_vs_reference_fixture():
    g = golden("metrics")
    for r, y in enumerate(g["y"]):
        for c, k in enumerate(g["ks"]):
            assert oracle.dcg_at_k(y, int(k)) == g["dcg"][r, c]
            assert oracle.recall_at_k(y, int(k)) == g["recall"][r, c]
            assert oracle.ap_at_k(y, int(k)) == g["ap"][r, c]


@pytest.mark.parametrize("K", [8, 64])
def test_wmf_vs_lapack_restatement_unpinned(K):
    """PARITY UNPINNED by the reference (its wmf/linalg modules need cblas.h, absent here):
    the fixture comes from a numpy/LAPACK-dgesv restatement of cymf/wmf.pyx:136-174."""
    g = golden("wmf_200x300_unpinned")
    assert int(g["unpinned"]) == 1
    X = csr_from_golden(g)
    np.random.seed(4321)
    W = np.random.uniform(-0.1, 0.1, (200, K)) / K
    H = np.random.uniform(-0.1, 0.1, (300, K)) / K
    oracle.wmf_fit(X, W, H, 2, float(g["weight"]), float(g["wd"]))
    assert rel_fro(W, g[f"W_k{K}"]) < 1e-11
    assert rel_fro(H, g[f"H_k{K}"]) < 1e-11
    # empty rows are zeroed (wmf.pyx:154-156)
    empty = np.diff(X.indptr) == 0
    assert (W[empty] == 0).all()
