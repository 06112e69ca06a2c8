"""Several ranks on ONE GPU: the local-group communicator (cymf_comm_create_local_group; collectives meet in device
memory, one host thread per rank) runs the sharded trainers as they run under RCCL on an 8-GPU node -- user-sharded
BPR with the overlapped item-delta exchange, row-sharded WMF, central-word-sharded GloVe -- which RCCL itself
refuses to do with two ranks on one device."""
import threading

import numpy as np
import pytest

import oracle
from cymf_amd import BPR, GloVe, RelMF, WMF, dist, synthetic

pytestmark = pytest.mark.gpu


def _run_ranks(world, fn):
    out, err = [None] * world, []

    def work(r):
        try:
            out[r] = fn(r)
        except BaseException as e:   # noqa: BLE001 -- reported to the main thread
            err.append((r, repr(e)))

    threads = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not err, err
    assert all(not t.is_alive() for t in threads)
    return out


def test_local_group_allreduce_and_barrier():
    comms = dist.Comm.local_group(3, 4096)

    def fn(r):
        c = comms[r]
        a = c.allreduce(np.full(1000, float(r + 1), dtype=np.float32))
        m = c.allreduce(np.arange(10, dtype=np.float32) * (r + 1), op="max")
        c.barrier()
        return a, m

    for a, m in _run_ranks(3, fn):
        assert (a == 6.0).all() and np.array_equal(m, np.arange(10, dtype=np.float32) * 3)
    for c in comms:
        c.close()


@pytest.mark.parametrize("n", [1000, 1001, 1002, 7, 2])
def test_padded_reduce_scatter_all_gather_layout_equals_the_all_reduce(n, monkeypatch):
    """CYMF_COMM_RS_AG=1: the exchange as reduce-scatter + all-gather on shards padded to a multiple of the world
    size, n % world != 0 included -- same bits as the plain all-reduce (same rank order of the additions); a slot too small
    for the padded count is refused, not overrun."""
    world = 3
    comms = dist.Comm.local_group(world, 1002)
    rs = np.random.RandomState(n)
    data = [rs.randn(n).astype(np.float32) for _ in range(world)]

    def fn(r):
        return comms[r].allreduce(data[r])

    monkeypatch.setenv("CYMF_COMM_RS_AG", "0")
    plain = _run_ranks(world, fn)
    monkeypatch.setenv("CYMF_COMM_RS_AG", "1")
    padded = _run_ranks(world, fn)
    want = (data[0] + data[1]) + data[2]
    for r in range(world):
        assert np.array_equal(plain[r], want) and np.array_equal(padded[r], want)
    for c in comms:
        c.close()


@pytest.mark.parametrize("sync_exchange", ["0", "1"])
@pytest.mark.parametrize("optimizer,lr", [("sgd", 0.05), ("adam", 0.005)])
def test_bpr_three_ranks_on_one_gpu(optimizer, lr, sync_exchange, monkeypatch):
    monkeypatch.setenv("CYMF_BPR_SYNC_EXCHANGE", sync_exchange)
    X = synthetic.implicit_matrix(3000, 800, 90000, 91)
    K, world, S, epochs = 32, 3, 4, 10
    comms = dist.Comm.local_group(world, 800 * K + 3000 * K)        # slots hold the item-delta buffer and a rank's rows of W
    shards = dist.user_shards(X.indptr, world)

    def fn(r):
        m = BPR(K, lr, optimizer, 0.01)
        m.fit(X, num_epochs=epochs, num_threads=0, verbose=False, comm=comms[r], shard=shards[r], steps_per_epoch=S)
        return m.W, m.H, np.array(m.losses)

    res = _run_ranks(world, fn)
    one = BPR(K, lr, optimizer, 0.01)
    one.fit(X, num_epochs=epochs, num_threads=0, verbose=False, steps_per_epoch=S)
    # item replicas identical on every rank; fit() leaves the whole model on every rank (the W rows are gathered)
    W0, _ = oracle.reference_init(*X.shape, K)
    for r, (W, H, _) in enumerate(res):
        assert np.array_equal(H, res[0][1]) and np.array_equal(W, res[0][0])
        lo, hi = shards[r]
        assert not np.array_equal(W[lo:hi], W0[lo:hi].astype(np.float32).astype(np.float64))
    # and the sharded job reaches the single GPU's level: loss of the whole job and factor norms.  (The take-off from the
    # tiny initial factors is slower with ranks -- within a step a rank only feels its own share of the updates of H:
    # after 4 epochs 0.59-0.63 against 0.47 -- the curves meet afterwards.)
    n_r = np.array([X.indptr[hi] - X.indptr[lo] for lo, hi in shards], dtype=np.float64)      # each rank reports the mean over its own triplets
    job_loss = sum(w * l for w, (_, _, l) in zip(n_r / n_r.sum(), res))
    np.testing.assert_allclose(job_loss[-1], one.losses[-1], rtol=0.05)
    assert job_loss[-1] < job_loss[0]
    assert abs(np.linalg.norm(res[0][1]) / np.linalg.norm(one.H) - 1) < 0.15
    for c in comms:
        c.close()


def test_bpr_three_ranks_with_the_reduce_scatter_all_gather_exchange(monkeypatch):
    """The overlapped exchange in its CYMF_COMM_RS_AG=1 form (I * K = 800 * 31 is not a multiple of 3): replicas identical,
    the job trains."""
    monkeypatch.setenv("CYMF_COMM_RS_AG", "1")
    X = synthetic.implicit_matrix(3000, 800, 90000, 91)
    K, world = 31, 3
    comms = dist.Comm.local_group(world, 800 * K + 3000 * K + 8)
    shards = dist.user_shards(X.indptr, world)

    def fn(r):
        m = BPR(K, 0.05, "sgd", 0.01)
        m.fit(X, num_epochs=6, num_threads=0, verbose=False, comm=comms[r], shard=shards[r], steps_per_epoch=4)
        return m.H, np.array(m.losses)

    res = _run_ranks(world, fn)
    assert all(np.array_equal(H, res[0][0]) for H, _ in res) and np.isfinite(res[0][0]).all()
    assert all(l[-1] < l[0] for _, l in res)
    for c in comms:
        c.close()


def test_bpr_six_ranks_strong_contraction_stays_bounded():
    """The regime in which a plain sum of the replicas' deltas diverges (tests/test_dist_gloo.py: many updates of a popular
    item per rank and step, lr * wd = 0.01): six ranks, one step per epoch, 200 items -- the damped sums keep the
    factors at the single-GPU scale."""
    X = synthetic.implicit_matrix(1200, 200, 30000, 78)
    K, world = 8, 6
    comms = dist.Comm.local_group(world, 200 * K + 1200 * K)
    shards = dist.user_shards(X.indptr, world)

    def fn(r):
        m = BPR(K, 0.05, "sgd", 0.2)
        m.fit(X, num_epochs=7, num_threads=0, verbose=False, comm=comms[r], shard=shards[r], steps_per_epoch=1)
        return m.H

    res = _run_ranks(world, fn)
    one = BPR(K, 0.05, "sgd", 0.2)
    one.fit(X, num_epochs=7, num_threads=0, verbose=False, steps_per_epoch=1)
    assert all(np.array_equal(H, res[0]) for H in res)
    assert np.isfinite(res[0]).all() and np.abs(res[0]).max() < 3 * np.abs(one.H).max() + 1e-3
    for c in comms:
        c.close()


def test_wmf_three_ranks_on_one_gpu():
    X = synthetic.implicit_matrix(900, 400, 30000, 92)
    world = 3
    comms = dist.Comm.local_group(world, 900 * 64 + 64)

    def fn(r):
        m = WMF(64, 0.01, 10.0)
        m.fit(X, num_epochs=2, verbose=False, comm=comms[r])
        return m.W, m.H

    res = _run_ranks(world, fn)
    one = WMF(64, 0.01, 10.0)
    one.fit(X, num_epochs=2, verbose=False)
    for W, H in res:
        assert np.array_equal(W, res[0][0]) and np.array_equal(H, res[0][1])
        assert np.linalg.norm(W - one.W) <= 1e-5 * np.linalg.norm(one.W) and np.linalg.norm(H - one.H) <= 1e-5 * np.linalg.norm(one.H)
    for c in comms:
        c.close()


def test_glove_three_ranks_on_one_gpu():
    X = synthetic.cooccurrence_matrix(900, 80000, 93)
    world, K = 3, 32
    comms = dist.Comm.local_group(world, 2 * 900 * K + 4 * 900 + 64)

    def fn(r):
        m = GloVe(K, 0.05)
        m.fit(X, 3, 0, comm=comms[r], steps_per_epoch=3, seed=5)     # every rank: the same initial tables and pair order
        return m.W, np.array(m.losses)

    res = _run_ranks(world, fn)
    one = GloVe(K, 0.05)
    one.fit(X, 3, 0, steps_per_epoch=3, seed=5)
    for W, losses in res:
        assert np.array_equal(W, res[0][0]) and np.array_equal(losses, res[0][1])
    np.testing.assert_allclose(res[0][1][-1], one.losses[-1], rtol=0.1)
    assert abs(np.linalg.norm(res[0][0]) / np.linalg.norm(one.W) - 1) < 0.1
    for c in comms:
        c.close()


@pytest.mark.parametrize("optimizer,lr", [("sgd", 0.02), ("adagrad", 0.05), ("adam", 0.002)])
def test_relmf_three_ranks_on_one_gpu(optimizer, lr):
    """Users sharded over three ranks, item deltas (AdaGrad: and accumulators) summed per sub-step.  AdaGrad runs more
    epochs: its first, largest steps are the ones the damping shortens, so the sharded run leaves the small random
    start a few epochs later than the single rank and then follows it (norms within 1 % after 12 epochs)."""
    epochs = 12 if optimizer == "adagrad" else 3
    rs = np.random.RandomState(3)
    U, I, K, world = 1500, 1400, 32, 3
    Xd = (rs.rand(U, I) < 0.03).astype(np.float64)
    comms = dist.Comm.local_group(world, 2 * I * K + U * K)

    def fn(r):
        m = RelMF(K, 0.1, lr, optimizer, 0.01)
        m.fit(Xd, num_epochs=epochs, num_threads=0, comm=comms[r])
        return m.W, m.H, np.array(m.losses)

    res = _run_ranks(world, fn)
    one = RelMF(K, 0.1, lr, optimizer, 0.01)
    one.fit(Xd, num_epochs=epochs, num_threads=0)
    for W, H, losses in res:
        assert np.array_equal(W, res[0][0]) and np.array_equal(H, res[0][1]) and np.array_equal(losses, res[0][2])
    np.testing.assert_allclose(res[0][2], one.losses, rtol=3e-2)
    tol = 0.5 if optimizer == "adam" else 0.15
    assert abs(np.linalg.norm(res[0][0]) / np.linalg.norm(one.W) - 1) < 0.15 and abs(np.linalg.norm(res[0][1]) / np.linalg.norm(one.H) - 1) < tol
    for c in comms:
        c.close()
