"""bench.py's host side for N > 1, without a GPU: the node-wide dataset in /dev/shm (local rank 0 generates, the others map),
the user shards, the per-rank membership pattern and the whole-epoch step count."""
import multiprocessing as mp
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _rank(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_PORT=str(port))
    import bench
    from cymf_amd import dist
    U, I, K, data = bench.shared_dataset(rank, rank, world, "C2", 0.2)
    indptr, cols = data["indptr"], data["cols"]
    lo, hi = dist.user_shards(indptr, world)[rank]
    users = data["users"]
    mine = np.nonzero((np.asarray(users) >= lo) & (np.asarray(users) < hi))[0]
    ip, ix = dist.shard_pattern(indptr, cols, (lo, hi))
    q.put((rank, U, I, K, int(len(cols)), int(np.asarray(users, dtype=np.int64).sum()), int(np.asarray(data["positives"], dtype=np.int64).sum()),
           int(len(mine)), int(ip[-1]), type(users).__name__, (lo, hi)))


def test_node_wide_dataset_is_generated_once_and_shards_partition_it():
    world, port = 3, 23000 + os.getpid() % 5000
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rank, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # the same data everywhere (shapes, checksums of the shuffled order), mapped -- not regenerated -- on the other ranks
    assert len({o[1:7] for o in out}) == 1
    assert out[0][9] == "ndarray" and out[1][9] == "memmap" and out[2][9] == "memmap"
    # the ranks' triplets and membership rows partition the whole
    nnz = out[0][4]
    assert sum(o[7] for o in out) == nnz and sum(o[8] for o in out) == nnz
    assert out[0][10][0] == 0 and out[-1][10][1] == out[0][1] and all(a[10][1] == b[10][0] for a, b in zip(out, out[1:]))
    import shutil
    shutil.rmtree(os.path.join("/dev/shm", f"cymf_bench_{port}_{os.getpid()}_C2_0.2"), ignore_errors=True)


def test_steps_per_epoch_divide_the_timed_window():
    sys.path.insert(0, ROOT)
    import bench
    for steps, ideal, want in ((20, 20.0, 20), (20, 12.5, 10), (20, 6.25, 5), (20, 2.5, 2), (50, 25.0, 25), (6, 0.4, 1), (40, 20.0, 20)):
        d = bench.nearest_divisor(steps, ideal)
        assert d == want and steps % d == 0
