"""dist.Comm.from_env with REAL processes and a fake cymf_comm_* underneath (no GPU, no RCCL): the file rendezvous,
the unique-id exchange, what happens when a rank dies before or after the rendezvous, and the shard helpers' bounds.
The fake library stands in for libcymf_hip's communicator entry points only; everything above it is the product code
that bench.py --gpus N and BPR.fit(comm=...) run under torch.distributed.run."""
import ctypes as C
import multiprocessing as mp
import os
import sys
import time

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from fakelib import FakeCommLib  # noqa: E402


def _rank_main(rank, world, port, scratch, mode, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_PORT=str(port),
                      MASTER_ADDR="127.0.0.1", TORCHELASTIC_RUN_ID="t")
    from cymf_amd import _lib, dist
    from fakelib import FakeCommLib
    fake = FakeCommLib(scratch)
    _lib.lib = lambda: fake            # the product code asks _lib.lib() for the library on every call
    try:
        if mode == "rank0_dies_before" and rank == 0:
            raise RuntimeError("rank 0 died before publishing the id")
        comm = dist.Comm.from_env(timeout=3.0 if mode.startswith("rank0_dies") else 30.0)
        assert (comm.rank, comm.world, comm.device) == (rank, world, rank)
        if mode == "rank1_raises_after" and rank == 1:
            comm.close()
            raise RuntimeError("rank 1 failed after the rendezvous")
        total = comm.allreduce(np.array([rank + 1.0, 1.0], dtype=np.float32))
        biggest = comm.allreduce(np.array([float(rank)], dtype=np.float32), op="max")
        comm.barrier()
        path = comm._rdzv_path
        comm.close()
        comm.close()                   # idempotent
        q.put((rank, "ok", total.tolist(), biggest.tolist(), path, os.path.exists(path), [c[0] for c in fake.calls]))
    except BaseException as e:         # noqa: BLE001 -- reported to the parent, which asserts on it
        q.put((rank, type(e).__name__, str(e)))


def _run(world, mode, tmp_path):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 20000 + (os.getpid() * 7 + world * 13 + len(mode)) % 20000
    procs = [ctx.Process(target=_rank_main, args=(r, world, port, str(tmp_path), mode, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = {}
    for _ in range(world):
        r = q.get(timeout=120)
        out[r[0]] = r
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return out


@pytest.mark.parametrize("world", [2, 5, 8])
def test_rendezvous_unique_id_and_collectives_with_real_processes(world, tmp_path):
    out = _run(world, "ok", tmp_path)
    for r in range(world):
        assert out[r][1] == "ok", out[r]
        assert out[r][2] == [world * (world + 1) / 2, float(world)] and out[r][3] == [float(world - 1)]
        assert out[r][6][0] == "create" and out[r][6].count("destroy") == 1     # close() twice destroys once
    # one rendezvous file for the launch, removed by rank 0's close (the other ranks may still see it when they close)
    assert len({out[r][4] for r in range(world)}) == 1 and not os.path.exists(out[0][4])


def test_rank0_dying_before_the_rendezvous_times_the_others_out(tmp_path):
    out = _run(3, "rank0_dies_before", tmp_path)
    assert out[0][1] == "RuntimeError"
    assert out[1][1] == "TimeoutError" and out[2][1] == "TimeoutError"      # no hang: the wait for the file is bounded


def test_a_rank_raising_after_the_rendezvous_fails_the_others_collective(tmp_path):
    out = _run(2, "rank1_raises_after", tmp_path)
    assert out[1][1] == "RuntimeError"
    assert out[0][1] == "CymfError"        # rank 0's all-reduce reports the missing peer (the fake times out; RCCL: async error / abort)


def test_stale_rendezvous_file_of_another_launch_is_not_picked_up(tmp_path, monkeypatch):
    """The file name carries MASTER_PORT, the run id, the world size and the launcher's pid: a file left behind by a crashed
    earlier launch (another pid) is never read."""
    sys.path.insert(0, ROOT)
    from cymf_amd import _lib, dist
    fake = FakeCommLib(str(tmp_path))
    monkeypatch.setattr(_lib, "lib", lambda: fake)
    monkeypatch.setenv("RANK", "1"); monkeypatch.setenv("WORLD_SIZE", "2"); monkeypatch.setenv("LOCAL_RANK", "1")
    monkeypatch.setenv("MASTER_PORT", "12345"); monkeypatch.setenv("TORCHELASTIC_RUN_ID", "x")
    stale = f"/tmp/cymf_amd_rdzv_12345_x_2_{os.getppid() + 1}"
    with open(stale, "wb") as f:
        f.write(b"\0" * 128)
    try:
        with pytest.raises(TimeoutError):
            dist.Comm.from_env(timeout=0.5)
    finally:
        os.remove(stale)


def test_shard_helpers_bounds():
    sys.path.insert(0, ROOT)
    from cymf_amd import dist, synthetic
    X = synthetic.implicit_matrix(1000, 300, 20000, 3)
    for world in (1, 2, 3, 8, 16):
        sh = dist.user_shards(X.indptr, world)
        assert sh[0][0] == 0 and sh[-1][1] == 1000 and all(a[1] == b[0] for a, b in zip(sh, sh[1:]))
        nnz = [X.indptr[hi] - X.indptr[lo] for lo, hi in sh]
        assert sum(nnz) == X.nnz and max(nnz) - min(nnz) <= 2 * np.diff(X.indptr).max()
        for lo, hi in sh:
            ip, ix = dist.shard_pattern(X.indptr, X.indices, (lo, hi))
            assert len(ip) == 1001 and ip[0] == 0 and ip[-1] == len(ix) == X.indptr[hi] - X.indptr[lo]
            assert (np.diff(ip)[:lo] == 0).all() and (np.diff(ip)[hi:] == 0).all()
            assert np.array_equal(np.diff(ip)[lo:hi], np.diff(X.indptr)[lo:hi])
            assert np.array_equal(ix, X.indices[X.indptr[lo]:X.indptr[hi]])
    # more ranks than users with interactions: empty shards are legal
    tiny = synthetic.implicit_matrix(3, 10, 12, 1)
    sh = dist.user_shards(tiny.indptr, 8)
    assert len(sh) == 8 and sh[-1][1] == 3 and all(lo <= hi for lo, hi in sh)
    b = dist.word_bounds(np.array([0, 0, 5, 5, 5, 9]), 10, 4)
    assert b[0] == 0 and b[-1] == 10 and (np.diff(b) >= 0).all() and len(b) == 5
