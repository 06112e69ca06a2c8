#!/usr/bin/env python3
"""Developer tool: the sharded BPR job with several ranks on ONE GPU (dist.Comm.local_group, one thread per rank)
against the single-rank run on a C3-shaped problem scaled down (Zipf items, lognormal users).
  [OPT=sgd LR=0.05 RHOS=default,0.005,...] python tools/multirank_check.py [world=8] [epochs=6]"""
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from cymf_amd import dist, synthetic  # noqa: E402
from cymf_amd.bpr import BprTrainer  # noqa: E402

OPT, LR = os.environ.get("OPT", "sgd"), float(os.environ.get("LR", "0.05"))
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
epochs = int(sys.argv[2]) if len(sys.argv) > 2 else 6
U, I, nnz, K, S = 200_000, 20_000, 10_000_000, 64, 3 * 8 // max(world, 1) or 1
rows, cols, indptr = synthetic.implicit_matrix_large(U, I, nnz, 102)
nnz = len(rows)
perm = np.random.default_rng(4321).permutation(nnz)
rs = np.random.RandomState(4321)
W0 = rs.uniform(-0.1, 0.1, (U, K)) / K
H0 = rs.uniform(-0.1, 0.1, (I, K)) / K


def run(world, sync):
    os.environ["CYMF_BPR_SYNC_EXCHANGE"] = sync
    spe = max(1, int(round(nnz / (400_000 * world))))          # ~400k triplets per rank and step, like 4M at full size
    comms = dist.Comm.local_group(world, I * K + 8 * I + 64) if world > 1 else [None]
    out = [None] * world

    def fn(r):
        if world > 1:
            lo, hi = dist.user_shards(indptr, world)[r]
            mine = np.nonzero((rows[perm] >= lo) & (rows[perm] < hi))[0]
            users, pos, gpos = rows[perm[mine]], cols[perm[mine]], mine.astype(np.int64)
        else:
            users, pos, gpos = rows[perm], cols[perm], None
        t = BprTrainer(U, I, K, OPT, LR, 0.01, mode="throughput", steps_per_epoch=spe, comm=comms[r])
        t.set_data(users, pos, indptr.astype(np.int32), cols, gpos, nnz)
        t.upload(W0, H0)
        losses = [t.epochs(1)[0] * len(users) for _ in range(epochs)]
        W, H = np.empty_like(W0), np.empty_like(H0)
        t.download(W, H)
        out[r] = (np.array(losses), np.linalg.norm(H), np.abs(H).max())
        t.close()

    t0 = time.time()
    th = [threading.Thread(target=fn, args=(r,)) for r in range(world)]
    [t.start() for t in th]
    [t.join() for t in th]
    job = sum(o[0] for o in out) / nnz
    print(f"world {world} {'sync' if sync == '1' else 'overlapped'} ({spe} steps/epoch): loss/epoch {np.round(job, 4)}  |H| {out[0][1]:.2f}  max|H| {out[0][2]:.3f}  "
          f"({time.time() - t0:.1f}s)", flush=True)
    for c in comms:
        if c is not None:
            c.close()


run(1, "1")
for rho in (os.environ.get("RHOS", "default").split(",")):
    if rho != "default":
        os.environ["CYMF_BPR_DELTA_RHO"] = rho
    print("rho", rho, flush=True)
    for sync in ("1", "0"):
        run(world, sync)
