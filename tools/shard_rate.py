"""Per-rank step rate of a 1/8 user shard of C3 without a communicator: what a rank of an 8-GPU job does minus the exchange."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from cymf_amd import dist, synthetic  # noqa: E402
from cymf_amd.bpr import BprTrainer  # noqa: E402

world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
U, I, nnz, K, seed = synthetic.CONFIGS["C3"]
rows, cols, indptr = synthetic.implicit_matrix_large(U, I, nnz, seed)
nnz = len(rows)
perm = np.random.default_rng(4321).permutation(nnz)
lo, hi = dist.user_shards(indptr, world)[0]
mine = np.nonzero((rows[perm] >= lo) & (rows[perm] < hi))[0]
users, positives, gpos = rows[perm[mine]], cols[perm[mine]], mine.astype(np.int64)
spe = max(1, int(round(nnz / (4_000_000 * world))))
rs = np.random.RandomState(4321)
W0 = rs.uniform(-0.1, 0.1, size=(U, K)) / K
H0 = rs.uniform(-0.1, 0.1, size=(I, K)) / K
t = BprTrainer(U, I, K, "sgd", 0.05, 0.01, dtype="float32", mode="throughput", steps_per_epoch=spe)
t.set_data(users, positives, indptr.astype(np.int32), cols, gpos, nnz)
t.upload(W0, H0)
t.steps(5); t.sync()
p0, _ = t.stats()
t0 = time.perf_counter(); t.steps(50); t.sync(); dt = time.perf_counter() - t0
p1, _ = t.stats()
print(f"world {world}: shard {len(users)} triplets, {spe} steps/epoch, {dt/50*1e3:.3f} ms/step, {(p1-p0)/dt/1e9:.3f} G updates/s per GPU", flush=True)
t.close()
