#!/usr/bin/env python3
"""Developer tool: time bpr_step_kernel on the C3 workload under different launch knobs
(environment variables read by cymf_bpr_create).  python tools/sweep_step.py 'A=1,B=2' 'A=3' ..."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from cymf_amd import synthetic  # noqa: E402
from cymf_amd.bpr import BprTrainer  # noqa: E402

scale = float(os.environ.get("SWEEP_SCALE", "1.0"))
U, I, nnz, K, seed = synthetic.CONFIGS["C3"]
U, nnz = int(U * scale), int(nnz * scale)
K = int(os.environ.get("SWEEP_K", K))
opt = os.environ.get("SWEEP_OPT", "sgd")
rows, cols, indptr = synthetic.implicit_matrix_large(U, I, nnz, seed)
nnz = len(rows)
perm = np.random.default_rng(4321).permutation(nnz)
users, positives = rows[perm], cols[perm]
rs = np.random.RandomState(4321)
W0 = rs.uniform(-0.1, 0.1, size=(U, K)) / K
H0 = rs.uniform(-0.1, 0.1, size=(I, K)) / K
spe = max(1, round(nnz / 4_000_000))
for cfg in sys.argv[1:] or [""]:
    env = dict(kv.split("=") for kv in cfg.split(",") if kv)
    for k, v in env.items():
        os.environ[k] = v
    t = BprTrainer(U, I, K, opt, 0.05, 0.01, dtype="float32", mode="throughput", steps_per_epoch=spe)
    t.set_data(users, positives, indptr.astype(np.int32), cols)
    t.upload(W0, H0)
    t.steps(3)
    t.sync()
    t.set_profiling(True)
    t.kernel_time()
    p0, _ = t.stats()
    t0 = time.perf_counter()
    t.steps(spe)
    t.sync()
    dt = time.perf_counter() - t0
    p1, _ = t.stats()
    ms, n, slots = t.kernel_time()
    loss = t.steps(spe, want_loss=True) / nnz
    print(f"{cfg:60s} kernel {ms/n:7.3f} ms/launch  {(p1-p0)/(ms/1e3)/1e9:6.3f} G/s in-kernel  {(p1-p0)/dt/1e9:6.3f} G/s wall  loss2 {loss:.5f}", flush=True)
    t.close()
    for k in env:
        os.environ.pop(k, None)
