#!/usr/bin/env python3
"""Developer tool: C2 (ml-1m-shaped BPR K=64, lock-free mode) epoch time and loss trajectory under the step kernel's
tuning switches.   CYMF_BPR_MEMTYPE=0 CYMF_BPR_XCD_STRIDE=8 python tools/c2_probe.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from cymf_amd import _lib, synthetic  # noqa: E402
from cymf_amd.bpr import BprTrainer  # noqa: E402

S = int(os.environ.get("C2_STEPS", "1"))          # steps_per_epoch (windows of the shuffled order)
X, K = synthetic.config_matrix(os.environ.get("C2_CONFIG", "C2"))
U, I = X.shape
r, c = X.nonzero()
perm = np.random.RandomState(5).permutation(len(r))
rs = np.random.RandomState(4321)
W, H = rs.uniform(-0.1, 0.1, (U, K)) / K, rs.uniform(-0.1, 0.1, (I, K)) / K
for opt, lr in (("sgd", 0.05), ("adam", 0.002)):
    if opt not in os.environ.get("C2_OPTS", "sgd,adam").split(","):
        continue
    t = BprTrainer(U, I, K, opt, lr, 0.01, dtype="float32", mode="throughput", steps_per_epoch=S)
    t.set_data(r[perm], c[perm], X.indptr, X.indices)
    t.upload(W, H)
    t_S = t.steps_per_epoch()
    losses = list(t.epochs(10))
    _lib.device_sync(0)
    t0 = time.perf_counter()
    losses += list(t.epochs(20))
    _lib.device_sync(0)
    ms = 1e3 * (time.perf_counter() - t0) / 20
    Wd, Hd = np.empty_like(W), np.empty_like(H)
    t.download(Wd, Hd)
    t.set_profiling(True)
    t.kernel_time()
    t.epochs(5)
    k_ms, k_n, _ = t.kernel_time()
    t.set_profiling(False)
    t.steps(t_S)
    _lib.device_sync(0)
    t0 = time.perf_counter()
    t.steps(t_S * 50)                 # 50 epochs back to back, no loss read-back in between (as bench.py times them)
    t.sync()
    ms_nosync = 1e3 * (time.perf_counter() - t0) / 50
    t.close()
    print(f"C2 {opt} S={t_S}: {ms:.3f} ms/epoch ({X.nnz / ms / 1e6:.3f} G triplets/s), back to back {ms_nosync:.3f} ms/epoch, kernels {k_ms / 5:.3f} ms/epoch in {k_n // 5} launches; loss after 1/5/10/30 epochs "
          f"{losses[0]:.4f} {losses[4]:.4f} {losses[9]:.4f} {losses[29]:.4f}; |W| {np.linalg.norm(Wd):.3f} |H| {np.linalg.norm(Hd):.3f}", flush=True)
