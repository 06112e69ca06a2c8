#!/usr/bin/env python3
"""Developer tool: RelMF lock-free mode (tile schedule) against the sequential oracle on 1500 x 1400, and the epoch
time at 20000 x 8000 K=64.   python tools/relmf_check.py [quality] [speed]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from cymf_amd import RelMF, _lib  # noqa: E402
from cymf_amd.relmf import RelMfTrainer  # noqa: E402

what = set(sys.argv[1:]) or {"quality", "speed"}

if "quality" in what:
    import oracle   # developer tool: the oracle is the checker here
    rs = np.random.RandomState(3)
    U, I, K = 1500, 1400, 32
    Xd = (rs.rand(U, I) < 0.03).astype(np.float64)
    prop = np.maximum(Xd.mean(axis=0) / Xd.mean(axis=0).max(), 1e-5) ** 0.5
    for opt, lr in (("sgd", 0.02), ("adagrad", 0.05), ("adam", 0.002)):
        W, H = oracle.reference_init(U, I, K)
        om = oracle.RelMf(W, H, opt, lr, 0.01, 0.1)
        want = [om.epoch(Xd, prop) for _ in range(3)]
        m = RelMF(K, 0.1, lr, opt, 0.01)
        m.fit(Xd, num_epochs=3, num_threads=0)
        print(f"{opt}: loss {np.array(m.losses) / np.array(want)} |W| {np.linalg.norm(m.W) / np.linalg.norm(W):.4f} "
              f"|H| {np.linalg.norm(m.H) / np.linalg.norm(H):.4f}", flush=True)

if "speed" in what:
    U, I, K = 20000, 8000, 64
    rs = np.random.RandomState(1)
    X = (rs.rand(U, I) < 0.02).astype(np.float64)
    prop = np.maximum(X.mean(axis=0) / X.mean(axis=0).max(), 1e-5) ** 0.5
    rs = np.random.RandomState(4321)
    W, H = rs.uniform(-0.1, 0.1, (U, K)) / K, rs.uniform(-0.1, 0.1, (I, K)) / K
    for opt in ("sgd", "adagrad", "adam"):
        t = RelMfTrainer(U, I, K, opt, 0.01, 0.01, 0.1, mode="throughput")
        t.set_data(X, prop)
        t.upload(W, H)
        t.epochs(1)
        _lib.device_sync(0)
        t0 = time.perf_counter()
        loss = t.epochs(3)
        _lib.device_sync(0)
        dt = (time.perf_counter() - t0) / 3
        print(f"RelMF {U}x{I} K={K} {opt}: {dt*1e3:.2f} ms/epoch ({U*I/dt/1e9:.2f} G draws/s, {U*I*(16*K+8)/dt/8e12:.3f} of HBM peak by "
              f"algorithmic bytes), loss/draw {loss / (U*I)}", flush=True)
        t.close()
