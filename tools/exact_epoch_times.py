#!/usr/bin/env python3
"""Epoch time of the exact (sequential-order) modes of RelMF and GloVe: one dataflow launch (default) against one launch per
level (CYMF_*_EXACT_LEVELS=1).   python tools/exact_epoch_times.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from cymf_amd import _lib, synthetic  # noqa: E402
from cymf_amd.glove import GloveTrainer  # noqa: E402
from cymf_amd.relmf import RelMfTrainer  # noqa: E402


def timed(run, n=3):
    run(1)
    _lib.device_sync(0)
    t0 = time.perf_counter()
    run(n)
    _lib.device_sync(0)
    return 1e3 * (time.perf_counter() - t0) / n


for flag in ("0", "1"):
    os.environ["CYMF_RELMF_EXACT_LEVELS"] = flag
    os.environ["CYMF_GLOVE_EXACT_LEVELS"] = flag
    rs = np.random.RandomState(1)
    U, I, K = 943, 1682, 20                                   # ml-100k-shaped, the reference's default K
    X = (rs.rand(U, I) < 0.03).astype(np.float64)
    prop = np.maximum(X.mean(axis=0) / X.mean(axis=0).max(), 1e-5) ** 0.5
    W, H = rs.uniform(-0.1, 0.1, (U, K)) / K, rs.uniform(-0.1, 0.1, (I, K)) / K
    t = RelMfTrainer(U, I, K, "adam", 0.001, 0.01, 0.1, mode="exact", dtype="float64")
    t.set_data(X, prop)
    t.upload(W, H)
    ms_r = timed(lambda k: t.epochs(k))
    t.close()
    V, K = 3000, 100
    C = synthetic.cooccurrence_matrix(V, 120000, 104)
    ce, cx = C.nonzero()
    p = rs.permutation(len(ce))
    ce, cx, cnt = ce[p], cx[p], C.data[p]
    g = GloveTrainer(V, V, K, 0.05, 10.0, 0.75, dtype="float64", mode="exact")
    g.set_data(ce, cx, cnt)
    g.upload(rs.uniform(-0.5, 0.5, (V, K)) / K, rs.uniform(-0.5, 0.5, (V,)) / K, rs.uniform(-0.5, 0.5, (V, K)) / K, rs.uniform(-0.5, 0.5, (V,)) / K)
    ms_g = timed(lambda k: g.epochs(k))
    g.close()
    print(f"{'one launch per level' if flag == '1' else 'dataflow (one launch)'}: RelMF 943 x 1682 K=20 adam f64 {ms_r:.1f} ms/epoch "
          f"({U * I} draws); GloVe V=3000 120k pairs K=100 f64 {ms_g:.1f} ms/epoch", flush=True)
