#!/usr/bin/env python3
"""Developer tool: timings of the secondary configs on one MI355X.
  python tools/bench_models.py [wmf] [glove] [bpr_opt] [relmf] [exact]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from cymf_amd import BPR, _lib, synthetic  # noqa: E402
from cymf_amd.bpr import BprTrainer  # noqa: E402
from cymf_amd.glove import GloveTrainer  # noqa: E402
from cymf_amd.relmf import RelMfTrainer  # noqa: E402
from cymf_amd.wmf import WmfTrainer  # noqa: E402

what = set(sys.argv[1:]) or {"wmf", "glove", "bpr_opt", "relmf", "exact"}


def init(U, I, K):
    rs = np.random.RandomState(4321)
    return rs.uniform(-0.1, 0.1, (U, K)) / K, rs.uniform(-0.1, 0.1, (I, K)) / K


if "wmf" in what:
    from scipy import sparse
    U, I, nnz, K, seed = synthetic.CONFIGS["C4"]
    rows, cols, indptr = synthetic.implicit_matrix_large(U, I, nnz, seed)
    X = sparse.csr_matrix((np.ones(len(rows), dtype=np.float32), cols, indptr), shape=(U, I))
    Xt = X.T.tocsr()
    for Kx in (64, 128):
        W, H = init(U, I, Kx)
        t = WmfTrainer(U, I, Kx, 10.0, 0.01, dtype="float32")
        t.set_data(X.indptr, X.indices, Xt.indptr, Xt.indices)
        t.upload(W, H)
        t.epochs(1)
        _lib.device_sync(0)
        t0 = time.perf_counter()
        n = 3
        t.epochs(n)
        _lib.device_sync(0)
        dt = (time.perf_counter() - t0) / n
        flops = 2 * (2 * Kx * Kx * X.nnz) + (U + I) * (Kx ** 3 / 3 + 2 * Kx * Kx)
        print(f"WMF C4 K={Kx}: {dt*1e3:.1f} ms/epoch, {flops/dt/1e12:.2f} TFLOP/s (Gramian+solve), max row {np.diff(Xt.indptr).max()}", flush=True)
        t.close()

if "glove" in what:
    V, _, nnz, K, seed = synthetic.CONFIGS["C5"]
    X = synthetic.cooccurrence_matrix(V, nnz, seed)
    ce, cx = X.nonzero()
    rs = np.random.RandomState(3)
    p = rs.permutation(len(ce))
    ce, cx, cnt = ce[p], cx[p], X.data[p]
    W = rs.uniform(-0.5, 0.5, (V, K)) / K
    b = rs.uniform(-0.5, 0.5, (V,)) / K
    Wc = rs.uniform(-0.5, 0.5, (V, K)) / K
    bc = rs.uniform(-0.5, 0.5, (V,)) / K
    t = GloveTrainer(V, V, K, 0.05, 10.0, 0.75, dtype="float32", mode="throughput")
    t.set_data(ce, cx, cnt)
    t.upload(W, b, Wc, bc)
    t.epochs(1)
    t0 = time.perf_counter()
    losses = t.epochs(3)
    dt = (time.perf_counter() - t0) / 3
    print(f"GloVe C5 K={K}: {dt*1e3:.1f} ms/epoch, {len(ce)/dt/1e6:.1f} M pairs/s, {len(ce)*(32*K+44)/dt/1e12:.2f} TB/s algorithmic "
          f"({len(ce)*(32*K+44)/dt/8e12:.3f} of HBM peak), loss {losses/len(ce)}", flush=True)
    t.close()

if "bpr_opt" in what:
    U, I, nnz, K, seed = synthetic.CONFIGS["C3"]
    rows, cols, indptr = synthetic.implicit_matrix_large(U, I, nnz, seed)
    perm = np.random.default_rng(4321).permutation(len(rows))
    users, pos = rows[perm], cols[perm]
    W, H = init(U, I, K)
    for opt, lr, bpt in (("adagrad", 0.05, 48 * K + 12), ("adam", 0.001, 72 * K + 12)):
        t = BprTrainer(U, I, K, opt, lr, 0.01, mode="throughput", steps_per_epoch=25)
        t.set_data(users, pos, indptr.astype(np.int32), cols)
        t.upload(W, H)
        t.epochs(1)
        t0 = time.perf_counter()
        losses = t.epochs(2)
        dt = (time.perf_counter() - t0) / 2
        print(f"BPR C3 {opt}: {dt*1e3:.1f} ms/epoch, {len(users)/dt/1e9:.3f} G triplets/s, {len(users)*bpt/dt/8e12:.3f} of HBM peak (algorithmic), loss {losses}", flush=True)
        t.close()

if "relmf" in what:
    U, I, K = 20000, 8000, 64
    rs = np.random.RandomState(1)
    X = (rs.rand(U, I) < 0.02).astype(np.float64)
    prop = np.maximum(X.mean(axis=0) / X.mean(axis=0).max(), 1e-5) ** 0.5
    W, H = init(U, I, K)
    t = RelMfTrainer(U, I, K, "sgd", 0.01, 0.01, 0.1, mode="throughput")
    t.set_data(X, prop)
    t.upload(W, H)
    t.epochs(1)
    t0 = time.perf_counter()
    loss = t.epochs(2)
    dt = (time.perf_counter() - t0) / 2
    print(f"RelMF {U}x{I} K={K}: {dt*1e3:.1f} ms/epoch ({U*I/dt/1e9:.3f} G draws/s, {U*I*(16*K+8)/dt/8e12:.3f} of HBM peak), loss/draw {loss[-1]/(U*I):.5f}", flush=True)
    t.close()

if "exact" in what:
    X, K = synthetic.config_matrix("C2")
    rs = np.random.RandomState(5)
    r, c = X.nonzero()
    p = rs.permutation(len(r))
    users, pos = r[p].astype(np.int32), c[p].astype(np.int32)
    W, H = init(X.shape[0], X.shape[1], K)
    for opt in ("sgd", "adam"):
        t = BprTrainer(X.shape[0], X.shape[1], K, opt, 0.01, 0.01, dtype="float64", mode="exact")
        t.set_data(users, pos, X.indptr.astype(np.int32), X.indices.astype(np.int32))
        t.upload(W, H)
        t.epochs(1)
        t.set_profiling(True)
        t.kernel_time()
        t0 = time.perf_counter()
        t.epochs(5)
        dt = (time.perf_counter() - t0) / 5
        k_ms, launches, units = t.kernel_time()
        print(f"BPR exact C2 {opt} f64: {dt*1e3:.1f} ms/epoch ({X.nnz/dt/1e6:.2f} M triplets/s, sequential-order parity mode); "
              f"device {k_ms/5:.1f} ms/epoch in {launches/5:.0f} launches", flush=True)
        t.close()

if "eval" in what:
    from scipy import sparse
    from cymf_amd import Evaluator
    U, I, nnz, K, seed = synthetic.CONFIGS["C4"]
    rows, cols, indptr = synthetic.implicit_matrix_large(U, I, nnz, seed)
    X = sparse.csr_matrix((np.ones(len(rows), dtype=np.float32), cols, indptr), shape=(U, I))
    rs = np.random.RandomState(0)
    mask = rs.rand(X.nnz) < 0.1
    Xte = X.copy(); Xte.data = Xte.data * mask; Xte.eliminate_zeros()
    Xtr = X.copy(); Xtr.data = Xtr.data * (~mask); Xtr.eliminate_zeros()
    W, H = init(U, I, K)
    ev = Evaluator(Xte, Xtr)
    t0 = time.perf_counter()
    ev.evaluate(W, H)
    t1 = time.perf_counter()
    for _ in range(3):
        r = ev.evaluate(W, H)
    t2 = time.perf_counter()
    print(f"Evaluator C4-shaped ({U} users, {Xte.nnz} held-out items, 100 negatives, K={K}): first call {1e3*(t1-t0):.0f} ms "
          f"(sequential candidate walk included), then {1e3*(t2-t1)/3:.1f} ms per call; {r}", flush=True)
    ev.close()
