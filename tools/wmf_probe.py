#!/usr/bin/env python3
"""WMF C4 epoch time for the given K (default 128), f32: `python3 tools/wmf_probe.py 96 128` (on the GPU box; wrap in
rocprofv3 --kernel-trace --stats for the per-kernel split: the longer wmf_row_* call is the user sweep)."""
import os
import sys
import time

import numpy as np
from scipy import sparse

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from cymf_amd import _lib, synthetic  # noqa: E402
from cymf_amd.wmf import WmfTrainer  # noqa: E402

U, I, nnz, _, seed = synthetic.CONFIGS["C4"]
rows, cols, indptr = synthetic.implicit_matrix_large(U, I, nnz, seed)
X = sparse.csr_matrix((np.ones(len(rows), dtype=np.float32), cols, indptr), shape=(U, I))
Xt = X.T.tocsr()
for K in [int(a) for a in sys.argv[1:]] or [128]:
    rs = np.random.RandomState(4321)
    W, H = rs.uniform(-0.1, 0.1, (U, K)) / K, rs.uniform(-0.1, 0.1, (I, K)) / K
    t = WmfTrainer(U, I, K, 10.0, 0.01, dtype="float32")
    t.set_data(X.indptr, X.indices, Xt.indptr, Xt.indices)
    t.upload(W, H)
    t.epochs(1)
    _lib.device_sync(0)
    t0 = time.perf_counter()
    n = 3
    t.epochs(n)
    _lib.device_sync(0)
    dt = (time.perf_counter() - t0) / n
    flops = 2 * (2 * K * K * X.nnz) + (U + I) * (K ** 3 / 3 + 2 * K * K)
    print(f"WMF C4 K={K}: {dt*1e3:.2f} ms/epoch, {flops/dt/1e12:.1f} TFLOP/s by SURVEY 8d flops ({flops/dt/157.3e12:.3f} of the f32 MFMA peak)", flush=True)
    t.close()
