#!/usr/bin/env python3
"""Developer tool: WMF C4 epoch time per K, optionally with CYMF_WMF_PROBE set (1 no solve, 2 no Gramian)."""
import os
import sys
import time

import numpy as np
from scipy import sparse

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from cymf_amd import _lib, synthetic  # noqa: E402
from cymf_amd.wmf import WmfTrainer  # noqa: E402

U, I, nnz, K, seed = synthetic.CONFIGS["C4"]
rows, cols, indptr = synthetic.implicit_matrix_large(U, I, nnz, seed)
X = sparse.csr_matrix((np.ones(len(rows), dtype=np.float32), cols, indptr), shape=(U, I))
Xt = X.T.tocsr()
for Kx in [int(k) for k in (sys.argv[1:] or ["64", "128"])]:
    rs = np.random.RandomState(4321)
    W, H = rs.uniform(-0.1, 0.1, (U, Kx)) / Kx, rs.uniform(-0.1, 0.1, (I, Kx)) / Kx
    for probe in os.environ.get("PROBES", "0,1,2").split(","):
        os.environ["CYMF_WMF_PROBE"] = probe
        t = WmfTrainer(U, I, Kx, 10.0, 0.01, dtype="float32")
        t.set_data(X.indptr, X.indices, Xt.indptr, Xt.indices)
        t.upload(W, H)
        t.half_sweep(0); t.half_sweep(1)
        _lib.device_sync(0)
        res = []
        for side in (0, 1):
            t0 = time.perf_counter()
            for _ in range(3):
                t.half_sweep(side)
            _lib.device_sync(0)
            res.append((time.perf_counter() - t0) / 3 * 1e3)
        print(f"K={Kx} probe={probe}: users {res[0]:.2f} ms, items {res[1]:.2f} ms", flush=True)
        t.close()
