#!/usr/bin/env python3
"""Developer tool: a few WMF C4 epochs at one K (argv[1], default 128) -- the command to put under rocprofv3 --kernel-trace --stats."""
import os
import sys

import numpy as np
from scipy import sparse

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from cymf_amd import _lib, synthetic  # noqa: E402
from cymf_amd.wmf import WmfTrainer  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 128
U, I, nnz, _, seed = synthetic.CONFIGS["C4"]
rows, cols, indptr = synthetic.implicit_matrix_large(U, I, nnz, seed)
X = sparse.csr_matrix((np.ones(len(rows), dtype=np.float32), cols, indptr), shape=(U, I))
Xt = X.T.tocsr()
rs = np.random.RandomState(4321)
W, H = rs.uniform(-0.1, 0.1, (U, K)) / K, rs.uniform(-0.1, 0.1, (I, K)) / K
t = WmfTrainer(U, I, K, 10.0, 0.01, dtype="float32")
t.set_data(X.indptr, X.indices, Xt.indptr, Xt.indices)
t.upload(W, H)
for _ in range(4):
    t.half_sweep(0)
    t.half_sweep(1)
_lib.device_sync(0)
t.close()
