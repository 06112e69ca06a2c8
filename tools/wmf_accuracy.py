#!/usr/bin/env python3
"""f32 WMF solvers against the f64 oracle on the small ill-conditioned problem of test_wmf_register_and_lds_solvers_agree
(500 x 300, 9000 entries, K = 128 > most row lengths: the systems are lambda-dominated): max |x - x_f64| / max |x_f64| after
two epochs for the blocked elimination, the register Gauss-Jordan and the in-LDS Cholesky.  Test tooling (uses oracle/)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import oracle  # noqa: E402
from cymf_amd import synthetic  # noqa: E402
from cymf_amd.wmf import WmfTrainer  # noqa: E402

for K in [int(a) for a in sys.argv[1:]] or [96, 128]:
    X = synthetic.implicit_matrix(500, 300, 9000, 34).tocsr()
    Xt = X.T.tocsr()
    W0, H0 = oracle.reference_init(500, 300, K)
    W, H = W0.copy(), H0.copy()
    for _ in range(2):
        oracle.wmf_half_sweep(X.indptr, X.indices, W, H, 10.0, 0.01)
        oracle.wmf_half_sweep(Xt.indptr, Xt.indices, H, W, 10.0, 0.01)
    for name, env in (("blocked elimination", {"CYMF_WMF_BLOCKED": "1"}), ("register Gauss-Jordan", {"CYMF_WMF_BLOCKED": "0"}),
                      ("in-LDS Cholesky", {"CYMF_WMF_LDS_SOLVE": "1"})):
        for k in ("CYMF_WMF_BLOCKED", "CYMF_WMF_LDS_SOLVE"):
            os.environ.pop(k, None)
        os.environ.update(env)
        t = WmfTrainer(500, 300, K, 10.0, 0.01, dtype="float32")
        t.set_data(X.indptr, X.indices, Xt.indptr, Xt.indices)
        t.upload(W0, H0)
        t.epochs(2)
        gW, gH = np.empty_like(W0), np.empty_like(H0)
        t.download(gW, gH)
        t.close()
        print(f"K={K} {name:24s} W {np.abs(gW - W).max() / np.abs(W).max():.2e}  H {np.abs(gH - H).max() / np.abs(H).max():.2e}", flush=True)
