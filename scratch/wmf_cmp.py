import os, sys
import numpy as np
sys.path.insert(0, ".")
import oracle
from cymf_amd import WMF, synthetic
X = synthetic.implicit_matrix(500, 300, 9000, 34).tolil()
X[3] = 0; X[7] = 0; X[7, 11] = 1.0
X = X.tocsr(); X.eliminate_zeros()
for K in (64, 128):
    W0, H0 = oracle.reference_init(500, 300, K)
    Wo, Ho = W0.copy(), H0.copy()
    oracle.wmf_fit(X, Wo, Ho, 2, 10.0, 0.01)
    for flag in ("0", "1"):
        os.environ["CYMF_WMF_LDS_SOLVE"] = flag
        m = WMF(K, 0.01, 10.0)
        m.fit(X, num_epochs=2, verbose=False, dtype="float32")
        def rel(a, b): return np.linalg.norm(a - b) / np.linalg.norm(b), np.abs(a - b).max() / np.abs(b).max()
        print(K, "lds" if flag == "1" else "reg", "W", rel(m.W, Wo), "H", rel(m.H, Ho), flush=True)
