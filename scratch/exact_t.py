import sys, time, numpy as np
sys.path.insert(0, '.')
from cymf_amd import RelMF, GloVe, BPR, synthetic
import oracle
rs = np.random.RandomState(3)
U, I, K = 1500, 1400, 32
Xd = (rs.rand(U, I) < 0.03).astype(np.float64)
for ep in (1, 3):
    m = RelMF(K, 0.1, 0.02, "sgd", 0.01)
    t0 = time.perf_counter(); m.fit(Xd, num_epochs=ep, num_threads=1); t1 = time.perf_counter()
    print(f"RelMF exact {U}x{I} K={K} epochs={ep}: {t1-t0:.3f}s", flush=True)
prop = np.maximum(Xd.mean(axis=0) / Xd.mean(axis=0).max(), 1e-5) ** 0.5
W, H = oracle.reference_init(U, I, K)
om = oracle.RelMf(W, H, "sgd", 0.02, 0.01, 0.1)
t0 = time.perf_counter(); om.epoch(Xd, prop); print(f"oracle RelMF epoch: {time.perf_counter()-t0:.3f}s", flush=True)
X = synthetic.cooccurrence_matrix(5000, 600000, 64)
for ep in (1, 3):
    np.random.seed(1)
    g = GloVe(64, 0.05)
    t0 = time.perf_counter(); g.fit(X, ep, 1); t1 = time.perf_counter()
    print(f"GloVe exact V=5000 pairs={X.nnz} K=64 epochs={ep}: {t1-t0:.3f}s", flush=True)
