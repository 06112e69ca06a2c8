#!/usr/bin/env python3
"""The lock-free step kernel at C3's full size (1 M x 100 k, 98 M training interactions, K = 128) against the sequential oracle in the
reference's own shuffled order (cymf/bpr.pyx:104,160-171): epoch losses, factor norms, held-out Recall@5 (100 sampled negatives) after
three epochs of SGD, by the number of item-bucketed windows per epoch.  The oracle takes ~4 minutes on one host core.

    python tools/c3_order_fidelity.py [windows ...]        (default: 20 40 50 100; C3_OPT=adam C3_LR=0.002 for the reference's default optimizer)
(the oracle is test infrastructure: it is imported here as the checker)"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import oracle
from cymf_amd import Evaluator
from cymf_amd.bpr import BprTrainer
from test_gpu_fullsize import _c3_with_holdout
d = _c3_with_holdout()
U, I, K = d["U"], d["I"], d["K"]
ev = Evaluator(d["Xte"], d["Xtr_head"])
n = d["n_eval_users"]
E = 3
OPT, LR = os.environ.get("C3_OPT", "sgd"), float(os.environ.get("C3_LR", "0.05"))
t0 = time.time()
W, H = d["W0"].copy(), d["H0"].copy()
m = oracle.Bpr(W, H, OPT, LR, 0.01)
lo = []
for e in range(E):
    lo.append(m.epoch(d["users"], d["pos"], d["indptr"], d["cols"]))
    print(f"  oracle epoch {e + 1}: loss {lo[-1]:.4f} ({time.time()-t0:.0f}s)", flush=True)
m.close()
r0 = ev.evaluate(W[:n], H)["Recall@5"]
nW, nH = np.linalg.norm(W), np.linalg.norm(H)
print(f"{OPT} lr {LR}: oracle, the given (shuffled) order: losses {np.round(lo, 4)} |W| {nW:.1f} |H| {nH:.1f} Recall@5 {r0:.4f} ({time.time()-t0:.0f}s)", flush=True)
del W, H
for S in ([int(a) for a in sys.argv[1:]] or [20, 40, 50, 100]):
    t = BprTrainer(U, I, K, OPT, LR, 0.01, mode="throughput", steps_per_epoch=S)
    t.set_data(d["users"], d["pos"], d["indptr"], d["cols"])
    t.upload(d["W0"], d["H0"])
    ls = t.epochs(E)
    W, H = np.empty_like(d["W0"]), np.empty_like(d["H0"])
    t.download(W, H)
    t.close()
    r = ev.evaluate(W[:n], H)["Recall@5"]
    print(f"device S={S}: losses {np.round(ls, 4)} (rel {np.round(np.array(ls)/np.array(lo)-1, 4)}) |W| rel {np.linalg.norm(W)/nW-1:+.4f} |H| rel {np.linalg.norm(H)/nH-1:+.4f} Recall@5 {r:.4f} ({r-r0:+.4f})", flush=True)
ev.close()
