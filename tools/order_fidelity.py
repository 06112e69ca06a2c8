#!/usr/bin/env python3
"""Throughput (lock-free) mode against the reference's OWN order.

The reference trains in the shuffled order (cymf/bpr.pyx:104,162-169); the lock-free kernels walk the triplets bucketed by
positive item inside each of `steps_per_epoch` windows of that order.  This tool fits the sequential oracle in the
shuffled order (oracle.bpr_fit) and the device in throughput mode for steps_per_epoch in {1, 4, 16, 64} on C1- and
C2-shaped data, SGD and Adam, and prints final loss, factor norms and held-out Recall@5 / DCG@5 of each -- the table of
DESIGN.md section 4 and the source of fit()'s default steps_per_epoch.

    python tools/order_fidelity.py [C1|C2|both] [--epochs N] [--steps 1,4,16,64] [--json out.json]
(the oracle is test infrastructure: it is imported here as the checker)"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import oracle  # noqa: E402
from cymf_amd import BPR, _lib, synthetic  # noqa: E402
from cymf_amd.evaluator import Evaluator  # noqa: E402


def split(X, seed):
    rs = np.random.RandomState(seed)
    mask = rs.rand(X.nnz) < 0.15
    Xte, Xtr = X.copy(), X.copy()
    Xte.data = Xte.data * mask
    Xtr.data = Xtr.data * (~mask)
    Xte.eliminate_zeros()
    Xtr.eliminate_zeros()
    return Xtr, Xte


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("which", nargs="?", default="both")
    ap.add_argument("--epochs", type=int, default=30)
    ap.add_argument("--steps", default="0,4,16,64", help="steps_per_epoch values; 0 = fit()'s default (chosen from the data)")
    ap.add_argument("--opts", default="sgd,adam")
    ap.add_argument("--json", default=None)
    args = ap.parse_args()
    steps = [int(s) for s in args.steps.split(",")]
    rows = []
    for name in (("C1", "C2") if args.which == "both" else (args.which,)):
        X, K = synthetic.config_matrix(name)
        Xtr, Xte = split(X, 3)
        ev = Evaluator(Xte, Xtr)
        for opt in args.opts.split(","):
            lr = {"sgd": 0.05, "adagrad": 0.05, "adam": 0.01 if name == "C1" else 0.002}[opt]
            t0 = time.time()
            W, H, losses = oracle.bpr_fit(Xtr, K, opt, lr, 0.01, args.epochs)
            ref = ev.evaluate(W, H)
            base = {"config": name, "opt": opt, "lr": lr, "epochs": args.epochs, "K": K}
            r0 = dict(base, run="oracle, shuffled order (the reference's)", loss=losses[-1], nW=float(np.linalg.norm(W)),
                      nH=float(np.linalg.norm(H)), recall5=ref["Recall@5"], dcg5=ref["DCG@5"], ms_per_epoch=1e3 * (time.time() - t0) / args.epochs)
            rows.append(r0)
            print(json.dumps(r0), flush=True)
            for S in steps:
                m = BPR(K, lr, opt, 0.01)
                m.fit(Xtr, num_epochs=2, num_threads=8, verbose=False, steps_per_epoch=S or None)       # warm: allocation, first launches
                m = BPR(K, lr, opt, 0.01)
                _lib.device_sync(0)
                t0 = time.time()
                m.fit(Xtr, num_epochs=args.epochs, num_threads=8, verbose=False, steps_per_epoch=S or None)   # 0 = fit()'s default
                dt = time.time() - t0
                got = ev.evaluate(m.W, m.H)
                r = dict(base, run=f"device, lock-free, steps_per_epoch={m.steps_per_epoch_}" + (" (default)" if not S else ""), S=m.steps_per_epoch_, loss=float(m.losses[-1]), nW=float(np.linalg.norm(m.W)),
                         nH=float(np.linalg.norm(m.H)), recall5=got["Recall@5"], dcg5=got["DCG@5"], fit_ms_per_epoch=1e3 * dt / args.epochs,
                         loss_rel=float(m.losses[-1] / losses[-1] - 1), nW_rel=float(np.linalg.norm(m.W) / np.linalg.norm(W) - 1),
                         nH_rel=float(np.linalg.norm(m.H) / np.linalg.norm(H) - 1), recall5_diff=got["Recall@5"] - ref["Recall@5"])
                rows.append(r)
                print(json.dumps(r), flush=True)
    if args.json:
        with open(args.json, "w") as f:
            json.dump(rows, f, indent=1)


if __name__ == "__main__":
    main()
