#!/usr/bin/env python3
"""Eight ranks at the benchmarked shape (C3: 1M x 100k, 98M training interactions, K = 128) on ONE GPU: time-to-quality of the
user-sharded schedule against the single rank.

Eight host threads = eight ranks over the in-process local-group communicator (RCCL refuses two ranks on one device); users
sharded by nnz, item table replicated, damped item-delta sums exchanged under the next step (csrc/bpr.hip).
For steps_per_epoch S in --steps: job loss after each epoch, norm of H against the single rank's, held-out Recall@5.

    python tools/c3_ranks_table.py [--steps 3,6,12,25] [--epochs 3] [--world 8] [--rho v1,v2] [--json out.json]
--rho overrides the per-touch contraction behind the sequentialisation factors (CYMF_BPR_DELTA_RHO; `auto` = the library's)."""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from cymf_amd import Evaluator, dist  # noqa: E402
from cymf_amd.bpr import BprTrainer  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", default="3,6,12,25")
    ap.add_argument("--epochs", type=int, default=3)
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--rho", default="auto")
    ap.add_argument("--opt", default="sgd")
    ap.add_argument("--lr", type=float, default=0.05)
    ap.add_argument("--json", default=None)
    args = ap.parse_args()
    from test_gpu_fullsize import _c3_with_holdout      # the test's own data set and split
    t0 = time.time()
    d = _c3_with_holdout()
    U, I, K = d["U"], d["I"], d["K"]
    print(f"[c3 ranks] data ready in {time.time()-t0:.0f}s", flush=True)
    ev = Evaluator(d["Xte"], d["Xtr_head"])
    n = d["n_eval_users"]
    rows = []
    # The epoch losses the trainers report are ONLINE means (every triplet against the factors as they stand when it is worked --
    # for a rank: against its own replica of H, which within a step has seen only that rank's share of the updates).  The state
    # of the MODEL is measured offline: the BPR loss of the downloaded factors on a fixed sample of 2 M training triplets with
    # fixed uniform negatives (cymf/model.pyx:60's data term; the same sample for every run).
    rs = np.random.RandomState(99)
    pick = rs.randint(0, len(d["users"]), 2_000_000)
    su, si, sj = d["users"][pick], d["pos"][pick], rs.randint(0, I, len(pick))

    def offline_loss(W, H):
        out = 0.0
        for b in range(0, len(su), 250_000):
            u, i, j = su[b:b + 250_000], si[b:b + 250_000], sj[b:b + 250_000]
            x = np.einsum("nk,nk->n", W[u], H[i] - H[j])
            out += np.logaddexp(0.0, -x).sum()
        return float(out / len(su))

    one = BprTrainer(U, I, K, args.opt, args.lr, 0.01, mode="throughput", steps_per_epoch=25)
    one.set_data(d["users"], d["pos"], d["indptr"], d["cols"])
    one.upload(d["W0"], d["H0"])
    loss1 = one.epochs(args.epochs)
    W1, H1 = np.empty_like(d["W0"]), np.empty_like(d["H0"])
    one.download(W1, H1)
    one.close()
    r1 = ev.evaluate(W1[:n], H1)["Recall@5"]
    nH1 = float(np.linalg.norm(H1))
    off1 = offline_loss(W1, H1)
    rows.append({"run": "single rank, 25 steps per epoch", "loss": [float(x) for x in loss1], "offline_loss": off1, "nH": nH1, "recall5": r1})
    print(json.dumps(rows[-1]), flush=True)
    del W1

    shards = dist.user_shards(d["indptr"], args.world)
    for rho in args.rho.split(","):
        if rho == "auto":
            os.environ.pop("CYMF_BPR_DELTA_RHO", None)
        else:
            os.environ["CYMF_BPR_DELTA_RHO"] = rho
        for S in [int(s) for s in args.steps.split(",")]:
            comms = dist.Comm.local_group(args.world, I * K + 1024)
            res, err = [None] * args.world, []

            def work(r):
                try:
                    u, p, gpos = dist.shard_triplets(d["users"], d["pos"], shards[r])
                    ip, ix = dist.shard_pattern(d["indptr"], d["cols"], shards[r])
                    t = BprTrainer(U, I, K, args.opt, args.lr, 0.01, mode="throughput", steps_per_epoch=S, comm=comms[r])
                    t.set_data(u, p, ip, ix, gpos, len(d["users"]))
                    t.upload(d["W0"], d["H0"])
                    losses = t.epochs(args.epochs) * len(u)
                    W, H = np.empty_like(d["W0"]), np.empty_like(d["H0"])
                    t.download(W, H)
                    t.close()
                    lo, hi = shards[r]
                    res[r] = (losses, W[lo:hi].copy(), H if r == 0 else None)
                except BaseException as e:   # noqa: BLE001
                    err.append((r, repr(e)))

            t0 = time.time()
            threads = [threading.Thread(target=work, args=(r,)) for r in range(args.world)]
            for t in threads:
                t.start()
            for t in threads:
                t.join(timeout=900)
            for c in comms:
                c.close()
            if err:
                print("[c3 ranks] failed:", err, flush=True)
                continue
            job_loss = sum(r[0] for r in res) / len(d["users"])
            H8 = res[0][2]
            W8 = np.concatenate([r[1] for r in res])
            r8 = ev.evaluate(W8[:n], H8)["Recall@5"]
            off8 = offline_loss(W8, H8)
            rows.append({"run": f"{args.world} ranks, {S} steps per epoch, rho {rho}", "S": S, "rho": rho, "loss": [float(x) for x in job_loss],
                         "loss_rel": [float(a / b - 1) for a, b in zip(job_loss, loss1)], "offline_loss": off8, "offline_loss_rel": off8 / off1 - 1, "nH_rel": float(np.linalg.norm(H8) / nH1 - 1),
                         "recall5": r8, "recall5_diff": r8 - r1, "wall_s": time.time() - t0})
            print(json.dumps(rows[-1]), flush=True)
            del W8, H8, res
    ev.close()
    if args.json:
        json.dump(rows, open(args.json, "w"), indent=1)


if __name__ == "__main__":
    main()
