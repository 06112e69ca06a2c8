#!/usr/bin/env python3
"""bench.py -- BPR triplet-updates/sec at K=128 on MI355X (BASELINE.json metric), HBM roofline beside it.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Workload (config C3): synthetic 1M users x 100k items, 100M interactions
(lognormal user activity, Zipf(1) item popularity, seed 102), K=128, SGD lr=0.05 wd=0.01, fp32,
HOGWILD (throughput) mode.  A "step" = one window of the shuffled triplet order:
~4M triplets per GPU (draw negative from the mt19937 stream, skip if positive, forward, backward).
Users are sharded over the N ranks by nnz; with N > 1 every step ends with the RCCL all-reduce of
the item-factor deltas.  Inputs are resident in HBM before the timed region.

One JSON line on rank 0: value = performed triplet updates of all ranks / wall time of the K steps
(barrier + device sync on both sides, max over ranks).  `roofline` prices the dominant kernel
(bpr_step_kernel) with HIP events on its own stream: algorithmic bytes = 24K+12 per performed
triplet against the 8 TB/s HBM3E peak.  `cpu_baseline` = the oracle (a fp64
single-thread port of the reference's loop) on a bounded prefix of the same triplets, rank 0, N=1.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from cymf_amd import _lib, dist, synthetic  # noqa: E402
from cymf_amd.bpr import BprTrainer  # noqa: E402

HBM_PEAK = 8.0e12   # B/s, MI355X_MICROARCH.md chip table


def host_cpu_share():
    """CPUs this process may actually use: the cgroup quota (cpu.max) if there is one, else the affinity mask."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def log(rank, *a):
    if rank == 0:
        print("[bench]", *a, file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="C3")
    ap.add_argument("--batch", type=int, default=4_000_000, help="triplets per GPU per step")
    ap.add_argument("--optimizer", default="sgd")
    ap.add_argument("--cpu-sample", type=int, default=4_000_000, help="triplets timed on the CPU oracle (0 = skip)")
    ap.add_argument("--scale", type=float, default=1.0, help="shrink the workload (debug only; invalid as a result)")
    args = ap.parse_args()

    rank, world, local = dist.env_rank_world()
    if world != args.gpus:
        if args.gpus != 1 or world != 1:
            raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    device = local if os.environ.get("CYMF_BENCH_SAME_DEVICE") != "1" else 0   # (test hook: all ranks on device 0)
    U, I, nnz, K, seed = synthetic.CONFIGS[args.config]
    if args.scale != 1.0:
        U, nnz = max(int(U * args.scale), 1000), max(int(nnz * args.scale), 10000)
    lr, wd = 0.05, 0.01

    t0 = time.time()
    rows, cols, indptr = synthetic.implicit_matrix_large(U, I, nnz, seed)
    nnz = len(rows)
    log(rank, f"synthetic {args.config}: U={U} I={I} nnz={nnz} K={K} generated in {time.time()-t0:.1f}s")
    # the single shuffled order of the fit (cymf/bpr.pyx:104): same permutation on every rank
    perm = np.random.default_rng(4321).permutation(nnz)
    comm = None
    if world > 1:
        comm = dist.Comm.from_env(device=device)
        lo, hi = dist.user_shards(indptr, world)[rank]
        mine = np.nonzero((rows[perm] >= lo) & (rows[perm] < hi))[0]     # global positions of my triplets
        users, positives, gpos = rows[perm[mine]], cols[perm[mine]], mine.astype(np.int64)
    else:
        users, positives, gpos = rows[perm], cols[perm], None
    spe = max(1, int(round(nnz / (args.batch * world))))
    log(rank, f"rank {rank}/{world}: {len(users)} local triplets, {spe} steps/epoch (~{nnz // (spe * world)} triplets/GPU/step)")

    rs = np.random.RandomState(4321)   # the reference's init (cymf/bpr.pyx:97-101), identical on every rank
    W0 = rs.uniform(-0.1, 0.1, size=(U, K)) / K
    H0 = rs.uniform(-0.1, 0.1, size=(I, K)) / K

    trainer = BprTrainer(U, I, K, args.optimizer, lr, wd, dtype="float32", mode="throughput", device=device,
                         steps_per_epoch=spe, comm=comm)
    t0 = time.time()
    trainer.set_data(users, positives, indptr.astype(np.int32), cols, gpos, nnz)
    trainer.upload(W0, H0)
    log(rank, f"device setup {time.time()-t0:.1f}s on {_lib.device_name(device)}")

    def barrier():
        trainer.sync()
        if comm is not None:
            comm.barrier()

    trainer.steps(args.warmup)
    barrier()
    p_before, _ = trainer.stats()
    trainer.set_profiling(True)
    trainer.kernel_time()            # reset
    barrier()
    t0 = time.perf_counter()
    trainer.steps(args.steps)
    barrier()
    elapsed = time.perf_counter() - t0
    p_after, s_after = trainer.stats()
    performed_local = p_after - p_before
    k_ms, k_launches, k_slots = trainer.kernel_time()
    trainer.set_profiling(False)

    if comm is not None:
        elapsed = float(comm.allreduce(np.array([elapsed], dtype=np.float32), op="max")[0])
        # float32 all-reduce of counts: exact below 2^24 per summand, so send them in 2^20 units + remainder
        parts = np.array([performed_local // (1 << 20), performed_local % (1 << 20)], dtype=np.float32)
        tot = comm.allreduce(parts)
        performed = int(tot[0]) * (1 << 20) + int(round(float(tot[1])))
    else:
        performed = performed_local

    value = performed / elapsed
    bytes_per_triplet = {"sgd": 24 * K + 12, "adagrad": 48 * K + 12, "adam": 72 * K + 12}[args.optimizer]
    avg_launch_s = (k_ms / 1e3) / max(k_launches, 1)
    achieved = bytes_per_triplet * (performed_local / max(k_launches, 1)) / max(avg_launch_s, 1e-12)

    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")   # HBM bytes per launch from rocprofv3 --pmc
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath)).get(f"bpr_step_{args.optimizer}_K{K}")
        except Exception:
            traffic = None

    cpu = None
    if rank == 0 and world == 1 and args.cpu_sample > 0:
        import oracle   # the CPU baseline leg: the oracle is the thing timed here, never the product path
        n = min(args.cpu_sample, nnz)
        ip32, u32, p32 = indptr.astype(np.int32), users[:n].copy(), positives[:n].copy()
        # (1) the faithful sequential port, one thread
        om = oracle.Bpr(W0.copy(), H0.copy(), args.optimizer, lr, wd)
        tc = time.perf_counter()
        om.epoch(u32, p32, ip32, cols)
        dt1 = time.perf_counter() - tc
        rate1 = (n - om.skipped) / dt1
        om.close()
        # (2) the reference's HOGWILD regime (prange over lock-free W/H, cymf/bpr.pyx:162) on all host cores
        cores = min(oracle.max_threads(), host_cpu_share())
        cpu_model = "unknown CPU"
        try:
            cpu_model = next(l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name"))
        except Exception:
            pass
        om = oracle.Bpr(W0.copy(), H0.copy(), args.optimizer, lr, wd)
        reps = max(1, min(8, cores // 2))        # keep the leg at roughly the single-thread leg's duration
        tc = time.perf_counter()
        done = 0
        for _ in range(reps):
            done += om.epoch_hogwild(u32, p32, ip32, cols, cores)[1]
        dtn = time.perf_counter() - tc
        om.close()
        cpu = {"value": done / dtn, "unit": "triplet-updates/s", "cores": cores, "kind": "port",
               "sample": f"first {n} triplets of the same shuffled order x{reps}, fp64, HOGWILD on {cores} threads, {dtn:.1f}s; "
                         f"sequential 1-thread port: {rate1:.0f} triplet-updates/s ({dt1:.1f}s); host: {cpu_model}, omp_get_max_threads()={oracle.max_threads()}"}

    copy_gbps = None
    if rank == 0 and world == 1:
        try:
            copy_gbps = _lib.stream_copy_gbps(device, 1 << 30, 10)
        except Exception:
            copy_gbps = None

    if rank == 0:
        out = {
            "metric": "BPR triplet-updates/sec at K=128",
            "value": value,
            "unit": "triplet-updates/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"{args.config}: {U} users x {I} items, {nnz} interactions, K={K}, "
                                   f"{args.optimizer} lr={lr} wd={wd}, HOGWILD mode",
                       "triplets_per_gpu_per_step": nnz // (spe * world), "steps_per_epoch": spe,
                       "sharding": f"users x{world}" + (", RCCL all-reduce of item deltas per step" if world > 1 else "")},
            "roofline": {"bound": "hbm", "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK, "traffic": traffic,
                         "kernel": "bpr_step_kernel", "avg_launch_ms": 1e3 * avg_launch_s, "launches": k_launches,
                         "bytes_per_unit": bytes_per_triplet, "stream_copy_GBps": copy_gbps},
            "cpu_baseline": cpu,
            "skipped_draws": int(s_after),
        }
        print(json.dumps(out), flush=True)
    trainer.close()
    if comm is not None:
        comm.close()


if __name__ == "__main__":
    main()
