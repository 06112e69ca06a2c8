#!/usr/bin/env python3
"""bench.py -- BPR triplet-updates/sec at K=128 on MI355X (BASELINE.json metric), HBM roofline beside it.

    python bench.py --gpus N --steps K --warmup W
    N > 1 either under a launcher (python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...:
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_PORT are read from the environment) or started plainly, as above: this
    process then stays a launcher -- it starts the N ranks as fresh child processes before anything here has touched
    HIP or RCCL, passes rank 0's JSON line through, and exits non-zero (after ending the other ranks) if any rank does.

Workload (config C3): synthetic 1M users x 100k items, 100M interactions
(lognormal user activity, Zipf(1) item popularity, seed 102), K=128, SGD lr=0.05 wd=0.01, fp32,
HOGWILD (throughput) mode.  A "step" = one window of the shuffled triplet order:
~4-5M triplets per GPU (draw negative from the mt19937 stream, skip if positive, forward, backward).
The number of steps per epoch is the divisor of --steps nearest to nnz / (batch * N), so the timed
window is a whole number of epochs (`epochs_covered`) and holds exactly that many epochs' worth of
the per-epoch side-stream work (index-stream generation, skip tests): both streams are drained
before the clock starts and before it stops.
Users are sharded over the N ranks by nnz; with N > 1 every step ends with the RCCL all-reduce of
the item-factor deltas.  Inputs are resident in HBM before the timed region.

One JSON line on rank 0: value = performed triplet updates of all ranks / wall time of the K steps
(barrier + device sync on both sides, max over ranks).  `roofline` prices the dominant kernel
(bpr_step_kernel) with HIP events on its own stream: algorithmic bytes = 24K+12 per performed
triplet against the 8 TB/s HBM3E peak; beside it the bytes the kernel cannot avoid
(the positive item's row stays in registers across its run: 16K+12) and the PMC-measured bytes of
profiles/traffic.json (static file, `traffic_source`).  `cpu_baseline` = the oracle (a fp64
single-thread port of the reference's loop) on a bounded prefix of the same triplets, rank 0, N=1.
`secondary` (N=1 only): BASELINE.json's other configs -- C2 BPR K=64, C4 WMF K=64, C5 GloVe K=100 --
and RelMF 20000 x 8000 K=64, each with its own roofline object (SURVEY.md 8d figures).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from cymf_amd import _lib, dist, synthetic  # noqa: E402   (pure Python: libcymf_hip.so is only loaded by the first _lib.lib() call)
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


def nearest_divisor(k, ideal):
    """The divisor of k closest (in ratio) to ideal: steps per epoch such that k steps are whole epochs."""
    divs = [d for d in range(1, k + 1) if k % d == 0]
    return min(divs, key=lambda d: abs(np.log(d / max(ideal, 1e-9))))


def shared_dataset(rank, local, world, config, scale):
    """The synthetic matrix and the fit's shuffled order, generated ONCE per node: local rank 0 writes them to
    /dev/shm, the other ranks map them read-only (8 x 10 s of generation and 8 x the host memory otherwise)."""
    U, I, nnz, K, seed = synthetic.CONFIGS[config]
    if scale != 1.0:
        U, nnz = max(int(U * scale), 1000), max(int(nnz * scale), 10000)

    def generate():
        rows, cols, indptr = synthetic.implicit_matrix_large(U, I, nnz, seed)
        # the single shuffled order of the fit (cymf/bpr.pyx:104): same permutation on every rank
        perm = np.random.default_rng(4321).permutation(len(rows))
        return {"users": rows[perm], "positives": cols[perm], "cols": cols, "indptr": indptr.astype(np.int64)}

    if world == 1:
        return U, I, K, generate()
    tag = f"cymf_bench_{os.environ.get('MASTER_PORT', '0')}_{os.getppid()}_{config}_{scale}"
    d = os.path.join("/dev/shm", tag)
    names = ("users", "positives", "cols", "indptr")
    if local == 0:
        data = generate()
        os.makedirs(d, exist_ok=True)
        for n in names:
            np.save(os.path.join(d, n + ".npy"), data[n])
        open(os.path.join(d, "done"), "w").close()
        return U, I, K, data
    t0 = time.time()
    while not os.path.exists(os.path.join(d, "done")):
        if time.time() - t0 > 900:
            raise TimeoutError("bench: the node's dataset never appeared in /dev/shm")
        time.sleep(0.2)
    return U, I, K, {n: np.load(os.path.join(d, n + ".npy"), mmap_mode="r") for n in names}


def launch_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: be the launcher.  Nothing in this process has loaded libcymf_hip or
    called HIP (a process that has must never be replaced or forked into ranks); the N ranks are fresh interpreters
    running this file with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, as torch.distributed.run would set them.  Rank 0
    inherits stdout (its one JSON line is the launcher's), the other ranks' stdout goes to stderr.  The first rank that
    fails ends the run: the others are killed by pid and the launcher returns that rank's code."""
    import shutil
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               TORCHELASTIC_RUN_ID=f"bench{os.getpid()}", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    procs = []
    for r in range(n):
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv,
                                      env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                                      stdout=None if r == 0 else sys.stderr))
    rc = 0
    try:
        alive = list(procs)
        while alive and rc == 0:
            time.sleep(0.2)
            for p in list(alive):
                code = p.poll()
                if code is not None:
                    alive.remove(p)
                    if code != 0:
                        rc = code if code > 0 else 128 - code
                        print(f"[bench] rank {procs.index(p)} exited with {code}: ending the other ranks", file=sys.stderr, flush=True)
                        break
    except KeyboardInterrupt:
        rc = 130
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
        for p in procs:
            p.wait()
        for d in os.listdir("/dev/shm") if os.path.isdir("/dev/shm") else []:    # the node's dataset, if a rank died holding it
            if d.startswith(f"cymf_bench_{port}_{os.getpid()}_"):
                shutil.rmtree(os.path.join("/dev/shm", d), ignore_errors=True)
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="C3")
    ap.add_argument("--batch", type=int, default=5_000_000, help="target triplets per GPU per step")
    ap.add_argument("--optimizer", default="sgd")
    ap.add_argument("--cpu-sample", type=int, default=4_000_000, help="triplets timed on the CPU oracle (0 = skip)")
    ap.add_argument("--scale", type=float, default=1.0, help="shrink the workload (debug only; invalid as a result)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary configs (C2 / C4 / C5 / RelMF)")
    ap.add_argument("--pretend-world", type=int, default=1,
                    help="diagnostic (N=1 only): run rank 0's share of a P-rank job with a one-rank communicator -- the per-rank "
                         "compute cost of the sharded schedule without the exchange; the line is marked and is not a result")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        assert _lib._lib is None, "the launcher must not have loaded libcymf_hip"
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))

    rank, world, local = dist.env_rank_world()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    device = local if os.environ.get("CYMF_BENCH_SAME_DEVICE") != "1" else 0   # (test hook: all ranks on device 0)
    lr, wd = 0.05, 0.01

    t0 = time.time()
    U, I, K, data = shared_dataset(rank, local, world, args.config, args.scale)
    indptr, cols = data["indptr"], data["cols"]
    nnz = len(cols)
    log(rank, f"synthetic {args.config}: U={U} I={I} nnz={nnz} K={K} ready in {time.time()-t0:.1f}s")
    comm = None
    pretend = args.pretend_world if world == 1 and args.pretend_world > 1 else 0
    if pretend:
        world = pretend                                   # rank 0 of `pretend`; restored below for the report
    if world > 1:
        # (pretend: a real RCCL communicator of ONE rank -- its collectives are asynchronous copies in stream order, as the ranks'
        # are; the in-process local group synchronises the host twice per collective and would charge that to the schedule)
        comm = dist.Comm(0, 1, device, dist.Comm.unique_id()) if pretend else dist.Comm.from_env(device=device)
        lo, hi = dist.user_shards(indptr, world)[rank]
        all_users = data["users"]
        mine = np.nonzero((all_users >= lo) & (all_users < hi))[0]       # global positions of my triplets
        users, positives, gpos = np.asarray(all_users[mine]), np.asarray(data["positives"][mine]), mine.astype(np.int64)
        # the membership structure (cymf/bpr.pyx:146-147) of this rank's users only: other users' rows are empty
        csr_indptr, csr_indices = dist.shard_pattern(indptr, cols, (lo, hi))
    else:
        users, positives, gpos = data["users"], data["positives"], None
        csr_indptr, csr_indices = indptr.astype(np.int32), cols
    # steps per epoch: windows of ~`batch` triplets per GPU -- but about 6 per epoch at least once the job is sharded: a step ends
    # with the exchange of the item deltas, and a job of eight ranks that meets only 3 times per epoch is still 9 % behind the
    # single rank in model loss after three epochs, at 6 within 2 % (profiles/r03_c3_eight_ranks.md)
    ideal = nnz / (args.batch * world)
    spe = nearest_divisor(args.steps, ideal if world == 1 else max(ideal, 6.0))
    epochs_covered = args.steps // spe
    log(rank, f"rank {rank}/{world}: {len(users)} local triplets, {spe} steps/epoch (~{nnz // (spe * world)} triplets/GPU/step), "
              f"timed window = {epochs_covered} epoch(s)")

    rs = np.random.RandomState(4321)   # the reference's init (cymf/bpr.pyx:97-101), identical on every rank
    W0 = rs.uniform(-0.1, 0.1, size=(U, K)) / K
    H0 = rs.uniform(-0.1, 0.1, size=(I, K)) / K

    # One GPU: the windows are the library's own (what fit() runs: ~2.5 M triplets each, chosen for fidelity to the reference's order,
    # DESIGN.md 4.1) -- a bench step of ~`batch` triplets is the whole number of them nearest to that choice, so that the K steps
    # are still whole epochs.  Sharded: a window ends with the exchange, one window per step.
    windows_per_step = 1
    if comm is None:
        probe = BprTrainer(U, I, K, args.optimizer, lr, wd, dtype="float32", mode="throughput", device=device, steps_per_epoch=None)
        probe.set_data(users, positives, csr_indptr, csr_indices, gpos, nnz)
        windows_per_step = max(1, int(round(probe.steps_per_epoch() / spe)))
        probe.close()
    trainer = BprTrainer(U, I, K, args.optimizer, lr, wd, dtype="float32", mode="throughput", device=device,
                         steps_per_epoch=spe * windows_per_step, comm=comm)
    t0 = time.time()
    trainer.set_data(users, positives, csr_indptr, csr_indices, gpos, nnz)
    trainer.upload(W0, H0)
    if pretend:
        world = 1
    log(rank, f"device setup {time.time()-t0:.1f}s on {_lib.device_name(device)}")

    def barrier():
        trainer.sync()               # drains the step stream AND the side stream (index stream, skip tests)
        if comm is not None:
            comm.barrier()

    trainer.steps(args.warmup * windows_per_step)
    barrier()
    p_before, _ = trainer.stats()
    trainer.set_profiling(True)
    trainer.kernel_time()            # reset
    barrier()
    t0 = time.perf_counter()
    trainer.steps(args.steps * windows_per_step)
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
    # what the kernel cannot avoid moving: H[i] stays in registers across the item run
    min_bytes_per_triplet = {"sgd": 16 * K + 12, "adagrad": 32 * K + 12, "adam": 48 * K + 12}[args.optimizer]
    avg_launch_s = (k_ms / 1e3) / max(k_launches, 1)
    per_launch = performed_local / max(k_launches, 1)
    achieved = bytes_per_triplet * per_launch / max(avg_launch_s, 1e-12)
    achieved_min = min_bytes_per_triplet * per_launch / max(avg_launch_s, 1e-12)

    traffic, traffic_source = None, None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")   # HBM bytes per launch from rocprofv3 --pmc
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath)).get("bench", {})
            ent = tj.get(f"C3_{args.optimizer}") if (args.config == "C3" and K == 128) else None
            if ent is not None:
                prof_slots = int(tj.get("c3_slots_per_launch", 5000000))
                traffic = ent["bytes"] * (k_slots / max(k_launches, 1)) / prof_slots      # per launch of THIS run's step size
                traffic_source = (f"{tj.get('source')} (static: rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE per launch of "
                                  f"{prof_slots} slots, scaled to this run's {int(k_slots / max(k_launches, 1))} slots per launch; not measured in this run)")
        except Exception:
            traffic = None

    cpu = None
    if rank == 0 and world == 1 and args.cpu_sample > 0 and not pretend:
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
    if rank == 0 and world == 1 and not pretend:
        try:
            copy_gbps = _lib.stream_copy_gbps(device, 1 << 30, 10)
        except Exception:
            copy_gbps = None

    trainer.close()
    del trainer
    secondary = None
    if rank == 0 and world == 1 and not args.no_secondary and not pretend:
        c3_adam = None
        if args.optimizer != "adam":
            # the reference's DEFAULT optimizer (cymf/bpr.pyx:50) on the headline workload: same data, same steps, same timing
            try:
                c3_adam = timed_steps(U, I, K, "adam", 0.002, wd, device, spe, users, positives, csr_indptr, csr_indices, nnz, W0, H0,
                                      args.warmup, args.steps, args.config, windows_per_step)
            except Exception as e:   # pragma: no cover
                c3_adam = {"error": f"{type(e).__name__}: {e}"}
            log(0, f"secondary C3_bpr_adam_k128: {c3_adam.get('value', c3_adam.get('error'))}")
        del data, users, positives, W0, H0
        secondary = secondary_paths(device, args.scale)
        if c3_adam is not None:
            if "roofline" in c3_adam and args.scale == 1.0:
                t_, src_ = _static_traffic("C3_adam")
                c3_adam["roofline"]["traffic"], c3_adam["roofline"]["traffic_source"] = t_, src_
            secondary = {"C3_bpr_adam_k128": c3_adam, **secondary}

    if rank == 0 and pretend:
        print(json.dumps({"diagnostic": f"rank 0 of a pretended {pretend}-rank job, one-rank communicator (no exchange traffic)",
                          "steps": args.steps, "steps_per_epoch": spe, "ms_per_step": elapsed / args.steps * 1e3,
                          "local_triplets_per_step": performed_local / args.steps, "kernel_ms_per_step": k_ms / max(k_launches, 1),
                          "job_rate_if_exchange_hidden": performed_local * pretend / elapsed}))
        comm.close()
        return
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
            "scaling": "strong",
            "scaling_note": "fixed data set (C3: an epoch is the same 100 M triplets at every N) sharded by user; the per-GPU step stays "
                            "at ~4-6 M triplets, so the global step grows with N, the steps per epoch shrink as 1/N and the K timed "
                            "steps cover N times as many epochs (`epochs_covered`); synchronous mini-batch on the item table, SURVEY.md 8e",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "epochs_covered": epochs_covered,
            "mode_note": "lock-free (throughput) mode, the reference's num_threads > 1 regime: statistical parity only.  This headline (step kernel, "
                         f"{spe * windows_per_step} item-bucketed windows of the shuffled order per epoch" + (" = fit()'s own choice" if comm is None else "")
                         + ") is held to the sequential oracle in the reference's own shuffled order at full size -- windows of 2 M triplets: loss -1.6 %, "
                         "held-out Recall@5 -0.0023 after three epochs, profiles/r03_c3_order_fidelity.md -- and to a low-concurrency run of the same windowed "
                         "order (tests/test_gpu_fullsize.py); the small-table lines (C2) and ml-20m-shaped data to the oracle as well "
                         "(tests/test_gpu_order_fidelity.py, test_gpu_fullsize.py).  W/H parity to 1e-4 is the exact mode's (num_threads == 1), which is not what is "
                         "timed here: DESIGN.md 4, 4.1",
            "config": {"workload": f"{args.config}: {U} users x {I} items, {nnz} interactions, K={K}, "
                                   f"{args.optimizer} lr={lr} wd={wd}, HOGWILD mode",
                       "triplets_per_gpu_per_step": nnz // (spe * world), "steps_per_epoch": spe,
                       "windows_per_step": windows_per_step, "windows_per_epoch": spe * windows_per_step,
                       "sharding": f"users x{world}" + (", RCCL all-reduce of item deltas per step" if world > 1 else "")},
            "roofline": {"bound": "hbm", "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK, "traffic": traffic, "traffic_source": traffic_source,
                         "kernel": "bpr_step_kernel", "avg_launch_ms": 1e3 * avg_launch_s, "launches": k_launches,
                         "bytes_per_unit": bytes_per_triplet,
                         "basis": "algorithmic bytes of SURVEY.md 8d (3 rows read + written + indices per performed triplet)",
                         "achieved_min_traffic": achieved_min / 1e9, "frac_min_traffic": achieved_min / HBM_PEAK,
                         "min_bytes_per_unit": min_bytes_per_triplet,
                         "frac_measured_traffic": (traffic / max(avg_launch_s, 1e-12) / HBM_PEAK) if traffic else None,
                         "stream_copy_GBps": copy_gbps},
            "cpu_baseline": cpu,
            "skipped_draws": int(s_after),
            "secondary": secondary,
        }
        print(json.dumps(out), flush=True)
    if comm is not None:
        comm.close()
    if world > 1 and local == 0:   # the node's dataset in /dev/shm (every rank has mapped it by now: the barriers above)
        import shutil
        shutil.rmtree(os.path.join("/dev/shm", f"cymf_bench_{os.environ.get('MASTER_PORT', '0')}_{os.getppid()}_{args.config}_{args.scale}"),
                      ignore_errors=True)


def timed_steps(U, I, K, opt, lr, wd, device, spe, users, positives, indptr, indices, nnz, W0, H0, warmup, steps, config, windows_per_step=1):
    """N = 1: `steps` timed steps of the lock-free step kernel on the headline data with another optimizer -- the headline's own
    timing rule (both streams drained on either side, HIP events on the kernel's stream for the roofline)."""
    bpt = {"sgd": 24, "adagrad": 48, "adam": 72}[opt] * K + 12
    t = BprTrainer(U, I, K, opt, lr, wd, dtype="float32", mode="throughput", device=device, steps_per_epoch=spe * windows_per_step)
    try:
        t.set_data(users, positives, indptr, indices, None, nnz)
        t.upload(W0, H0)
        t.steps(warmup * windows_per_step)
        t.sync()
        p0, _ = t.stats()
        t.set_profiling(True)
        t.kernel_time()
        t.sync()
        t0 = time.perf_counter()
        t.steps(steps * windows_per_step)
        t.sync()
        dt = time.perf_counter() - t0
        p1, _ = t.stats()
        k_ms, k_n, _ = t.kernel_time()
    finally:
        t.close()
    a = bpt * (p1 - p0) / max(k_n, 1) / max(k_ms / 1e3 / max(k_n, 1), 1e-12)
    return {"workload": f"{config}: {U} x {I}, {nnz} interactions, K={K}, {opt} lr={lr} wd={wd}, lock-free mode, {spe} steps of {windows_per_step} window(s) per epoch",
            "value": (p1 - p0) / dt, "unit": "triplet-updates/s", "ms": 1e3 * dt / steps, "kernel_ms": k_ms / max(k_n, 1),
            "roofline": {"bound": "hbm", "achieved": a / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": a / HBM_PEAK, "bytes_per_unit": bpt,
                         "kernel": "bpr_step_kernel", "traffic": None, "timing": "HIP events on the kernel's stream, average launch"}}


# ------------------------------------------------------------------------------------------------ secondary configs
MFMA_F32_PEAK = 157.3e12   # FLOP/s, MI355X_MICROARCH.md chip table (fp32 matrix = fp32 vector rate)


def _timed_epochs(run, n, device):
    """ms per epoch: device-synchronised wall time of ONE call run(n) (n epochs back to back, as fit() runs them) after a
    warm-up call run(1) -- n separate calls would put a host round trip (loss read-back) between the epochs, which on a
    busy host moved the GloVe figure between 18 and 22 ms."""
    run(1)
    _lib.device_sync(device)
    t0 = time.perf_counter()
    run(n)
    _lib.device_sync(device)
    return 1e3 * (time.perf_counter() - t0) / n


def _static_traffic(key):
    """HBM bytes of a secondary workload from profiles/traffic.json (rocprofv3 --pmc passes over this very command,
    tools/profile_bench.sh; static file, not measured in this run): (bytes, what the figure covers)."""
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json"))).get("bench", {})
        e = tj.get(key)
        return (e["bytes"], f"{e['unit']}; {tj.get('source')}, not measured in this run") if e else (None, None)
    except Exception:
        return None, None


def _hbm_roofline(units, bytes_per_unit, ms, kernel):
    a = units * bytes_per_unit / (ms * 1e-3)
    return {"bound": "hbm", "achieved": a / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": a / HBM_PEAK,
            "bytes_per_unit": bytes_per_unit, "kernel": kernel, "traffic": None,
            "timing": "device-synchronised wall time per epoch (all kernels of the epoch, launch gaps included)"}


def secondary_paths(device, scale=1.0):
    """BASELINE.json configs 2, 4, 5 and RelMF on the same GPU, one object each: value, unit, ms per epoch and a
    roofline priced with SURVEY.md 8d's algorithmic figures.  Failures are recorded, never raised: the headline stands."""
    from scipy import sparse
    from cymf_amd.glove import GloveTrainer
    from cymf_amd.relmf import RelMfTrainer
    from cymf_amd.wmf import WmfTrainer
    out = {}

    def init(U, I, K):
        rs = np.random.RandomState(4321)
        return rs.uniform(-0.1, 0.1, (U, K)) / K, rs.uniform(-0.1, 0.1, (I, K)) / K

    def attempt(name, fn):
        t0 = time.time()
        try:
            out[name] = fn()
        except Exception as e:   # pragma: no cover
            out[name] = {"error": f"{type(e).__name__}: {e}"}
        log(0, f"secondary {name}: {time.time()-t0:.1f}s {out[name].get('value', out[name].get('error'))}")

    def c2_bpr(opt="sgd", lr=0.05):
        # BASELINE config 2: MovieLens-1M-shaped BPR K=64 fp32 (cymf/bpr.pyx:160-171), lock-free mode, as fit(num_threads != 1)
        # runs it: steps_per_epoch chosen from the data (windows of the shuffled order), one launch per epoch on this table size
        X, K = synthetic.config_matrix("C2")
        U, I = X.shape
        r, c = X.nonzero()
        perm = np.random.RandomState(5).permutation(len(r))
        W, H = init(U, I, K)
        t = BprTrainer(U, I, K, opt, lr, 0.01, dtype="float32", mode="throughput", device=device, steps_per_epoch=None)
        t.set_data(r[perm], c[perm], X.indptr, X.indices)
        S = t.steps_per_epoch()
        t.upload(W, H)
        t.epochs(3)
        p0, _ = t.stats()
        t.set_profiling(True)
        t.kernel_time()
        n = 50
        ms = _timed_epochs(lambda k: t.steps(k * S), n, device)
        p1, _ = t.stats()
        k_ms, k_n, _ = t.kernel_time()
        t.close()
        per_epoch = (p1 - p0) / (n + 1)
        bpt = {"sgd": 24, "adagrad": 48, "adam": 72}[opt] * K + 12
        k_epoch_ms = k_ms / (n + 1)                      # all launches of an epoch (one, with the group kernel)
        rl = _hbm_roofline(per_epoch, bpt, k_epoch_ms, "bpr_group_kernel" if k_n <= n + 1 else "bpr_step_kernel")
        rl["timing"] = "HIP events on the kernel's stream, launches of one epoch"
        rl["launches_per_epoch"] = k_n / (n + 1)
        # what bounds it: every write-back is a float-atomic delta executed at the memory side (~1.3 TB/s chip-wide,
        # MI355X_MICROARCH.md 'Global float atomics'): W[u] and H[j] per triplet, H[i] once per run of its item inside a block
        rl["atomic_bound_note"] = ("bound by the memory-side float-atomic rate (~1300 GB/s), not by HBM: >= 2 rows of "
                                   f"{4 * K} B per triplet are added atomically")
        rl["atomic_GBps_min"] = per_epoch * 2 * 4 * K * (2 if opt == "adagrad" else (1.5 if opt == "adam" else 1)) / (k_epoch_ms * 1e-3) / 1e9
        rl["traffic"], rl["traffic_source"] = _static_traffic("C2_" + opt)
        try:   # the roofline this kernel actually runs against: WRITE_SIZE of its launch (its atomics; Adam: + the moment stores) over the ~1.3 TB/s atomic rate
            wb = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))["bench"]["C2_" + opt]["write_bytes"]
            rl["atomic_roofline"] = {"bound": "memory-side float atomics", "achieved": wb / (k_epoch_ms * 1e-3) / 1e9, "peak": 1300.0, "unit": "GB/s",
                                     "frac": wb / (k_epoch_ms * 1e-3) / 1.3e12, "bytes_per_launch": wb,
                                     "basis": "rocprofv3 --pmc WRITE_SIZE per launch (static, profiles/traffic.json) over this run's kernel time; peak: MI355X_MICROARCH.md 'Global float atomics'"}
        except Exception:
            pass
        return {"workload": f"C2: {U} x {I}, {X.nnz} interactions, K={K}, {opt} lr={lr}, lock-free mode, steps_per_epoch={S} (fit()'s default)",
                "value": per_epoch / (ms * 1e-3), "unit": "triplet-updates/s", "ms": ms, "kernel_ms": k_epoch_ms, "steps_per_epoch": S,
                "roofline": rl}

    def c4_wmf():
        # BASELINE config 4: WMF ALS K=64 on ml-20m-shaped data (cymf/wmf.pyx:150-171)
        U, I, nnz, K, seed = synthetic.CONFIGS["C4"]
        if scale != 1.0:
            U, nnz = max(int(U * scale), 1000), max(int(nnz * scale), 10000)
        rows, cols, indptr = synthetic.implicit_matrix_large(U, I, nnz, seed)
        X = sparse.csr_matrix((np.ones(len(rows), dtype=np.float32), cols, indptr), shape=(U, I))
        Xt = X.T.tocsr()
        res = {}
        for Kx in (64, 128):
            W, H = init(U, I, Kx)
            t = WmfTrainer(U, I, Kx, 10.0, 0.01, dtype="float32", device=device)
            t.set_data(X.indptr, X.indices, Xt.indptr, Xt.indices)
            t.upload(W, H)
            ms = _timed_epochs(lambda k: t.epochs(k), 5, device)
            t.close()
            # 2 (2 K^2 nnz) Gramian + (U+I)(2/3 K^3 + 2 K^2) solve + 2 (U+I) K^2 YtY
            flops = 2 * (2 * Kx * Kx * X.nnz) + (U + I) * (2.0 / 3.0 * Kx ** 3 + 2 * Kx * Kx) + 2 * (U + I) * Kx * Kx
            a = flops / (ms * 1e-3)
            res[Kx] = {"workload": f"C4: {U} x {I}, {X.nnz} entries, K={Kx}, weight 10, f32", "value": 1e3 / ms, "unit": "epochs/s", "ms": ms,
                       "roofline": {"bound": "mfma", "achieved": a / 1e12, "peak": MFMA_F32_PEAK / 1e12, "unit": "TFLOP/s",
                                    "frac": a / MFMA_F32_PEAK, "flops_per_epoch": flops,
                                    "kernel": ("wmf_row_reg_kernel<2,1>" if Kx == 64 else "wmf_row_blk_kernel<4>") + " (+ wmf_seg / wmf_long_reg / YtY kernels)",
                                    "gather_GBps": 2 * X.nnz * (4 * Kx + 4) / (ms * 1e-3) / 1e9,
                                    "traffic": _static_traffic(f"C4_k{Kx}")[0] if scale == 1.0 else None, "traffic_source": _static_traffic(f"C4_k{Kx}")[1],
                                    "flops_note": "SURVEY.md 8d prices the full K x K Gramian (2 K^2 nnz per half-sweep); the kernels form the upper tiles only "
                                                  f"({Kx // 32 * (Kx // 32 + 1) // 2} of {(Kx // 32) ** 2}), so MFMA-executed flops are that fraction of the Gramian term",
                                    "bound_note": "f32 MFMA and VALU instructions take turns on a SIMD (tools/micro/mfma_valu_mix.hip): the floor is MFMA cycles plus four cycles "
                                                  "per vector instruction, not the MFMA peak (K=64 row kernel: matrix pipe busy 0.485 + VALU issue busy 0.41 of the cycles, "
                                                  "profiles/r03_wmf_k64_analysis.md)",
                                    "timing": "device-synchronised wall time per epoch (both half-sweeps, all kernels)"}}
        r = res[64]
        r["k128"] = res[128]
        return r

    def c5_glove():
        # BASELINE config 5: GloVe K=100 on a text8-shaped co-occurrence COO (cymf/glove.pyx:149-156)
        V, _, nnz, K, seed = synthetic.CONFIGS["C5"]
        if scale != 1.0:
            nnz = max(int(nnz * scale), 10000)
        X = synthetic.cooccurrence_matrix(V, nnz, seed)
        ce, cx = X.nonzero()
        rs = np.random.RandomState(3)
        p = rs.permutation(len(ce))
        ce, cx, cnt = ce[p], cx[p], X.data[p]
        W, b = rs.uniform(-0.5, 0.5, (V, K)) / K, rs.uniform(-0.5, 0.5, (V,)) / K
        Wc, bc = rs.uniform(-0.5, 0.5, (V, K)) / K, rs.uniform(-0.5, 0.5, (V,)) / K
        t = GloveTrainer(V, V, K, 0.05, 10.0, 0.75, dtype="float32", mode="throughput", device=device)
        t.set_data(ce, cx, cnt)
        t.upload(W, b, Wc, bc)
        ms = _timed_epochs(lambda k: t.epochs(k), 5, device)
        t.close()
        return {"workload": f"C5: V={V}, {len(ce)} pairs, K={K}, AdaGrad lr 0.05, lock-free mode", "value": len(ce) / (ms * 1e-3),
                "unit": "pairs/s", "ms": ms, "roofline": dict(_hbm_roofline(len(ce), 32 * K + 44, ms, "glove_step_kernel"),
                                                               traffic=_static_traffic("C5_glove")[0] if scale == 1.0 else None,
                                                               traffic_source=_static_traffic("C5_glove")[1])}

    def relmf():
        # RelMF (cymf/relmf.pyx:142-148): U*I uniform cell draws per epoch
        U, I, K = (20000, 8000, 64) if scale == 1.0 else (2000, 800, 64)
        rs = np.random.RandomState(1)
        X = (rs.rand(U, I) < 0.02).astype(np.float64)
        prop = np.maximum(X.mean(axis=0) / X.mean(axis=0).max(), 1e-5) ** 0.5
        W, H = init(U, I, K)
        t = RelMfTrainer(U, I, K, "sgd", 0.01, 0.01, 0.1, mode="throughput", device=device)
        t.set_data(X, prop)
        t.upload(W, H)
        ms = _timed_epochs(lambda k: t.epochs(k), 4, device)
        t.close()
        return {"workload": f"RelMF {U} x {I} dense, K={K}, sgd, {U*I} draws per epoch, lock-free mode", "value": U * I / (ms * 1e-3),
                "unit": "draws/s", "ms": ms,
                "roofline": dict(_hbm_roofline(U * I, 16 * K + 8, ms, "relmf_tile_kernel (+ tile_count / tile_scatter / index-stream kernels)"),
                                 traffic=_static_traffic("RelMF")[0] if scale == 1.0 else None, traffic_source=_static_traffic("RelMF")[1],
                                 bound_note="the algorithmic-byte figure is what SURVEY.md 8d defines, not a utilisation: a tile's item rows live in LDS and reach HBM "
                                            "once per tile (~31 draws per row), so real HBM traffic is ~1/10 of it (`traffic`) and the figure can exceed 1; the tile kernel is "
                                            "bound by instruction issue and LDS latency (LDS read -> dot -> DPP sum -> integer LDS atomic add, ~100 VALU wave-instructions "
                                            "per step of four draws; two wavefronts per SIMD keep the VALU busy), beside ~6 ms of index-stream and bucketing kernels per epoch")}

    attempt("C2_bpr_k64", c2_bpr)
    attempt("C2_bpr_adam_k64", lambda: c2_bpr("adam", 0.002))     # the reference's default optimizer (cymf/bpr.pyx:50)
    attempt("C4_wmf_k64", c4_wmf)
    attempt("C5_glove_k100", c5_glove)
    attempt("relmf_20000x8000_k64", relmf)
    return out


if __name__ == "__main__":
    main()
