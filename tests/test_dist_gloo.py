"""N > 1 host logic on CPU: two gloo ranks run the user-sharded step loop of SURVEY.md 8e with
the ORACLE standing in for the step kernel (tests may use it), the item-factor deltas are summed
with a gloo all-reduce where the GPU build uses RCCL.  Checks: every global triplet is processed
exactly once per epoch by exactly one rank, each with ITS draw of the one global stream; the item
replicas stay bit-identical; the result equals a single-process emulation of the same schedule."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _problem():
    from cymf_amd import synthetic
    X = synthetic.implicit_matrix(400, 300, 9000, 77)
    rs = np.random.RandomState(5)
    r, c = X.nonzero()
    perm = rs.permutation(len(r))
    return X, r[perm].astype(np.int32), c[perm].astype(np.int32)


def _run_rank_schedule(rank_shards, X, users, positives, K, S, epochs, allreduce, world=None, lr=0.05, wd=0.01, corrected=True,
                       optimizer="sgd", delayed=True):
    """One process's part: rank_shards = list of (rank, shard) this process emulates.  The ranks' deltas
    are combined as the library does: H = snapshot + s_i * sum of deltas with the sequentialisation
    factors of cymf_amd.dist.delta_scale (host mirror of build_delta_scales in csrc/bpr.hip).
    delayed (the library's default, the overlapped exchange): the sum of step s is applied after step s+1 has
    been computed -- H += s_i * sum(deltas of step s) - own delta of step s -- and once more at the end."""
    import oracle
    from cymf_amd import dist
    U, I = X.shape
    N = len(users)
    dense = X.toarray() != 0
    W0, H0 = oracle.reference_init(U, I, K)
    state = {}
    for rank, shard in rank_shards:
        W, H = W0.copy(), H0.copy()
        u_l, p_l, gpos = dist.shard_triplets(users, positives, shard)
        state[rank] = dict(W=W, H=H, snap=H.copy(), base=H.copy(), m=oracle.Bpr(W, H, optimizer, lr, wd), u=u_l, p=p_l, g=gpos,
                           step=dist.step_of(gpos, S, N), seen=np.zeros(N, dtype=np.int64))
    world = world or len(rank_shards)
    step_glob = dist.step_of(np.arange(N), S, N)      # the global windows are known to every rank
    scales = []
    for s in range(S):
        n_i = np.bincount(positives[step_glob == s], minlength=I) + (step_glob == s).sum() / I
        scales.append(dist.delta_scale(n_i, world, lr, wd, optimizer) if corrected else np.ones(I))
    for ep in range(epochs):
        draws = oracle.uniform_stream(1234, I, N, skip=ep * N).astype(np.int32)
        for s in range(S):
            for rank, st in state.items():
                sel = np.nonzero(st["step"] == s)[0]
                sel = sel[np.argsort(st["p"][sel], kind="stable")]            # item-bucketed, as the kernel walks them
                neg = draws[st["g"][sel]]
                ok = ~dense[st["u"][sel], neg]
                st["loss"] = st.get("loss", 0.0) * (ep == st.get("ep", -1)) + st["m"].apply(st["u"][sel][ok], st["p"][sel][ok], neg[ok])
                st["ep"] = ep
                st["seen"][st["g"][sel]] += 1
            deltas = {rank: st["H"] - st["snap"] for rank, st in state.items()}
            total = allreduce(deltas) * scales[s][:, None]                    # sum over ALL ranks of the job, damped
            for rank, st in state.items():
                if delayed:                                                   # base: the state all ranks agree on bit for bit
                    if "pending" in st:
                        st["base"] += st["pending"]
                        st["H"][:] = st["base"] + deltas[rank]
                    st["pending"] = total
                    st["snap"][:] = st["H"]
                else:
                    st["H"][:] = st["snap"] + total
                    st["snap"][:] = st["H"]
    for st in state.values():                                                 # flush (cymf_bpr_sync / download)
        if "pending" in st:
            st["base"] += st["pending"]
            st["H"][:] = st["base"]
            del st["pending"]
    return state


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as td
    from cymf_amd import dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    td.init_process_group("gloo", rank=rank, world_size=world)
    X, users, positives = _problem()
    shards = dist.user_shards(X.indptr, world)

    def allreduce(deltas):
        t = torch.from_numpy(deltas[rank].copy())
        td.all_reduce(t)
        return t.numpy()

    st = _run_rank_schedule([(rank, shards[rank])], X, users, positives, 16, 5, 2, allreduce, world=world)[rank]
    seen = torch.from_numpy(st["seen"].copy())
    td.all_reduce(seen)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), W=st["W"], H=st["H"], seen=seen.numpy(), shard=np.array(shards[rank]))
    td.destroy_process_group()


def test_two_rank_sharded_schedule_gloo(tmp_path):
    torch = pytest.importorskip("torch")
    import torch.multiprocessing as mp
    from cymf_amd import dist
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0 = np.load(tmp_path / "rank0.npz")
    r1 = np.load(tmp_path / "rank1.npz")
    # every triplet exactly once per epoch, on exactly one rank
    assert (r0["seen"] == 2).all() and np.array_equal(r0["seen"], r1["seen"])
    # item replicas identical on the two ranks
    assert np.array_equal(r0["H"], r1["H"])
    # single-process emulation of the same two-rank schedule
    X, users, positives = _problem()
    shards = dist.user_shards(X.indptr, world)
    st = _run_rank_schedule([(0, shards[0]), (1, shards[1])], X, users, positives, 16, 5, 2,
                            lambda deltas: deltas[0] + deltas[1])
    assert np.array_equal(st[0]["H"], r0["H"]) and np.array_equal(st[1]["H"], r1["H"])
    for r, rr in ((0, r0), (1, r1)):
        lo, hi = rr["shard"]
        assert np.array_equal(st[r]["W"][lo:hi], rr["W"][lo:hi])
    # a rank never touches the user rows of the other shard
    import oracle
    W0, _ = oracle.reference_init(*X.shape, 16)
    lo1, hi1 = r1["shard"]
    assert np.array_equal(r0["W"][lo1:hi1], W0[lo1:hi1])
    # and the sharded run learns like the unsharded one (same windows, deltas summed): norms within 2%
    one = _run_rank_schedule([(0, (0, X.shape[0]))], X, users, positives, 16, 5, 2, lambda d: d[0])[0]
    assert abs(np.linalg.norm(one["H"]) / np.linalg.norm(r0["H"]) - 1) < 0.02


def test_step_windows_partition_the_global_order():
    from cymf_amd import dist
    N, S = 1003, 7
    st = dist.step_of(np.arange(N), S, N)
    assert st.min() == 0 and st.max() == S - 1 and (np.diff(st) >= 0).all()
    counts = np.bincount(st, minlength=S)
    assert counts.sum() == N and counts.max() - counts.min() <= 1


def test_delta_sum_needs_the_sequentialisation_factor():
    """Why the summed deltas are damped per item: with a strongly contracting step (many updates of a
    popular item per rank and step) a plain sum of N replicas' deltas has gain -(N-1) and diverges;
    with the factor (1 - a^N) / (N (1 - a)) the N-rank schedule stays as calm as the 1-rank one."""
    from cymf_amd import dist, synthetic
    X = synthetic.implicit_matrix(1200, 200, 30000, 78)
    rs = np.random.RandomState(6)
    r, c = X.nonzero()
    perm = rs.permutation(len(r))
    users, positives = r[perm].astype(np.int32), c[perm].astype(np.int32)
    lr, wd = 0.05, 0.2
    out = {}
    for world, corrected in ((1, True), (6, False), (6, True)):
        shards = dist.user_shards(X.indptr, world)
        st = _run_rank_schedule(list(enumerate(shards)), X, users, positives, 8, 1, 7, lambda d: sum(d.values()),
                                world=world, lr=lr, wd=wd, corrected=corrected, delayed=corrected)   # plain sum: applied at once
        out[(world, corrected)] = float(np.abs(st[0]["H"]).max())
    assert out[(6, False)] > 50 * out[(1, True)]          # plain sum: blows up
    assert out[(6, True)] < 3 * out[(1, True)] + 1e-3     # damped sum: bounded like the single rank
    # rarely touched rows keep (nearly) the plain sum: nothing is slowed down there
    s = dist.delta_scale(np.array([0.0, 5.0, 50.0]), 8, 0.05, 0.01)
    assert s[0] == 1.0 and s[1] > 0.97 and 0.75 < s[2] < 0.85
    assert dist.delta_scale(np.array([1e6]), 8, 0.05, 0.01)[0] == pytest.approx(1 / 8, rel=1e-6)
    assert (dist.delta_scale(np.array([0.0, 10.0, 1e6]), 1, 0.05, 0.01) == 1.0).all()


@pytest.mark.parametrize("optimizer,lr", [("adam", 0.05), ("adagrad", 0.5)])
def test_adaptive_optimizers_shard_with_private_state(optimizer, lr):
    """Adam / AdaGrad across ranks: the moments / accumulators of the item rows stay private to the rank (like W),
    only H is exchanged, damped with the optimizer's own per-touch contraction (dist.delta_rho).  At an aggressive
    learning rate the plain sum of six replicas' deltas diverges; the damped sum trains like a single rank."""
    from cymf_amd import dist, synthetic
    X = synthetic.implicit_matrix(1200, 200, 30000, 78)
    rs = np.random.RandomState(6)
    r, c = X.nonzero()
    perm = rs.permutation(len(r))
    users, positives = r[perm].astype(np.int32), c[perm].astype(np.int32)
    out = {}
    for world, corrected in ((1, True), (6, False), (6, True)):
        shards = dist.user_shards(X.indptr, world)
        st = _run_rank_schedule(list(enumerate(shards)), X, users, positives, 8, 1, 6, lambda d: sum(d.values()),
                                world=world, lr=lr, wd=0.01, corrected=corrected, optimizer=optimizer, delayed=corrected)
        out[(world, corrected)] = (float(np.abs(st[0]["H"]).max()), sum(s["loss"] for s in st.values()))
    assert out[(6, False)][0] > 3 * out[(1, True)][0]                 # plain sum: item factors run away
    assert out[(6, True)][0] < 2.5 * out[(1, True)][0]                # damped: same scale as the single rank
    assert out[(6, True)][1] < 0.7 * out[(6, False)][1]               # and a far lower last-epoch loss
    assert dist.delta_rho("sgd", 0.05, 0.01) == pytest.approx(0.011) and dist.delta_rho("adam", 0.001, 0.01) == pytest.approx(0.005)
    assert dist.delta_rho("adagrad", 0.05, 0.01) == pytest.approx(0.011) and dist.delta_rho("adam", 1.0, 0.01) == 0.5


def test_delta_rho_with_a_measured_data_term():
    """Host mirror of delta_scale_kernel (csrc/bpr.hip): rho = 2 lr wd + lr * c with c the measured mean of sigma'(x) |w|^2; with
    the tiny initial factors (c ~ 0) only the weight decay contracts and rarely touched rows are summed plainly; round 2's
    constant stand-in is c = 1/5; Adam has no data term."""
    from cymf_amd import dist
    lr, wd = 0.05, 0.01
    assert dist.delta_rho("sgd", lr, wd) == pytest.approx(2 * lr * wd + 0.2 * lr)
    assert dist.delta_rho("sgd", lr, wd, curvature=0.0) == pytest.approx(2 * lr * wd)
    assert dist.delta_rho("sgd", lr, wd, curvature=0.3) == pytest.approx(2 * lr * wd + 0.3 * lr)
    assert dist.delta_rho("adam", lr, wd, curvature=0.3) == pytest.approx(5 * lr)
    assert dist.delta_rho("sgd", 1.0, 1.0, curvature=10.0) == 0.5                     # capped
    n = np.array([0.0, 1.0, 100.0, 1e6])
    s0 = dist.delta_scale(n, 8, lr, wd, curvature=0.0)
    s3 = dist.delta_scale(n, 8, lr, wd, curvature=0.3)
    assert s0[0] == 1.0 and (s3 <= s0 + 1e-12).all() and s3[-1] == pytest.approx(1 / 8, rel=1e-6) and s0[1] > 0.999
