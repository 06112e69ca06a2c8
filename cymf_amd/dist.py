"""One process per GPU: RCCL communicator over xGMI + user-range sharding.

The reference is single-process; this is the only exchange step the user-sharded design adds
(sum of item-factor deltas after every step).  Rendezvous is a file in /tmp keyed by
MASTER_PORT (single node, as launched by `python -m torch.distributed.run --nnodes=1 ...`,
whose env vars RANK / LOCAL_RANK / WORLD_SIZE / MASTER_PORT are read) -- no torch import."""
import ctypes as C
import os
import time

import numpy as np

from . import _lib


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def user_shards(indptr, world):
    """Contiguous user ranges balanced by nnz: [(lo, hi)] * world."""
    indptr = np.asarray(indptr, dtype=np.int64)
    U = len(indptr) - 1
    nnz = indptr[-1]
    cuts = [0]
    for r in range(1, world):
        cuts.append(int(np.searchsorted(indptr, nnz * r // world, side="left")))
    cuts.append(U)
    cuts = np.maximum.accumulate(np.minimum(cuts, U))
    return [(int(cuts[r]), int(cuts[r + 1])) for r in range(world)]


def shard_triplets(users, positives, shard):
    """This rank's triplets of the global shuffled order (its users only) and their positions in
    that order, so that triplet l of epoch e consumes draw e*N_global + l on whichever rank holds it."""
    lo, hi = shard
    users = np.asarray(users)
    keep = np.nonzero((users >= lo) & (users < hi))[0]
    return users[keep], np.asarray(positives)[keep], keep.astype(np.int64)


def shard_pattern(indptr, indices, shard):
    """CSR pattern of X restricted to this rank's users: the rows of the other users are empty.  The membership
    structure (cymf/bpr.pyx:146-147) is only ever asked about a rank's own users, so its device form (a hash set of
    (user, item) pairs, 2 x 8 bytes per interaction) shrinks by the world size."""
    lo, hi = shard
    indptr = np.asarray(indptr, dtype=np.int64)
    out = np.zeros(len(indptr), dtype=np.int64)
    out[lo:hi + 1] = indptr[lo:hi + 1] - indptr[lo]
    out[hi + 1:] = out[hi]
    return out.astype(np.int32), np.ascontiguousarray(np.asarray(indices)[indptr[lo]:indptr[hi]], dtype=np.int32)


def step_of(global_pos, steps_per_epoch, n_global):
    """Step (window of the global order) a triplet belongs to; mirrors build_throughput_layout in csrc/bpr.hip."""
    return (np.asarray(global_pos, dtype=np.int64) * int(steps_per_epoch)) // max(int(n_global), 1)


def word_bounds(central, n_words, world):
    """Contiguous central-word ranges with about equal numbers of pairs: bounds[r] .. bounds[r+1] is rank r's
    (GloVe sharding).  Returns an int64 array of world + 1 entries from 0 to n_words."""
    counts = np.bincount(np.asarray(central, dtype=np.int64), minlength=n_words)
    cum = np.concatenate([[0], np.cumsum(counts)])
    bounds = np.searchsorted(cum, cum[-1] * np.arange(1, world) / world, side="left")
    return np.concatenate([[0], np.minimum(bounds, n_words), [n_words]]).astype(np.int64)


def delta_rho(optimizer, lr, wd, curvature=None):
    """Per-touch contraction assumed for a replica of an item row (host mirror of build_step_counts / delta_scale_kernel,
    csrc/bpr.hip).  curvature = the job-wide mean of sigma'(x) |w|^2 over a step's triplets, which the device measures at
    the start of every step (bpr_curvature_kernel); None = round 2's constant stand-in lr / 5 (the CPU emulations)."""
    rho = 2.0 * lr * wd + (0.2 * lr if curvature is None else lr * curvature)
    if optimizer == "adam":
        rho = 5.0 * lr
    return min(0.5, rho)


def delta_scale(n_i, world, lr, wd, optimizer="sgd", curvature=None):
    """Sequentialisation factor of the summed item-factor deltas (host mirror of delta_scale_kernel in
    csrc/bpr.hip): n_i = updates of the item in the step over all ranks."""
    base = 1.0 - delta_rho(optimizer, lr, wd, curvature)
    a = base ** (np.asarray(n_i, dtype=np.float64) / world)
    small = a >= 1.0 - 1e-12
    return np.where(small, 1.0, (1.0 - a ** world) / (world * np.where(small, 1.0, 1.0 - a)))


class Comm:
    """RCCL communicator handle (cymf_comm_*)."""

    def __init__(self, rank, world, device, unique_id):
        self.L = _lib.lib()
        self.rank, self.world, self.device = rank, world, device
        self.h = C.c_void_p()
        buf = C.create_string_buffer(bytes(unique_id), _lib.UNIQUE_ID_BYTES)
        _lib.check(self.L.cymf_comm_create(C.byref(self.h), buf, rank, world, device))
        _lib.track(self, last=True)

    @classmethod
    def local_group(cls, world, max_floats, device=0):
        """world communicators inside this process on one device (cymf_comm_create_local_group): run one rank per
        host thread.  For tests on a one-GPU box; RCCL needs one device per rank."""
        L = _lib.lib()
        arr = (C.c_void_p * world)()
        _lib.check(L.cymf_comm_create_local_group(arr, int(world), int(device), int(max_floats)))
        out = []
        for r in range(world):
            c = cls.__new__(cls)
            c.L, c.rank, c.world, c.device, c.h = L, r, world, device, C.c_void_p(arr[r])
            _lib.track(c, last=True)
            out.append(c)
        return out

    @staticmethod
    def unique_id():
        buf = C.create_string_buffer(_lib.UNIQUE_ID_BYTES)
        _lib.check(_lib.lib().cymf_comm_unique_id(buf))
        return buf.raw

    @classmethod
    def from_env(cls, device=None, timeout=300.0):
        rank, world, local = env_rank_world()
        device = local if device is None else device
        port = os.environ.get("MASTER_PORT", "0")
        run = os.environ.get("TORCHELASTIC_RUN_ID", "none")
        # the launcher (torchrun agent) is the common parent of all ranks: its pid makes the name
        # unique per launch, so a file left behind by a crashed earlier run is never picked up
        path = f"/tmp/cymf_amd_rdzv_{port}_{run}_{world}_{os.getppid()}"
        if rank == 0:
            uid = cls.unique_id()
            tmp = path + f".{os.getpid()}"
            with open(tmp, "wb") as f:
                f.write(uid)
            os.replace(tmp, path)
        else:
            t0 = time.time()
            while True:
                try:
                    if os.path.exists(path):
                        with open(path, "rb") as f:
                            uid = f.read()
                        if len(uid) == _lib.UNIQUE_ID_BYTES:
                            break
                except OSError:
                    pass
                if time.time() - t0 > timeout:
                    raise TimeoutError("cymf_amd.dist: rendezvous file never appeared")
                time.sleep(0.05)
        c = cls(rank, world, device, uid)
        c._rdzv_path = path
        return c

    def allreduce(self, arr, op="sum"):
        a = np.ascontiguousarray(arr, dtype=np.float32).copy()
        _lib.check(self.L.cymf_comm_allreduce_f32(self.h, _lib.ptr(a), a.size, 1 if op == "max" else 0))
        return a

    def barrier(self):
        self.allreduce(np.zeros(1, dtype=np.float32))

    def close(self):
        if getattr(self, "h", None) is not None and self.h:
            self.L.cymf_comm_destroy(self.h)
            self.h = None
            if self.rank == 0 and getattr(self, "_rdzv_path", None):
                try:
                    os.remove(self._rdzv_path)
                except OSError:
                    pass


_START = time.time()
