"""cymf.BPR on MI355X.  Same class surface as the reference (cymf/bpr.pyx:37-113): constructor,
fit(), .W / .H float64 arrays trained in place; the epoch loop (cymf/bpr.pyx:160-171) runs in
libcymf_hip.so (cymf_amd/csrc/bpr.hip) through ctypes."""
import ctypes as C

import numpy as np

from . import _host, _lib


class BPR(object):
    """
    Bayesian Personalized Ranking (BPR), https://arxiv.org/pdf/1205.2618.pdf

    Attributes (as cymf/bpr.pyx:42-48):
        num_components, learning_rate, optimizer ('adam' | 'adagrad' | 'sgd'), weight_decay,
        W (U,K) user factors, H (I,K) item factors (np.float64)
    """

    def __init__(self, num_components=20, learning_rate=0.001, optimizer="adam", weight_decay=0.01):
        self.num_components = int(num_components)
        self.learning_rate = float(learning_rate)
        self.optimizer = optimizer
        self.weight_decay = float(weight_decay)
        self.W = None
        self.H = None
        if self.optimizer not in ("sgd", "adagrad", "adam"):
            raise Exception(f"{self.optimizer} is invalid.")   # cymf/bpr.pyx:65-66

    def fit(self, X, num_epochs=10, num_threads=1, valid_evaluator=None, early_stopping=False, verbose=True,
            *, mode=None, dtype=None, device=0, steps_per_epoch=None, comm=None, shard=None):
        """Train in place.  Positional arguments as cymf/bpr.pyx:68.

        num_threads == 1 (the reference's deterministic setting) selects the exact sequential-order
        mode, any other value its HOGWILD counterpart (throughput mode); `mode=` overrides.
        Keyword-only extras: dtype ('float32' | 'float64' device arithmetic), device, and for
        throughput mode steps_per_epoch / comm / shard (see cymf_amd.dist).  steps_per_epoch = windows of the shuffled order
        inside which the triplets are bucketed by positive item; None = chosen from the data so that the lock-free mode follows
        the reference's order closely; the number used is left in `steps_per_epoch_`.
        """
        X = _host.coerce_csr(X)
        self.valid_evaluator = valid_evaluator
        self.valid_dcg = -np.inf
        self.count = 0
        self.early_stopping = early_stopping
        if early_stopping and self.valid_evaluator is None:
            raise ValueError()                                   # cymf/bpr.pyx:94-95
        U, I = X.shape
        with _host.GLOBAL_RNG_LOCK:
            _host.init_factors(self, U, I, self.num_components)
            users, positives = _host.reference_shuffle(*X.nonzero())  # cymf/bpr.pyx:104
        users = users.astype(np.int32)
        positives = positives.astype(np.int32)
        indptr, indices = _host.membership_pattern(X)
        mode = _host.pick_mode(mode, num_threads)
        dtype = _host.pick_dtype(dtype, mode)

        global_pos, n_global = None, len(users)
        if shard is not None:   # user-sharded: keep this rank's users, remember global positions
            from .dist import shard_pattern, shard_triplets
            users, positives, global_pos = shard_triplets(users, positives, shard)
            indptr, indices = shard_pattern(indptr, indices, shard)

        trainer = BprTrainer(U, I, self.num_components, self.optimizer, self.learning_rate, self.weight_decay,
                             dtype=dtype, mode=mode, device=device, steps_per_epoch=steps_per_epoch, comm=comm)
        try:
            if comm is not None and shard is not None:
                from .dist import user_shards
                all_shards = user_shards(X.indptr, comm.world)            # the same cut on every rank
                if tuple(all_shards[comm.rank]) == tuple(shard):
                    trainer.set_user_bounds([lo for lo, _ in all_shards] + [all_shards[-1][1]])
            trainer.set_data(users, positives, indptr, indices, global_pos, n_global)
            self.steps_per_epoch_ = trainer.steps_per_epoch()
            trainer.upload(self.W, self.H)
            stopper = _host.EarlyStopping(self)
            bar = _host.Progress(num_epochs, verbose)
            width = len(str(num_epochs))
            self.losses = []
            epoch = 0
            chunks = _host.EpochChunks(num_epochs, self.valid_evaluator is not None or comm is not None)
            for n in chunks:                                     # (one call per epoch only where each epoch is looked at)
                for loss in trainer.epochs(n):
                    self.losses.append(loss)
                    epoch += 1
                    desc = f"EPOCH={epoch:{width}} "
                    if self.valid_evaluator:
                        trainer.download(self.W, self.H)
                        valid_dcg = self.valid_evaluator.evaluate(self.W, self.H)["DCG@5"]
                        if stopper.update(valid_dcg):
                            chunks.stop()
                            break
                        desc += ", DCG@5=" + str(np.round(valid_dcg, 3))
                    bar.step(desc)
            bar.close()
            trainer.download(self.W, self.H)
            self.performed_, self.skipped_ = trainer.stats()
            stopper.finish()
        finally:
            trainer.close()


class BprTrainer:
    """Thin object wrapper of the cymf_bpr_* C ABI (include/cymf_amd.h)."""

    def __init__(self, U, I, K, optimizer="adam", lr=0.001, wd=0.01, neg_seed=1234, dtype="float32",
                 mode="exact", device=0, steps_per_epoch=1, comm=None):
        self.L = _lib.lib()
        self.U, self.I, self.K = int(U), int(I), int(K)
        self.h = C.c_void_p()
        _lib.check(self.L.cymf_bpr_create(C.byref(self.h), self.U, self.I, self.K, _lib.OPT_IDS[optimizer], lr, wd,
                                          neg_seed, _lib.DTYPE_IDS[dtype], _lib.MODE_IDS[mode], device))
        _lib.track(self)
        self.N = 0
        if mode == "throughput" and steps_per_epoch != 1:      # None / 0: chosen from the data at set_data
            _lib.check(self.L.cymf_bpr_set_steps_per_epoch(self.h, int(steps_per_epoch or 0)))
        if comm is not None:
            _lib.check(self.L.cymf_bpr_attach_comm(self.h, comm.h))

    def steps_per_epoch(self):
        n = C.c_int32(0)
        _lib.check(self.L.cymf_bpr_get_steps_per_epoch(self.h, C.byref(n)))
        return n.value

    def set_user_bounds(self, bounds):
        """User ranges of all ranks (world + 1 boundaries): download() then returns every rank's rows of W on every rank."""
        b = np.ascontiguousarray(bounds, dtype=np.int64)
        _lib.check(self.L.cymf_bpr_set_user_bounds(self.h, _lib.ptr(b)))

    def set_data(self, users, positives, indptr, indices, global_pos=None, n_global=None):
        users, positives = _lib.i32c(users), _lib.i32c(positives)
        indptr, indices = _lib.i32c(indptr), _lib.i32c(indices)
        if len(indptr) != self.U + 1:
            raise ValueError("indptr must have U+1 entries")
        gp = None if global_pos is None else np.ascontiguousarray(global_pos, dtype=np.int64)
        self.N = len(users)
        _lib.check(self.L.cymf_bpr_set_data(self.h, _lib.ptr(users), _lib.ptr(positives), self.N, _lib.ptr(indptr),
                                            _lib.ptr(indices), _lib.ptr(gp), int(n_global if n_global is not None else self.N)))

    def upload(self, W, H):
        W, H = _lib.f64c(W), _lib.f64c(H)
        if W.shape != (self.U, self.K) or H.shape != (self.I, self.K):
            raise ValueError("W/H shape mismatch")
        _lib.check(self.L.cymf_bpr_upload(self.h, _lib.ptr(W), _lib.ptr(H)))

    def download(self, W, H):
        """Writes into the given C-contiguous float64 arrays (in place, as the reference trains)."""
        _lib.out_f64(W, H)
        _lib.check(self.L.cymf_bpr_download(self.h, _lib.ptr(W), _lib.ptr(H)))

    def epochs(self, n=1):
        loss = np.zeros(n, dtype=np.float64)
        _lib.check(self.L.cymf_bpr_epochs(self.h, int(n), _lib.ptr(loss)))
        return loss

    def steps(self, n, want_loss=False):
        loss = C.c_double(0.0)
        _lib.check(self.L.cymf_bpr_steps(self.h, int(n), C.byref(loss) if want_loss else None))
        return loss.value

    def sync(self):
        _lib.check(self.L.cymf_bpr_sync(self.h))

    def stats(self):
        p, s = C.c_int64(0), C.c_int64(0)
        _lib.check(self.L.cymf_bpr_stats(self.h, C.byref(p), C.byref(s)))
        return p.value, s.value

    def set_profiling(self, on=True):
        _lib.check(self.L.cymf_bpr_set_profiling(self.h, int(bool(on))))

    def kernel_time(self):
        ms, n, units = C.c_double(0), C.c_int64(0), C.c_int64(0)
        _lib.check(self.L.cymf_bpr_kernel_time(self.h, C.byref(ms), C.byref(n), C.byref(units)))
        return ms.value, n.value, units.value

    def last_negatives(self):
        out = np.empty(self.N, dtype=np.int32)
        _lib.check(self.L.cymf_bpr_last_negatives(self.h, _lib.ptr(out), self.N))
        return out

    def close(self):
        if getattr(self, "h", None) is not None and self.h:
            self.L.cymf_bpr_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
