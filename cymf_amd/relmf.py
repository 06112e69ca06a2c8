"""cymf.RelMF on MI355X (class surface of cymf/relmf.pyx:37-101; loop in csrc/sgd_models.hip)."""
import ctypes as C

import numpy as np
from scipy import sparse

from . import _host, _lib


class RelMF(object):
    """Relevance Matrix Factorization (Rel-MF), https://arxiv.org/pdf/1909.03601.pdf"""

    def __init__(self, num_components=20, clip_value=0.1, learning_rate=0.001, optimizer="adam", weight_decay=0.01):
        self.num_components = int(num_components)
        self.clip_value = float(clip_value)
        self.learning_rate = float(learning_rate)
        self.optimizer = optimizer
        self.weight_decay = float(weight_decay)
        self.W = None
        self.H = None
        if self.optimizer not in ("sgd", "adagrad", "adam"):
            raise Exception(f"{self.optimizer} is invalid.")   # cymf/relmf.pyx:64-65

    def fit(self, X, num_epochs=10, num_threads=1, valid_evaluator=None, early_stopping=False, verbose=False,
            *, mode=None, dtype=None, device=0, comm=None, steps_per_epoch=None):
        """cymf/relmf.pyx:67-101.  comm (a dist.Comm, one process per GPU, throughput mode): users are cut into equal ranges
        over the ranks (every user has about I draws per epoch), the item table is replicated and synchronised after each
        of the steps_per_epoch sub-steps (default 4 per rank); every rank ends with the same full W and H."""
        if X is None:
            raise ValueError()
        if sparse.issparse(X):                                   # cymf/relmf.pyx:79-81: densify
            X = X.toarray()
        X = np.ascontiguousarray(np.asarray(X).astype(np.float64))
        self.valid_evaluator = valid_evaluator
        self.valid_dcg = -np.inf
        self.count = 0
        self.early_stopping = early_stopping
        propensities = np.maximum(X.mean(axis=0) / X.mean(axis=0).max(), 1e-5) ** 0.5   # cymf/relmf.pyx:88
        U, I = X.shape
        with _host.GLOBAL_RNG_LOCK:
            _host.init_factors(self, U, I, self.num_components)
        mode = _host.pick_mode(mode, num_threads)
        dtype = _host.pick_dtype(dtype, mode)
        bounds = None
        if comm is not None:
            bounds = np.round(np.linspace(0, U, comm.world + 1)).astype(np.int64)
            device = comm.device
            if steps_per_epoch is None:
                steps_per_epoch = 4 * comm.world
        trainer = RelMfTrainer(U, I, self.num_components, self.optimizer, self.learning_rate, self.weight_decay,
                               self.clip_value, dtype=dtype, mode=mode, device=device, comm=comm, user_bounds=bounds,
                               steps_per_epoch=steps_per_epoch or 1)
        try:
            trainer.set_data(X, propensities)
            trainer.upload(self.W, self.H)
            stopper = _host.EarlyStopping(self)
            bar = _host.Progress(num_epochs, verbose, ncols=100)
            width = len(str(num_epochs))
            self.losses = []
            epoch = 0
            chunks = _host.EpochChunks(num_epochs, self.valid_evaluator is not None or comm is not None)
            for n in chunks:
                losses = np.asarray(trainer.epochs(n), dtype=np.float64)
                if comm is not None:   # the loss of the whole job (each rank sums over its users' draws)
                    losses = comm.allreduce(losses.astype(np.float32)).astype(np.float64)
                for loss in losses:
                    self.losses.append(float(loss))
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
            stopper.finish()
        finally:
            trainer.close()


class RelMfTrainer:
    def __init__(self, U, I, K, optimizer="adam", lr=0.001, wd=0.01, clip=0.1, seed=1234, dtype="float32",
                 mode="exact", device=0, comm=None, user_bounds=None, steps_per_epoch=1):
        self.L = _lib.lib()
        self.U, self.I, self.K = int(U), int(I), int(K)
        self.h = C.c_void_p()
        _lib.check(self.L.cymf_relmf_create(C.byref(self.h), self.U, self.I, self.K, _lib.OPT_IDS[optimizer], lr, wd,
                                            clip, seed, _lib.DTYPE_IDS[dtype], _lib.MODE_IDS[mode], device))
        _lib.track(self)
        self.comm = comm
        if comm is not None:
            b = np.ascontiguousarray(user_bounds, dtype=np.int64)
            _lib.check(self.L.cymf_relmf_attach_comm(self.h, comm.h, _lib.ptr(b)))
            _lib.check(self.L.cymf_relmf_set_steps_per_epoch(self.h, int(steps_per_epoch)))

    def set_data(self, X, propensities):
        X, p = _lib.f64c(X), _lib.f64c(propensities)
        if X.shape != (self.U, self.I) or p.shape != (self.I,):
            raise ValueError("X / propensities shape mismatch")
        _lib.check(self.L.cymf_relmf_set_data(self.h, _lib.ptr(X), _lib.ptr(p)))

    def upload(self, W, H):
        W, H = _lib.f64c(W), _lib.f64c(H)
        if W.shape != (self.U, self.K) or H.shape != (self.I, self.K):
            raise ValueError("W/H shape mismatch")
        _lib.check(self.L.cymf_relmf_upload(self.h, _lib.ptr(W), _lib.ptr(H)))

    def download(self, W, H):
        _lib.out_f64(W, H)
        _lib.check(self.L.cymf_relmf_download(self.h, _lib.ptr(W), _lib.ptr(H)))

    def epochs(self, n=1):
        loss = np.zeros(n, dtype=np.float64)
        _lib.check(self.L.cymf_relmf_epochs(self.h, int(n), _lib.ptr(loss)))
        return loss

    def close(self):
        if getattr(self, "h", None) is not None and self.h:
            self.L.cymf_relmf_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
