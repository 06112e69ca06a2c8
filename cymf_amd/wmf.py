"""cymf.WMF on MI355X (class surface of cymf/wmf.pyx:32-93; ALS half-sweeps in csrc/wmf.hip)."""
import ctypes as C

import numpy as np

from . import _host, _lib


class WMF(object):
    """Weighted Matrix Factorization (Hu, Koren, Volinsky), http://yifanhu.net/PUB/cf.pdf

    Attributes (cymf/wmf.pyx:37-42): num_components, weight_decay, weight, W, H.
    """

    def __init__(self, num_components=20, weight_decay=0.01, weight=10.0):
        self.num_components = int(num_components)
        self.weight_decay = float(weight_decay)
        self.weight = float(weight)
        self.W = None
        self.H = None

    def fit(self, X, num_epochs=5, num_threads=1, valid_evaluator=None, early_stopping=False, verbose=True,
            *, dtype="float32", device=0, comm=None):
        """cymf/wmf.pyx:59-93.  ALS is deterministic and thread-count independent in the reference,
        so num_threads is accepted and ignored; only the pattern of X is used (wmf.pyx:111).
        comm (a dist.Comm, one process per GPU): the rows of every half-sweep are split over the ranks and
        all-gathered; every rank passes the same X and ends with the same full W and H."""
        X = _host.coerce_csr(X)
        self.valid_evaluator = valid_evaluator
        self.valid_dcg = -np.inf
        self.count = 0
        self.early_stopping = early_stopping
        if early_stopping and self.valid_evaluator is None:
            raise ValueError()
        U, I = X.shape
        with _host.GLOBAL_RNG_LOCK:
            _host.init_factors(self, U, I, self.num_components)
        Xt = X.T.tocsr()                                          # wmf.pyx:112
        trainer = WmfTrainer(U, I, self.num_components, self.weight, self.weight_decay, dtype=dtype,
                             device=device if comm is None else comm.device, comm=comm)
        try:
            trainer.set_data(X.indptr, X.indices, Xt.indptr, Xt.indices)
            trainer.upload(self.W, self.H)
            stopper = _host.EarlyStopping(self)
            bar = _host.Progress(num_epochs, verbose, ncols=100)
            width = len(str(num_epochs))
            epoch = 0
            chunks = _host.EpochChunks(num_epochs, self.valid_evaluator is not None or comm is not None)
            for n in chunks:
                trainer.epochs(n)
                for _ in range(n):
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


class WmfTrainer:
    """Object wrapper of the cymf_wmf_* C ABI."""

    def __init__(self, U, I, K, weight=10.0, weight_decay=0.01, dtype="float32", device=0, comm=None):
        self.L = _lib.lib()
        self.U, self.I, self.K = int(U), int(I), int(K)
        self.h = C.c_void_p()
        _lib.check(self.L.cymf_wmf_create(C.byref(self.h), self.U, self.I, self.K, weight, weight_decay,
                                          _lib.DTYPE_IDS[dtype], device))
        _lib.track(self)
        self.comm = comm          # keeps the communicator alive as long as the trainer
        if comm is not None:
            _lib.check(self.L.cymf_wmf_attach_comm(self.h, comm.h))

    def set_data(self, indptr, indices, t_indptr, t_indices):
        a, b, c, d = _lib.i32c(indptr), _lib.i32c(indices), _lib.i32c(t_indptr), _lib.i32c(t_indices)
        if len(a) != self.U + 1 or len(c) != self.I + 1:
            raise ValueError("indptr sizes do not match (U, I)")
        _lib.check(self.L.cymf_wmf_set_data(self.h, _lib.ptr(a), _lib.ptr(b), _lib.ptr(c), _lib.ptr(d)))

    def upload(self, W, H):
        W, H = _lib.f64c(W), _lib.f64c(H)
        if W.shape != (self.U, self.K) or H.shape != (self.I, self.K):
            raise ValueError("W/H shape mismatch")
        _lib.check(self.L.cymf_wmf_upload(self.h, _lib.ptr(W), _lib.ptr(H)))

    def download(self, W, H):
        _lib.out_f64(W, H)
        _lib.check(self.L.cymf_wmf_download(self.h, _lib.ptr(W), _lib.ptr(H)))

    def row_range(self, side):
        """Rows [lo, hi) of side 0 (users) / 1 (items) this rank solves."""
        lo, hi = C.c_int32(0), C.c_int32(0)
        _lib.check(self.L.cymf_wmf_row_range(self.h, int(side), C.byref(lo), C.byref(hi)))
        return lo.value, hi.value

    def half_sweep(self, side):
        _lib.check(self.L.cymf_wmf_half_sweep(self.h, int(side)))

    def epochs(self, n=1):
        _lib.check(self.L.cymf_wmf_epochs(self.h, int(n)))

    def close(self):
        if getattr(self, "h", None) is not None and self.h:
            self.L.cymf_wmf_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
