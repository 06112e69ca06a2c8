"""cymf.ExpoMF on MI355X (class surface of cymf/expomf.pyx:39-103; EM epochs in csrc/expomf.hip)."""
import ctypes as C

import numpy as np
from scipy import sparse

from . import _host, _lib


class ExpoMF(object):
    """Exposure Matrix Factorization, https://arxiv.org/pdf/1510.07025.pdf

    Attributes (cymf/expomf.pyx:44-49): num_components, lam_y, weight_decay, W, H.
    """

    def __init__(self, num_components=20, lam_y=1.0, weight_decay=0.01):
        self.num_components = int(num_components)
        self.lam_y = float(lam_y)
        self.weight_decay = float(weight_decay)
        self.W = None
        self.H = None

    def fit(self, X, num_epochs=5, num_threads=1, valid_evaluator=None, early_stopping=False, verbose=True, *, device=0):
        """cymf/expomf.pyx:64-103.  The EM iteration is deterministic and thread-count independent in the
        reference, so num_threads is accepted and ignored; arithmetic is float64 on the device."""
        if X is None:
            raise ValueError()
        if sparse.isspmatrix(X):
            X = X.tocsr()
        elif isinstance(X, np.ndarray):
            X = sparse.csr_matrix(X)
        else:
            raise ValueError()
        X = X.astype(np.float64)
        self.valid_evaluator = valid_evaluator
        self.valid_dcg = -np.inf
        self.count = 0
        self.early_stopping = early_stopping
        if early_stopping and self.valid_evaluator is None:
            raise ValueError()
        if self.W is None:                                          # expomf.pyx:99-102: normal init, unlike BPR / WMF
            np.random.seed(4321)
            self.W = np.random.randn(X.shape[0], self.num_components) * 0.01
        if self.H is None:
            self.H = np.random.randn(X.shape[1], self.num_components) * 0.01
        self.W = np.ascontiguousarray(self.W, dtype=np.float64)
        self.H = np.ascontiguousarray(self.H, dtype=np.float64)
        # the reference tests A[X.nonzero()] (expomf.pyx:143): stored zeros do not count as exposure-one entries
        P = X.copy()
        P.eliminate_zeros()
        P.sort_indices()
        Pt = P.T.tocsr()
        Pt.sort_indices()
        U, I = X.shape
        L = _lib.lib()
        h = C.c_void_p()
        _lib.check(L.cymf_expomf_create(C.byref(h), U, I, self.num_components, self.lam_y, self.weight_decay, device))
        try:
            _lib.check(L.cymf_expomf_set_data(h, _lib.ptr(_lib.i32c(P.indptr)), _lib.ptr(_lib.i32c(P.indices)),
                                              _lib.ptr(_lib.i32c(Pt.indptr)), _lib.ptr(_lib.i32c(Pt.indices))))
            _lib.check(L.cymf_expomf_upload(h, _lib.ptr(self.W), _lib.ptr(self.H)))
            stopper = _host.EarlyStopping(self)
            bar = _host.Progress(num_epochs, verbose, ncols=100)
            width = len(str(num_epochs))
            for epoch in range(num_epochs):
                _lib.check(L.cymf_expomf_epochs(h, 1))
                desc = f"EPOCH={epoch+1:{width}} "
                if self.valid_evaluator:
                    _lib.check(L.cymf_expomf_download(h, _lib.ptr(self.W), _lib.ptr(self.H)))
                    valid_dcg = self.valid_evaluator.evaluate(self.W, self.H)["DCG@5"]
                    if stopper.update(valid_dcg):
                        break
                    desc += ", DCG@5=" + str(np.round(valid_dcg, 3))
                bar.step(desc)
            bar.close()
            _lib.check(L.cymf_expomf_download(h, _lib.ptr(self.W), _lib.ptr(self.H)))
            stopper.finish()
        finally:
            L.cymf_expomf_destroy(h)
