"""cymf.GloVe on MI355X (class surface of cymf/glove.pyx:46-177; loop in csrc/sgd_models.hip)."""
import ctypes as C
from collections import Counter

import numpy as np
from scipy import sparse

from . import _host, _lib


class GloVe(object):
    """GloVe: Global Vectors for Word Representation, https://nlp.stanford.edu/projects/glove/

    Attributes (cymf/glove.pyx:51-57): num_components, learning_rate (AdaGrad), alpha, x_max, W.
    """

    def __init__(self, num_components=50, learning_rate=0.01, alpha=0.75, x_max=10.0):
        self.num_components = int(num_components)
        self.learning_rate = float(learning_rate)
        self.alpha = float(alpha)
        self.x_max = float(x_max)
        self.W = None

    def fit(self, X, num_epochs, num_threads, verbose=False, *, mode=None, dtype=None, device=0, comm=None, steps_per_epoch=None, seed=None):
        """cymf/glove.pyx:75-112.  No seeding here either: initial factors come from the caller's
        global numpy state (glove.pyx:91-94).
        comm (a dist.Comm, one process per GPU, throughput mode): every rank passes the same X and the same numpy
        state; the pairs are sharded by central word, the context table is replicated and synchronised after each
        of the steps_per_epoch steps (default: 1 on one GPU, 4 per rank with a communicator -- few large steps learn
        visibly slower across ranks); every rank ends with the same full tables."""
        if X is None:
            raise ValueError()
        if not isinstance(X, (sparse.lil_matrix, sparse.csr_matrix, sparse.csc_matrix)):
            raise TypeError("X must be a type of scipy.sparse.*_matrix.")
        K = self.num_components
        with _host.GLOBAL_RNG_LOCK:
            if seed is not None:                                    # (extension: the reference leaves the seeding to the caller)
                np.random.seed(seed)
            self.W = np.random.uniform(low=-0.5, high=0.5, size=(X.shape[0], K)) / K
            self.bias = np.random.uniform(low=-0.5, high=0.5, size=(X.shape[0],)) / K
            _W = np.random.uniform(low=-0.5, high=0.5, size=(X.shape[1], K)) / K
            _bias = np.random.uniform(low=-0.5, high=0.5, size=(X.shape[0],)) / K
            central_words, context_words = X.nonzero()
            counts = X.data
            central_words, context_words, counts = _host.reference_shuffle(central_words, context_words, counts)
        mode = _host.pick_mode(mode, num_threads)
        dtype = _host.pick_dtype(dtype, mode)
        if steps_per_epoch is None:
            steps_per_epoch = 1 if comm is None else 4 * comm.world
        bounds = None
        n_all = len(counts)
        if comm is not None:
            from . import dist
            bounds = dist.word_bounds(central_words, X.shape[0], comm.world)
            mine = (central_words >= bounds[comm.rank]) & (central_words < bounds[comm.rank + 1])
            central_words, context_words, counts = central_words[mine], context_words[mine], counts[mine]
            device = comm.device
        trainer = GloveTrainer(X.shape[0], X.shape[1], K, self.learning_rate, self.x_max, self.alpha,
                               dtype=dtype, mode=mode, device=device, comm=comm, central_bounds=bounds,
                               steps_per_epoch=steps_per_epoch)
        try:
            trainer.set_data(central_words, context_words, counts)
            trainer.upload(self.W, self.bias, _W, _bias)
            bar = _host.Progress(num_epochs, verbose, ncols=100)
            width = len(str(num_epochs))
            n = max(n_all, 1)
            self.losses = []
            it = 0
            for m in _host.EpochChunks(num_epochs, comm is not None):
                losses = np.asarray(trainer.epochs(m), dtype=np.float64)
                if comm is not None:   # the loss of the whole job
                    losses = comm.allreduce(losses.astype(np.float32)).astype(np.float64)
                for loss in losses:
                    it += 1
                    self.losses.append(float(loss) / n)
                    bar.step(f"ITER={it:{width}}, LOSS: {np.round(float(loss) / n, 4):.4f}")   # glove.pyx:158-162
            bar.close()
            trainer.download(self.W, self.bias, _W, _bias)
        finally:
            trainer.close()
        self.W = (self.W + _W) / 2.0                               # glove.pyx:112

    def save_word2vec_format(self, path, index2word):
        """gensim KeyedVectors text format (cymf/glove.pyx:164-177)."""
        from pathlib import Path
        with Path(path).open("w") as f:
            f.write(f"{self.W.shape[0]} {self.W.shape[1]}\n")
            for i in range(self.W.shape[0]):
                f.write(f"{index2word[i]} " + " ".join(list(map(str, self.W[i]))) + "\n")


def read_text(fname, min_count=5, window_size=10):
    """Co-occurrence builder (cymf/glove.pyx:183-241): one-sided window, weight 1/distance,
    words rarer than min_count dropped, vocabulary ids in order of first appearance.
    Returns (csr_matrix (V,V) with X[cur, prev] accumulated, index->word dict).

    Multi-line files, as the reference treats them (:199-203): the counts are taken on
    `raw.replace("\n", "<eos>").split(" ")`, so the last word of a line and the first word of the next
    are ONE token "last<eos>first" and neither occurrence counts for its word, while the
    windows are built per line of `raw.split("\n")`.  `count` is a plain dict there: a word that only ever
    stands at a line boundary (or the empty token of a trailing newline) is missing from it and the
    lookup raises KeyError -- kept (a Counter would silently drop the word instead)."""
    with open(fname) as f:
        raw = f.read()
    count = dict(Counter(raw.replace("\n", "<eos>").split(" ")))
    w2i, i2w, lines_ids = {}, {}, []
    for line in raw.split("\n"):
        ids = []
        for w in line.split(" "):
            if count[w] >= min_count:
                if w not in w2i:
                    w2i[w] = len(w2i)
                    i2w[w2i[w]] = w
                ids.append(w2i[w])
        lines_ids.append(np.asarray(ids, dtype=np.int64))
    V = len(w2i)
    rows, cols, vals = [], [], []
    for ids in lines_ids:
        n = len(ids)
        for d in range(1, min(window_size, n - 1) + 1 if n > 1 else 1):
            rows.append(ids[d:])        # current word j
            cols.append(ids[:-d])       # previous word k = j - d
            vals.append(np.full(n - d, 1.0 / d))
    if rows:
        r, c, v = np.concatenate(rows), np.concatenate(cols), np.concatenate(vals)
    else:
        r = c = np.zeros(0, dtype=np.int64)
        v = np.zeros(0)
    M = sparse.coo_matrix((v, (r, c)), shape=(V, V)).tocsr()   # duplicates summed, as the hash map does
    return M, i2w


class GloveTrainer:
    def __init__(self, V, Vc, K, lr=0.01, x_max=10.0, alpha=0.75, dtype="float32", mode="exact", device=0,
                 comm=None, central_bounds=None, steps_per_epoch=1):
        self.L = _lib.lib()
        self.V, self.Vc, self.K = int(V), int(Vc), int(K)
        self.h = C.c_void_p()
        _lib.check(self.L.cymf_glove_create(C.byref(self.h), self.V, self.Vc, self.K, lr, x_max, alpha,
                                            _lib.DTYPE_IDS[dtype], _lib.MODE_IDS[mode], device))
        _lib.track(self)
        self.comm = comm
        if steps_per_epoch != 1:
            _lib.check(self.L.cymf_glove_set_steps_per_epoch(self.h, int(steps_per_epoch)))
        if comm is not None:
            b = np.ascontiguousarray(central_bounds, dtype=np.int64)
            if len(b) != comm.world + 1:
                raise ValueError("central_bounds must have world + 1 entries")
            _lib.check(self.L.cymf_glove_attach_comm(self.h, comm.h, _lib.ptr(b)))

    def set_data(self, central, context, counts):
        c, x, n = _lib.i32c(central), _lib.i32c(context), _lib.f64c(counts)
        if not (len(c) == len(x) == len(n)):
            raise ValueError("central/context/counts length mismatch")
        _lib.check(self.L.cymf_glove_set_data(self.h, _lib.ptr(c), _lib.ptr(x), _lib.ptr(n), len(c)))

    def upload(self, W, bias, Wc, bias_c):
        a, b, c, d = _lib.f64c(W), _lib.f64c(bias), _lib.f64c(Wc), _lib.f64c(bias_c)
        if a.shape != (self.V, self.K) or c.shape != (self.Vc, self.K) or b.shape != (self.V,) or d.shape != (self.V,):
            raise ValueError("parameter shape mismatch")
        _lib.check(self.L.cymf_glove_upload(self.h, _lib.ptr(a), _lib.ptr(b), _lib.ptr(c), _lib.ptr(d)))

    def download(self, W, bias, Wc, bias_c):
        _lib.out_f64(W, bias, Wc, bias_c)
        _lib.check(self.L.cymf_glove_download(self.h, _lib.ptr(W), _lib.ptr(bias), _lib.ptr(Wc), _lib.ptr(bias_c)))

    def epochs(self, n=1):
        loss = np.zeros(n, dtype=np.float64)
        _lib.check(self.L.cymf_glove_epochs(self.h, int(n), _lib.ptr(loss)))
        return loss

    def close(self):
        if getattr(self, "h", None) is not None and self.h:
            self.L.cymf_glove_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
