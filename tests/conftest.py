import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # a fresh checkout has no libcymf_hip.so yet (built artefacts are git-ignored): build it once
    from cymf_amd import _lib, build as hip_build
    if not os.path.exists(_lib.SO_PATH):
        hip_build.build(verbose=False)


def golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def csr_from_golden(g, data=None):
    from scipy import sparse
    shape = tuple(int(x) for x in g["shape"]) if "shape" in g.files else (int(g["V"]), int(g["V"]))
    d = np.ones(len(g["indices"])) if data is None else data
    return sparse.csr_matrix((d, g["indices"], g["indptr"]), shape=shape)


def rel_fro(a, b):
    """The parity norm of SURVEY.md section 7 hard-3: ||a-b||_F / ||b||_F."""
    return float(np.linalg.norm(np.asarray(a, dtype=np.float64) - b) / max(np.linalg.norm(b), 1e-300))


def rel_maxabs(a, b):
    return float(np.abs(np.asarray(a, dtype=np.float64) - b).max() / max(np.abs(b).max(), 1e-300))


@pytest.fixture(scope="session")
def has_gpu():
    from cymf_amd import _lib
    return _lib.device_count() > 0
