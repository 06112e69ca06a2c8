"""The C-ABI library loads and exports every symbol include/cymf_amd.h declares; without a GPU the
compute entry points fail loudly (no CPU fallback).  CPU only."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT
from cymf_amd import _lib


def _declared_functions():
    src = open(os.path.join(ROOT, "include", "cymf_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(cymf_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    names = _declared_functions()
    assert len(names) >= 40
    L = C.CDLL(_lib.SO_PATH)
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing


def test_binding_covers_every_declared_symbol():
    L = _lib.lib()
    assert sorted(L._signatures) == _declared_functions()


def test_no_cpu_fallback_without_device():
    if _lib.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(_lib.CymfError) as e:
        _lib.rng_fill_uniform(1234, 100, 4)
    assert e.value.code == -2 and "no CPU fallback" in str(e.value)
    h = C.c_void_p()
    rc = _lib.lib().cymf_bpr_create(C.byref(h), 4, 4, 8, 0, 0.1, 0.01, 1234, 0, 0, 0)
    assert rc == -2 and not h
    from cymf_amd import BPR, WMF
    from scipy import sparse
    X = sparse.csr_matrix(np.eye(4))
    with pytest.raises(_lib.CymfError):
        BPR(4).fit(X, num_epochs=1, verbose=False)
    with pytest.raises(_lib.CymfError):
        WMF(4).fit(X, num_epochs=1, verbose=False)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "cymf_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in text and "from oracle" not in text, f
                assert "libcymf_oracle" not in text and "orc_" not in text, f
