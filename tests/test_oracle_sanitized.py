"""SURVEY.md section 5 (race detection / sanitizers): the CPU restatement under AddressSanitizer + UBSan.

oracle/cymf_oracle.c is rebuilt with -fsanitize=address,undefined (oracle.build_sanitized) and the golden-vector tests of
tests/test_oracle_golden.py plus the HOGWILD leg that bench.py times are run against THAT library in a child interpreter
started with libasan preloaded (the interpreter itself is not instrumented).  Any heap overflow, use after free, signed
overflow, misaligned or out-of-range access in the oracle aborts the child; leak checking is off (CPython)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _libasan():
    p = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


@pytest.mark.skipif(_libasan() is None, reason="gcc has no libasan here")
def test_oracle_golden_vectors_under_asan_ubsan(tmp_path):
    import oracle
    so = oracle.build_sanitized()
    extra = tmp_path / "test_hogwild_leg.py"
    extra.write_text(
        "import numpy as np, oracle\n"
        "from cymf_amd import synthetic\n"
        "def test_hogwild_and_sequential_legs():\n"
        "    X = synthetic.implicit_matrix(200, 300, 6000, 7)\n"
        "    r, c = X.nonzero()\n"
        "    rs = np.random.RandomState(4321)\n"
        "    W, H = rs.uniform(-.1, .1, (200, 16)) / 16, rs.uniform(-.1, .1, (300, 16)) / 16\n"
        "    for opt in ('sgd', 'adagrad', 'adam'):\n"
        "        m = oracle.Bpr(W.copy(), H.copy(), opt, 0.05, 0.01)\n"
        "        m.epoch(r.astype(np.int32), c.astype(np.int32), X.indptr.astype(np.int32), X.indices.astype(np.int32))\n"
        "        m.epoch_hogwild(r.astype(np.int32), c.astype(np.int32), X.indptr.astype(np.int32), X.indices.astype(np.int32), 4)\n"
        "        assert np.isfinite(m.W).all() and np.isfinite(m.H).all()\n"
        "        m.close()\n")
    env = dict(os.environ, LD_PRELOAD=_libasan(), CYMF_ORACLE_SO=so, PYTHONPATH=os.pathsep.join([ROOT, os.path.join(ROOT, "tests")]),
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1",
               OMP_NUM_THREADS="4")
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-p", "no:cacheprovider",
                        os.path.join(ROOT, "tests", "test_oracle_golden.py"), str(extra)],
                       env=env, cwd=ROOT, capture_output=True, text=True, timeout=1200)
    tail = (r.stdout + r.stderr)[-4000:]
    assert r.returncode == 0, tail
    assert "AddressSanitizer" not in tail and "runtime error" not in tail, tail
    assert " passed" in r.stdout
