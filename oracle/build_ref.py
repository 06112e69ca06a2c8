#!/usr/bin/env python3
"""Build the REFERENCE's own hot-path modules (test infrastructure only) OUTSIDE this repository.

This compiles the reference's Cython sources *where they lie* under /root/reference into
`ref_dir()` = $CYMF_REF_BUILD_DIR or <tempdir>/cymf_amd_ref -- never under the repository root:
gpurun ships the whole tree (git-ignored files included) to the GPU box, and the reference must not
travel in any form, compiled or otherwise.  Nothing is copied into this repository.
It works only in the build container (/root/reference does not exist on the GPU box); there the tests
that compare against the compiled reference skip, and the committed fixtures (tests/golden/*.npz,
generated from it by tests/golden/make_golden.py) keep the bit-exact pin.
The product (cymf_amd/) never imports anything from oracle/.

What is built (unchanged sources, the reference's own toolchain = Cython + g++):
    cymf/math.pyx, optimizer.pyx, model.pyx, bpr.pyx, relmf.pyx, glove.pyx, metrics.pyx
with `legacy_implicit_noexcept=True`, i.e. the semantics of the Cython 0.29.15 the
reference pins (/root/reference/requirements.txt:1).

What is NOT built, and why (recorded in DESIGN.md):
    cymf/linalg.pyx, wmf.pyx, expomf.pyx  -- need "cblas.h" (/root/reference/cymf/linalg.pxd:22)
        and -lcblas, which this image lacks; a stand-in header would be required -> unbuildable.
    cymf/evaluator.pyx -- does not compile under Cython 3 (stray ')' at :89 and :137) and
        cannot be fixed without editing a copy of the source -> unbuildable.
    cymf/__init__.py is not used: it imports the dataset loaders which need `wget`.
        <ref_dir>/cymf/__init__.py is an empty file of ours, so `import cymf.bpr` works.

Usage:  python oracle/build_ref.py        (idempotent; skips if up to date)
"""
import os
import subprocess
import sys
import sysconfig
import tempfile

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def ref_dir() -> str:
    """Where the compiled reference lives: outside the repository, so that it cannot ship with it."""
    d = os.path.abspath(os.environ.get("CYMF_REF_BUILD_DIR") or os.path.join(tempfile.gettempdir(), "cymf_amd_ref"))
    if os.path.commonpath([d, ROOT]) == ROOT:
        raise RuntimeError(f"CYMF_REF_BUILD_DIR={d} is inside the repository; the compiled reference must stay outside")
    return d
MODULES = ["math", "optimizer", "model", "bpr", "relmf", "glove", "metrics"]


def available() -> bool:
    return os.path.isdir(os.path.join(REF, "cymf"))


def build(verbose: bool = True) -> bool:
    if not available():
        if verbose:
            print("[oracle ref] /root/reference absent: nothing to build (fixtures in tests/golden keep the pin)")
        return False
    import numpy

    OUT = ref_dir()
    pkg = os.path.join(OUT, "cymf")
    bld = os.path.join(OUT, "build")
    os.makedirs(pkg, exist_ok=True)
    os.makedirs(bld, exist_ok=True)
    init = os.path.join(pkg, "__init__.py")
    if not os.path.exists(init):
        with open(init, "w") as f:
            f.write("# empty on purpose (ours): the reference's __init__ needs `wget`\n")
    ext = sysconfig.get_config_var("EXT_SUFFIX")
    inc = [sysconfig.get_paths()["include"], numpy.get_include(), os.path.join(REF, "cymf")]
    for m in MODULES:
        src = os.path.join(REF, "cymf", m + ".pyx")
        cpp = os.path.join(bld, m + ".cpp")
        so = os.path.join(pkg, m + ext)
        if os.path.exists(so) and os.path.getmtime(so) >= os.path.getmtime(src):
            continue
        if verbose:
            print(f"[oracle ref] cython {src}")
        subprocess.check_call(
            [sys.executable, "-m", "cython", "--cplus", "-3", "-X", "legacy_implicit_noexcept=True",
             src, "-o", cpp],
            stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        cmd = ["g++", "-O3", "-fopenmp", "-std=c++11", "-fPIC", "-shared", "-w",
               "-DNPY_NO_DEPRECATED_API=NPY_1_7_API_VERSION"]
        for i in inc:
            cmd += ["-I", i]
        cmd += [cpp, "-o", so]
        if verbose:
            print(f"[oracle ref] g++ -> {so}")
        subprocess.check_call(cmd)
    return True


def import_ref():
    """Return the compiled reference modules (cymf.bpr etc.), or None where /root/reference is absent
    or the build failed.  Never loads anything from under the repository root."""
    if not available():
        return None
    OUT = ref_dir()
    pkg = os.path.join(OUT, "cymf")
    if not os.path.isdir(pkg):
        return None
    if OUT not in sys.path:
        sys.path.insert(0, OUT)
    try:
        import importlib
        mods = {m: importlib.import_module("cymf." + m) for m in ("bpr", "relmf", "glove", "math", "metrics")}
    except Exception as e:  # pragma: no cover
        print("[oracle ref] import failed:", e)
        return None
    return mods


if __name__ == "__main__":
    ok = build()
    if ok:
        mods = import_ref()
        print("[oracle ref] built:", sorted(mods) if mods else None)
