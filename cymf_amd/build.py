"""Build libcymf_hip.so (gfx950 only) in-tree with hipcc.  `python -m cymf_amd.build`."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "build")
SO = os.path.join(HERE, "libcymf_hip.so")
SOURCES = ["core.hip", "rng.hip", "bpr.hip", "bpr_groups.hip", "sgd_models.hip", "relmf_tiles.hip", "wmf.hip", "comm.hip", "eval.hip", "expomf.hip"]
FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-pthread", "-Wall", "-Wno-unused-function"]


def _hipcc():
    for c in ("hipcc", "/opt/rocm/bin/hipcc"):
        try:
            subprocess.check_output([c, "--version"], stderr=subprocess.STDOUT)
            return c
        except Exception:
            continue
    raise RuntimeError("hipcc not found")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    hipcc = _hipcc()
    os.makedirs(OBJ, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(os.path.dirname(HERE), "include", "cymf_amd.h"))
    jobs = []
    sources = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    for s in sources:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJ, s.replace(".hip", ".o"))
        if force or _stale(obj, [src] + headers):
            jobs.append((src, obj))

    def cc(job):
        src, obj = job
        if verbose:
            print(f"[cymf_amd.build] hipcc {os.path.basename(src)}", flush=True)
        subprocess.check_call([hipcc] + FLAGS + os.environ.get("CYMF_EXTRA_HIPCC_FLAGS", "").split() + ["-c", src, "-o", obj])   # (developer: -D switches of experiments)

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(cc, jobs))
    objs = [os.path.join(OBJ, s.replace(".hip", ".o")) for s in sources]
    if force or jobs or _stale(SO, objs):
        if verbose:
            print("[cymf_amd.build] link libcymf_hip.so", flush=True)
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs +
                              ["-o", SO, "-pthread", "-L/opt/rocm/lib", "-lrccl", "-Wl,-rpath,/opt/rocm/lib"])
    return SO


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(SO)
