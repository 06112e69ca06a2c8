"""ctypes binding of libcymf_hip.so (include/cymf_amd.h).  No CPU fallback: if the HIP
library is missing or no gfx950 device is visible the compute calls raise."""
import atexit
import ctypes as C
import os
import weakref

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.path.join(_HERE, "libcymf_hip.so")
_lib = None

OPT_IDS = {"sgd": 0, "adagrad": 1, "adam": 2}
DTYPE_IDS = {"float32": 0, "f32": 0, "float64": 1, "f64": 1}
MODE_IDS = {"exact": 0, "throughput": 1}
UNIQUE_ID_BYTES = 128


# Live native handles.  Python does not promise to run __del__ for objects that are still alive at interpreter
# shutdown, and when it does run them the order is arbitrary -- a trainer freed late called hipStreamDestroy / hipFree
# while the process was already inside exit() (round 1: SIGSEGV after rocprofv3's tool finalization).  So every wrapper
# registers itself here and an atexit hook closes what is still open BEFORE interpreter teardown, trainers and
# evaluators first, communicators (which they reference) last.
_live = [weakref.WeakSet(), weakref.WeakSet()]


def track(obj, last=False):
    _live[1 if last else 0].add(obj)
    return obj


def close_all():
    """Close every live trainer / evaluator / communicator handle (also the atexit hook)."""
    for group in _live:
        for o in list(group):
            try:
                o.close()
            except Exception:
                pass
    if _lib is not None:
        try:
            if _lib.cymf_device_count() > 0:
                _lib.cymf_device_sync(0)   # nothing in flight when the runtime's own exit handlers run
        except Exception:
            pass


atexit.register(close_all)


def out_f64(*arrays):
    """Download targets are written through raw pointers: they must be C-contiguous float64 ndarrays."""
    for a in arrays:
        if not isinstance(a, np.ndarray) or a.dtype != np.float64 or not a.flags.c_contiguous or not a.flags.writeable:
            raise ValueError("download targets must be writeable C-contiguous float64 ndarrays")


class CymfError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libcymf_hip error {code}: {msg}")
        self.code = code


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise ImportError(f"{SO_PATH} is missing: build it with `python -m cymf_amd.build` "
                          "(there is no CPU fallback for the MI355X kernels)")
    L = C.CDLL(SO_PATH)
    vp, i32, i64, u32, u64, f64, ci = C.c_void_p, C.c_int32, C.c_int64, C.c_uint32, C.c_uint64, C.c_double, C.c_int
    pp = C.POINTER(vp)
    sig = {
        "cymf_last_error": ([], C.c_char_p),
        "cymf_version": ([], ci),
        "cymf_device_count": ([], ci),
        "cymf_device_name": ([ci, C.c_char_p, ci], ci),
        "cymf_device_sync": ([ci], ci),
        "cymf_device_stream_copy_gbps": ([ci, i64, ci, vp], ci),
        "cymf_device_seam_probe": ([ci, ci, ci, vp, vp], ci),
        "cymf_rng_fill_uniform": ([ci, u32, u64, i64, i64, vp], ci),
        "cymf_bpr_create": ([pp, i32, i32, i32, ci, f64, f64, u32, ci, ci, ci], ci),
        "cymf_bpr_set_data": ([vp, vp, vp, i64, vp, vp, vp, i64], ci),
        "cymf_bpr_set_steps_per_epoch": ([vp, i32], ci),
        "cymf_bpr_get_steps_per_epoch": ([vp, vp], ci),
        "cymf_bpr_upload": ([vp, vp, vp], ci),
        "cymf_bpr_download": ([vp, vp, vp], ci),
        "cymf_bpr_epochs": ([vp, i32, vp], ci),
        "cymf_bpr_steps": ([vp, i32, vp], ci),
        "cymf_bpr_sync": ([vp], ci),
        "cymf_bpr_stats": ([vp, vp, vp], ci),
        "cymf_bpr_set_profiling": ([vp, ci], ci),
        "cymf_bpr_kernel_time": ([vp, vp, vp, vp], ci),
        "cymf_bpr_last_negatives": ([vp, vp, i64], ci),
        "cymf_bpr_destroy": ([vp], ci),
        "cymf_comm_unique_id": ([vp], ci),
        "cymf_comm_create": ([pp, vp, ci, ci, ci], ci),
        "cymf_comm_destroy": ([vp], ci),
        "cymf_comm_create_local_group": ([vp, ci, ci, i64], ci),
        "cymf_comm_allreduce_f32": ([vp, vp, i64, ci], ci),
        "cymf_bpr_attach_comm": ([vp, vp], ci),
        "cymf_bpr_set_user_bounds": ([vp, vp], ci),
        "cymf_relmf_create": ([pp, i32, i32, i32, ci, f64, f64, f64, u32, ci, ci, ci], ci),
        "cymf_relmf_set_data": ([vp, vp, vp], ci),
        "cymf_relmf_upload": ([vp, vp, vp], ci),
        "cymf_relmf_download": ([vp, vp, vp], ci),
        "cymf_relmf_epochs": ([vp, i32, vp], ci),
        "cymf_relmf_destroy": ([vp], ci),
        "cymf_relmf_set_steps_per_epoch": ([vp, i32], ci),
        "cymf_relmf_attach_comm": ([vp, vp, vp], ci),
        "cymf_glove_create": ([pp, i32, i32, i32, f64, f64, f64, ci, ci, ci], ci),
        "cymf_glove_set_data": ([vp, vp, vp, vp, i64], ci),
        "cymf_glove_upload": ([vp, vp, vp, vp, vp], ci),
        "cymf_glove_download": ([vp, vp, vp, vp, vp], ci),
        "cymf_glove_epochs": ([vp, i32, vp], ci),
        "cymf_glove_destroy": ([vp], ci),
        "cymf_glove_set_steps_per_epoch": ([vp, i32], ci),
        "cymf_glove_attach_comm": ([vp, vp, vp], ci),
        "cymf_wmf_create": ([pp, i32, i32, i32, f64, f64, ci, ci], ci),
        "cymf_wmf_set_data": ([vp, vp, vp, vp, vp], ci),
        "cymf_wmf_upload": ([vp, vp, vp], ci),
        "cymf_wmf_download": ([vp, vp, vp], ci),
        "cymf_wmf_half_sweep": ([vp, ci], ci),
        "cymf_wmf_epochs": ([vp, i32], ci),
        "cymf_wmf_destroy": ([vp], ci),
        "cymf_wmf_attach_comm": ([vp, vp], ci),
        "cymf_wmf_row_range": ([vp, ci, vp, vp], ci),
        "cymf_expomf_create": ([pp, i32, i32, i32, f64, f64, ci], ci),
        "cymf_expomf_set_data": ([vp, vp, vp, vp, vp], ci),
        "cymf_expomf_upload": ([vp, vp, vp], ci),
        "cymf_expomf_download": ([vp, vp, vp], ci),
        "cymf_expomf_epochs": ([vp, i32], ci),
        "cymf_expomf_destroy": ([vp], ci),
        "cymf_eval_create": ([pp, i32, i32, vp, vp, vp, vp, vp, i32, ci], ci),
        "cymf_eval_num_users": ([vp, vp], ci),
        "cymf_eval_negatives": ([vp, u32, i32, vp, vp, vp], ci),
        "cymf_eval_run": ([vp, vp, vp, i32, u32, i32, vp, i32, vp, ci, vp], ci),
        "cymf_eval_destroy": ([vp], ci),
    }
    for name, (args, res) in sig.items():
        fn = getattr(L, name)   # AttributeError here = the .so does not export the header's symbol
        fn.argtypes = args
        fn.restype = res
    L._signatures = sig
    _lib = L
    return L


def check(rc):
    if rc != 0:
        msg = lib().cymf_last_error()
        raise CymfError(rc, msg.decode("utf-8", "replace") if msg else "")


def ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def f64c(a):
    a = np.asarray(a)
    if a.dtype != np.float64 or not a.flags.c_contiguous:
        a = np.ascontiguousarray(a, dtype=np.float64)
    return a


def i32c(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def device_count():
    try:
        return int(lib().cymf_device_count())
    except (ImportError, OSError):
        return 0


def device_name(device=0):
    buf = C.create_string_buffer(256)
    check(lib().cymf_device_name(device, buf, 256))
    return buf.value.decode()


def device_sync(device=0):
    check(lib().cymf_device_sync(device))


def stream_copy_gbps(device=0, nbytes=1 << 30, iters=10):
    """Measured float4 stream-copy rate (read + write) in GB/s."""
    out = C.c_double(0.0)
    check(lib().cymf_device_stream_copy_gbps(device, int(nbytes), int(iters), C.byref(out)))
    return out.value


def seam_probe(device=0, memtype=0, rounds=64):
    """cymf_device_seam_probe: stale words seen at the four producer->consumer seams (all 0 on a healthy stack)."""
    stale = np.zeros(4, dtype=np.int64)
    words = C.c_int64(0)
    check(lib().cymf_device_seam_probe(device, int(memtype), int(rounds), ptr(stale), C.byref(words)))
    names = ("next_kernel_all_cus", "next_kernel_one_workgroup", "other_stream_behind_event", "copy_engine_to_host")
    return {"memtype": ("hipMalloc", "fine-grained", "uncached")[int(memtype)], "words_per_seam": int(words.value),
            **{n: int(v) for n, v in zip(names, stale)}}


def rng_fill_uniform(seed, rng_range, n, skip=0, device=0):
    """Draws [skip, skip+n) of UniformGenerator(0, rng_range, seed) (cymf/math.pyx:12-18), generated on the GPU."""
    out = np.empty(int(n), dtype=np.int64)
    check(lib().cymf_rng_fill_uniform(device, int(seed), int(rng_range), int(n), int(skip), ptr(out)))
    return out
