"""Stand-ins for libcymf_hip's entry points on a box without a GPU (test infrastructure, never shipped):

FakeCommLib   -- cymf_comm_* over files in a scratch directory (used by tests/test_dist_rendezvous.py)
FakeBenchLib  -- the above plus the few cymf_bpr_* / cymf_device_* calls bench.py's N-rank path makes, so that
                 `python bench.py --gpus N` can be run end to end by real processes (tests/test_bench_spawn.py).
                 Its "trainer" performs no arithmetic: it counts the triplets it was given.

The product code asks `_lib.lib()` for the library on every call; the tests replace that function."""
import ctypes as C
import os
import time

import numpy as np


class _FakeFn:
    def __init__(self, fn):
        self.fn = fn

    def __call__(self, *a):
        return self.fn(*a)


def _val(h):
    return h.value if hasattr(h, "value") else h


class FakeCommLib:
    """cymf_comm_unique_id / create / destroy / allreduce_f32 over files in a scratch directory: rank 0's id is a random
    token; create() fails if the id a rank presents is not the token rank 0 published; allreduce meets in the directory."""

    def __init__(self, scratch):
        self.scratch = scratch
        self.handles = {}
        self.calls = []
        self.cymf_comm_unique_id = _FakeFn(self._unique_id)
        self.cymf_comm_create = _FakeFn(self._create)
        self.cymf_comm_destroy = _FakeFn(self._destroy)
        self.cymf_comm_allreduce_f32 = _FakeFn(self._allreduce)
        self.cymf_last_error = _FakeFn(lambda: b"fake error")

    def _unique_id(self, buf):
        token = os.urandom(128)
        C.memmove(buf, token, 128)
        with open(os.path.join(self.scratch, "token"), "wb") as f:
            f.write(token)
        return 0

    def _create(self, out_ref, id_buf, rank, world, device):
        want = open(os.path.join(self.scratch, "token"), "rb").read()
        if bytes(id_buf.raw[:128]) != want:
            return -4
        h = 1000 + rank
        self.handles[h] = (rank, world, 0)
        out_ref._obj.value = h
        self.calls.append(("create", rank, world, device))
        return 0

    def _destroy(self, h):
        self.calls.append(("destroy", _val(h)))
        return 0

    def _allreduce(self, h, ptr, n, op):
        rank, world, gen = self.handles[_val(h)]
        self.handles[_val(h)] = (rank, world, gen + 1)
        a = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_float)), shape=(n,))
        np.save(os.path.join(self.scratch, f"ar{gen}_{rank}.npy.tmp"), a)
        os.replace(os.path.join(self.scratch, f"ar{gen}_{rank}.npy.tmp.npy"), os.path.join(self.scratch, f"ar{gen}_{rank}.npy"))
        t0 = time.time()
        parts = []
        for r in range(world):
            p = os.path.join(self.scratch, f"ar{gen}_{r}.npy")
            while not os.path.exists(p):
                if time.time() - t0 > 20:
                    return -4
                time.sleep(0.01)
            parts.append(np.load(p))
        a[:] = np.max(parts, axis=0) if op == 1 else np.sum(parts, axis=0)
        return 0


class FakeBenchLib(FakeCommLib):
    """What bench.py's rank path calls, and nothing else: create / set_data / upload / steps / sync / stats /
    set_profiling / kernel_time / destroy, the communicator attachment, and the device queries.  CYMF_FAKE_FAIL_RANK=r
    makes rank r's first cymf_bpr_steps call fail (the launcher must then end the other ranks)."""

    def __init__(self, scratch):
        super().__init__(scratch)
        self.trainers = {}
        for name in ("bpr_create", "bpr_set_data", "bpr_set_steps_per_epoch", "bpr_upload", "bpr_steps", "bpr_sync", "bpr_stats",
                     "bpr_set_profiling", "bpr_kernel_time", "bpr_destroy", "bpr_attach_comm", "device_count", "device_name",
                     "device_sync", "comm_create_local_group"):
            setattr(self, "cymf_" + name, _FakeFn(getattr(self, "_" + name)))

    def _bpr_create(self, out_ref, U, I, K, opt, lr, wd, seed, dtype, mode, device):
        h = 5000 + len(self.trainers)
        self.trainers[h] = {"spe": 1, "N": 0, "steps": 0, "comm": None, "device": device}
        out_ref._obj.value = h
        return 0

    def _bpr_set_steps_per_epoch(self, h, n):
        self.trainers[_val(h)]["spe"] = int(n)
        return 0

    def _bpr_attach_comm(self, h, comm):
        self.trainers[_val(h)]["comm"] = _val(comm)
        return 0

    def _bpr_set_data(self, h, users, positives, n, indptr, indices, gpos, n_global):
        self.trainers[_val(h)].update(N=int(n), n_global=int(n_global))
        return 0

    def _bpr_upload(self, h, W, H):
        return 0

    def _bpr_steps(self, h, n, loss):
        t = self.trainers[_val(h)]
        if os.environ.get("CYMF_FAKE_FAIL_RANK") is not None and os.environ.get("CYMF_FAKE_FAIL_RANK") == os.environ.get("RANK"):
            return -3
        t["steps"] += int(n)
        time.sleep(0.002 * int(n))
        return 0

    def _bpr_sync(self, h):
        return 0

    def _bpr_stats(self, h, p_ref, s_ref):
        t = self.trainers[_val(h)]
        p_ref._obj.value = t["steps"] * (t["N"] // t["spe"])          # every slot "performed", none skipped
        s_ref._obj.value = 0
        return 0

    def _bpr_set_profiling(self, h, on):
        self.trainers[_val(h)]["t_steps0"] = self.trainers[_val(h)]["steps"]
        return 0

    def _bpr_kernel_time(self, h, ms_ref, n_ref, units_ref):
        t = self.trainers[_val(h)]
        n = t["steps"] - t.get("t_steps0", 0)
        ms_ref._obj.value = 2.0 * n
        n_ref._obj.value = n
        units_ref._obj.value = n * (t["N"] // t["spe"])
        t["t_steps0"] = t["steps"]
        return 0

    def _bpr_destroy(self, h):
        self.calls.append(("bpr_destroy", _val(h)))
        return 0

    def _comm_create_local_group(self, arr, world, device, max_floats):
        for r in range(world):
            h = 2000 + r
            self.handles[h] = (r, world, 0)
            arr[r] = h
        return 0

    def _device_count(self):
        return 8

    def _device_name(self, device, buf, n):
        name = f"fake gfx950 #{device}".encode()
        C.memmove(buf, name + b"\0", len(name) + 1)
        return 0

    def _device_sync(self, device):
        return 0
