"""Put on PYTHONPATH by tests/test_bench_spawn.py only: every interpreter started with CYMF_FAKE_LIB_SCRATCH set (the
launcher bench.py becomes and the rank processes it starts) gets tests/fakelib.FakeBenchLib in place of libcymf_hip,
so that `python bench.py --gpus N` runs end to end on a box without a GPU.  bench.py itself has no test hook."""
import os
import sys

_scratch = os.environ.get("CYMF_FAKE_LIB_SCRATCH")
if _scratch:
    _tests = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.dirname(_tests))
    sys.path.insert(0, _tests)
    import fakelib
    from cymf_amd import _lib

    _fake = fakelib.FakeBenchLib(_scratch)
    _real_lib = _lib.lib

    def _lib_or_fake():
        if "RANK" not in os.environ:                 # the launcher: record that it asked for the library at all
            open(os.path.join(_scratch, "launcher_loaded_lib"), "w").close()
        return _fake

    _lib.lib = _lib_or_fake
