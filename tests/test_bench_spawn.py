"""`python bench.py --gpus N` the way the driver starts it (no torch.distributed.run around it): the process becomes a
launcher, starts N rank processes, passes rank 0's one JSON line through and fails when a rank fails.  Real processes,
the real bench.py and dist.Comm.from_env rendezvous; libcymf_hip is replaced by tests/fakelib.FakeBenchLib through a
sitecustomize module on PYTHONPATH (tests/bench_stub), because this box has no GPU.  VERDICT r2 item 1."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(tmp_path, *argv, **env_extra):
    env = dict(os.environ, CYMF_FAKE_LIB_SCRATCH=str(tmp_path),
               PYTHONPATH=os.pathsep.join([os.path.join(ROOT, "tests", "bench_stub"), ROOT, os.environ.get("PYTHONPATH", "")]))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], env=env, capture_output=True, text=True, timeout=600)


def test_bench_gpus_2_starts_its_own_ranks_and_prints_one_line(tmp_path):
    r = _run(tmp_path, "--gpus", "2", "--scale", "0.02", "--steps", "4", "--warmup", "1")
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 4 and out["warmup"] == 1 and out["scaling"] == "strong"
    assert out["metric"].startswith("BPR triplet-updates/sec") and out["value"] > 0
    assert "RCCL" in out["config"]["sharding"] and out["cpu_baseline"] is None and out["secondary"] is None
    # a sharded job meets about 6 times per epoch at least (here: the divisor of --steps 4 nearest to 6), whatever the batch size says
    assert out["config"]["steps_per_epoch"] == 4 and out["epochs_covered"] == 1
    # the launcher never asked for the native library; both ranks made and destroyed a communicator and a trainer
    assert not os.path.exists(tmp_path / "launcher_loaded_lib")
    assert "rank 0/2" in r.stderr and "[bench] rank 1" not in r.stderr.replace("rank 1/2", "")
    # the node's dataset in /dev/shm is gone
    assert not [d for d in os.listdir("/dev/shm") if d.startswith("cymf_bench_") and "_C3_0.02" in d]


def test_bench_launcher_fails_when_a_rank_fails(tmp_path):
    r = _run(tmp_path, "--gpus", "3", "--scale", "0.02", "--steps", "2", "--warmup", "1", CYMF_FAKE_FAIL_RANK="1")
    assert r.returncode != 0
    assert "rank 1 exited with" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.strip().startswith("{")]          # no result line from a failed run
    assert not [d for d in os.listdir("/dev/shm") if d.startswith("cymf_bench_") and "_C3_0.02" in d]


def test_bench_pretend_world_diagnostic_exits_cleanly(tmp_path):
    """the --pretend-world branch closed its trainer twice (UnboundLocalError) and ran the secondary suite."""
    r = _run(tmp_path, "--gpus", "1", "--scale", "0.02", "--steps", "4", "--warmup", "1", "--pretend-world", "4")
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and "diagnostic" in json.loads(lines[0])
    assert "secondary" not in r.stderr
