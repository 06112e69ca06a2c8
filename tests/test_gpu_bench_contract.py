"""bench.py's one-line JSON contract, exercised on a shrunken workload (--scale: the numbers are not a
result, the structure is): keys, units, roofline and cpu_baseline objects, exactly K timed steps."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line_contract():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "6", "--warmup", "2",
                          "--scale", "0.02", "--cpu-sample", "20000"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["metric"] == "BPR triplet-updates/sec at K=128" and d["unit"] == "triplet-updates/s"
    assert d["n_gpus"] == 1 and d["steps"] == 6 and d["warmup"] == 2 and d["higher_is_better"] is True
    assert d["scaling"] == "strong" and d["vs_baseline"] is None and d["dtype"] == "f32" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and r["launches"] == 6
    assert r["frac"] == pytest.approx(r["achieved"] / r["peak"]) and r["achieved"] > 0 and r["bytes_per_unit"] == 24 * 128 + 12
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "triplet-updates/s" and c["sample"]
    assert d["value"] > 0 and d["ms_per_step"] > 0
    # whole epochs inside the timed window, the side-stream work with them
    assert d["epochs_covered"] * d["config"]["steps_per_epoch"] == d["steps"] and "scaling_note" in d
    assert r["frac_min_traffic"] < r["frac"] and "traffic_source" in r and "basis" in r
    # BASELINE.json's other configs, driver-visible
    sec = d["secondary"]
    assert set(sec) == {"C3_bpr_adam_k128", "C2_bpr_k64", "C2_bpr_adam_k64", "C4_wmf_k64", "C5_glove_k100", "relmf_20000x8000_k64"}
    assert "adam" in sec["C3_bpr_adam_k128"]["workload"] and "adam" in sec["C2_bpr_adam_k64"]["workload"]     # the reference's default optimizer (cymf/bpr.pyx:50)
    assert sec["C2_bpr_k64"]["steps_per_epoch"] >= 16
    # every roofline names where its HBM traffic figure comes from (profiles/traffic.json, static); at --scale != 1 the
    # full-size figures do not apply and the field is null for the size-dependent ones
    for name, e in sec.items():
        assert "traffic" in e["roofline"], name
    assert "wmf_row_reg_kernel" in sec["C4_wmf_k64"]["roofline"]["kernel"] and "wmf_row_blk_kernel" in sec["C4_wmf_k64"]["k128"]["roofline"]["kernel"]
    for name, e in sec.items():
        assert "error" not in e, (name, e)
        assert e["value"] > 0 and e["ms"] > 0 and e["unit"] and e["workload"]
        assert e["roofline"]["bound"] in ("hbm", "mfma") and 0 < e["roofline"]["frac"] < 1.5 and e["roofline"]["peak"] > 0
    assert sec["C4_wmf_k64"]["roofline"]["bound"] == "mfma" and sec["C4_wmf_k64"]["k128"]["ms"] > 0
