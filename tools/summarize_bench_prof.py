#!/usr/bin/env python3
"""Summarise tools/profile_bench.sh's rocprofv3 passes over `python3 bench.py --steps 20 --warmup 5 --cpu-sample 0`:

  python tools/summarize_bench_prof.py gpurun_out/r03prof r03 [--compact]

writes profiles/<tag>_bench_kernels.md (per-kernel time of the whole command, HBM bytes per launch) and the entries of
profiles/traffic.json that bench.py puts into every roofline's `traffic` field.
HBM bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) KiB, separate --pmc passes: FETCH_SIZE doubled as MI355X_MICROARCH.md 'HBM'
prescribes for gfx950 (128-B read requests tallied at 64 B; calibrated there on 16-B-per-lane streaming reads -- dword gathers
are uncalibrated, so the absolute figure of the gather kernels is an upper-side estimate), WRITE_SIZE as read (exact for
streaming stores and float atomics).
The command runs the workloads one after the other; a dispatch belongs to the workload whose first characteristic kernel was
seen last (C3 sgd -> C3 adam -> C2 sgd -> C2 adam -> C4 K=64 -> C4 K=128 -> C5 -> RelMF).  --compact replaces the per-dispatch
counter CSVs (tens of MB) by the per-kernel aggregates this script needs (kept as <pass>/agg.json)."""
import collections
import csv
import glob
import json
import os
import re
import sys

root, tag = sys.argv[1], sys.argv[2]
compact = "--compact" in sys.argv
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")

# (workload, regex on the kernel name that opens it)
PHASES = [("C3_sgd", r"bpr_step_kernel<\d+, \w+, 0,"), ("C3_adam", r"bpr_step_kernel<\d+, \w+, 2,"),
          ("C2_sgd", r"bpr_group_kernel<\d+, \w+, 0,"), ("C2_adam", r"bpr_group_kernel<\d+, \w+, 2,"),
          ("C4_k64", r"wmf_\w+<2[,>]"), ("C4_k128", r"wmf_\w+<4[,>]"), ("C5_glove", r"glove_\w+"), ("RelMF", r"relmf_\w+|tile_\w+")]
SETUP = re.compile(r"pair_table_build|cast_|fillBuffer|copyBuffer|layout_kernel|seed_kernel|widen_|unsort|stream_copy|seam_")


def kname(s):
    m = re.search(r"(\w+_kernel)(<[^>]*>)?", s)
    return (m.group(1) + (m.group(2) or "")) if m else re.sub(r"\(.*", "", s)[:48]


def load_pass(d, counter):
    """[(dispatch_id, kernel, value, duration_us)] in dispatch order, from the raw CSV or the compacted aggregate."""
    agg = os.path.join(root, d, "agg.json")
    if os.path.exists(agg):
        return json.load(open(agg))
    files = glob.glob(os.path.join(root, d, "**", "*_counter_collection.csv" if counter else "*_kernel_trace.csv"), recursive=True)
    rows = []
    for r in csv.DictReader(open(files[0])):
        if counter and r["Counter_Name"] != counter:
            continue
        dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        rows.append([int(r["Dispatch_Id"]), kname(r["Kernel_Name"]), float(r["Counter_Value"]) if counter else 0.0, dur])
    rows.sort()
    if compact:
        json.dump(rows, open(agg, "w"))
        for f in files:
            os.remove(f)
    return rows


def by_phase(rows):
    """{phase: {kernel: [values]}}, {phase: {kernel: [durations]}}"""
    cur, vals, durs = "setup", collections.defaultdict(lambda: collections.defaultdict(list)), collections.defaultdict(lambda: collections.defaultdict(list))
    order = [p for p, _ in PHASES]
    for _, k, v, d in rows:
        for p, rx in PHASES:
            if re.match(rx, k) and (cur == "setup" or order.index(p) > order.index(cur)):
                cur = p
        vals[cur][k].append(v)
        durs[cur][k].append(d)
    return vals, durs


st = load_pass("stats", None)
fe = load_pass("fetch", "FETCH_SIZE")
wr = load_pass("write", "WRITE_SIZE")
_, dur = by_phase(st)
fv, _ = by_phase(fe)
wv, _ = by_phase(wr)

lines = [f"# rocprofv3 over the driver's bench command, {tag}", "",
         "`python3 bench.py --steps 20 --warmup 5 --cpu-sample 0` (headline C3 + every secondary), three passes: `--kernel-trace --stats`, "
         "`--pmc FETCH_SIZE`, `--pmc WRITE_SIZE`.  HBM bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) KiB (MI355X_MICROARCH.md 'HBM').", ""]
traffic = {}
for p, _ in PHASES:
    if p not in dur:
        continue
    lines += [f"## {p}", "", "| kernel | launches | total ms | avg us | min | max | FETCH_SIZE KiB (raw mean) | WRITE_SIZE KiB | HBM bytes / launch |", "|---|---|---|---|---|---|---|---|---|"]
    tot_bytes = 0.0
    for k, v in sorted(dur[p].items(), key=lambda kv: -sum(kv[1])):
        f_ = fv[p].get(k)
        w_ = wv[p].get(k)
        fm = sum(f_) / len(f_) if f_ else float("nan")
        wm = sum(w_) / len(w_) if w_ else float("nan")
        b = (2 * fm + wm) * 1024
        if not SETUP.search(k) and b == b:
            tot_bytes += b * len(v)
        if sum(v) >= 20.0:
            lines.append(f"| {k} | {len(v)} | {sum(v)/1e3:.3f} | {sum(v)/len(v):.1f} | {min(v):.1f} | {max(v):.1f} | {fm:.1f} | {wm:.1f} | {b:.4g} |")
    traffic[p] = {"bytes_all_kernels": tot_bytes, "kernels": {k: {"launches": len(v), "avg_us": sum(v) / len(v),
                                                                  "bytes_per_launch": (2 * (sum(fv[p][k]) / len(fv[p][k])) + sum(wv[p][k]) / len(wv[p][k])) * 1024
                                                                  if fv[p].get(k) and wv[p].get(k) else None,
                                                                  "write_bytes_per_launch": sum(wv[p][k]) / len(wv[p][k]) * 1024 if wv[p].get(k) else None}
                                                              for k, v in dur[p].items() if sum(v) >= 20.0}}
    lines += ["", f"all kernels of the phase except set-up copies / casts: {tot_bytes/1e9:.3f} GB", ""]
# the headline's dominant kernel over the TIMED launches only (the stats row includes the warm-up launches and the first-launch
# outlier), next to what bench.py measured with HIP events in the same profiled run (<root>/stats.json)
try:
    timed = [d_ for _, k, _, d_ in st if re.match(r"bpr_step_kernel<\d+, \w+, 0,", k)]
    line = json.loads(open(os.path.join(root, "stats.json")).read().strip().splitlines()[-1])
    n_t = int(line["roofline"]["launches"])
    if len(timed) >= n_t:
        mean_t = sum(timed[-n_t:]) / n_t
        # the C3 Adam leg launches the OPT=2 instantiation, so the last n_t OPT=0 launches are the timed window
        lines += ["## headline kernel, timed window", "",
                  f"`bpr_step_kernel` (SGD), the {n_t} timed launches of the kernel-trace pass: **{mean_t / 1e3:.3f} ms** per launch (rocprofv3); the same run's "
                  f"bench line, HIP events on the kernel's stream: **{line['roofline']['avg_launch_ms']:.3f} ms** "
                  f"(`roofline.achieved` {line['roofline']['achieved']:.0f} GB/s, frac {line['roofline']['frac']:.3f}; value {line['value'] / 1e9:.3f} G/s under the profiler).", ""]
except Exception as e:   # pragma: no cover
    lines += [f"(headline agreement not computed: {e})", ""]

# what bench.py reads: bytes per unit of each roofline's launch definition
#   (phase, dominant kernel, launches of it per epoch, what `traffic` means in bench.py's line)
UNITS = [("C3_sgd", r"bpr_step_kernel", None, "per launch of bpr_step_kernel"), ("C3_adam", r"bpr_step_kernel", None, "per launch of bpr_step_kernel"),
         ("C2_sgd", r"bpr_group_kernel", 1, "per epoch (one launch of bpr_group_kernel)"), ("C2_adam", r"bpr_group_kernel", 1, "per epoch (one launch of bpr_group_kernel)"),
         ("C4_k64", r"wmf_row_\w+", 2, "per epoch, all kernels of both half-sweeps"), ("C4_k128", r"wmf_row_\w+", 2, "per epoch, all kernels of both half-sweeps"),
         ("C5_glove", r"glove_step_kernel", 1, "per epoch (one launch of glove_step_kernel)"),
         ("RelMF", r"relmf_tile_kernel", 250, "per epoch: 250 tile launches + index stream + bucketing kernels")]
flat = {}
for p, rx, per_epoch, what in UNITS:
    if p not in traffic:
        continue
    dom = [(k, v) for k, v in traffic[p]["kernels"].items() if re.match(rx, k)]
    if not dom:
        continue
    k, v = max(dom, key=lambda kv: kv[1]["launches"] * kv[1]["avg_us"])
    if per_epoch is None:
        flat[p] = {"bytes": v["bytes_per_launch"], "unit": what, "kernel": k, "avg_us": v["avg_us"]}
    elif per_epoch == 1:
        flat[p] = {"bytes": v["bytes_per_launch"], "unit": what, "kernel": k, "avg_us": v["avg_us"], "write_bytes": v["write_bytes_per_launch"]}
    else:
        epochs = v["launches"] / per_epoch
        flat[p] = {"bytes": traffic[p]["bytes_all_kernels"] / epochs, "unit": what, "kernel": k, "epochs_in_profile": epochs}
lines += ["## what bench.py's `traffic` fields hold (profiles/traffic.json: bench)", "", "| workload | HBM bytes | unit | dominant kernel |", "|---|---|---|---|"]
for p, e in flat.items():
    lines.append(f"| {p} | {e['bytes']:.4g} | {e['unit']} | {e['kernel']} |")
open(os.path.join(ROOT, "profiles", f"{tag}_bench_kernels.md"), "w").write("\n".join(lines) + "\n")
tj_path = os.path.join(ROOT, "profiles", "traffic.json")
tj = json.load(open(tj_path)) if os.path.exists(tj_path) else {}
tj = {k: v for k, v in tj.items() if not k.endswith("_phases")}     # one set of per-kernel aggregates: the newest
tj[f"{tag}_phases"] = traffic
c3_slots = 5000000   # slots per launch of the step kernel in the profiled run: its bench line says (a step may be several windows)
try:
    cfg = json.load(open(os.path.join(root, "bench_plain.json")))["config"]
    c3_slots = int(cfg["triplets_per_gpu_per_step"]) // int(cfg.get("windows_per_step", 1))
except Exception:
    pass
tj["bench"] = dict(flat, source=f"profiles/{tag}_bench_kernels.md", c3_slots_per_launch=c3_slots)
for stale in ("glove_step_C5_K100", "relmf_step_20000x8000_K64", "_source_secondary"):   # round-1 figures of kernels that no longer exist
    tj.pop(stale, None)
tj[f"{tag}_source"] = f"profiles/{tag}_bench_kernels.md (tools/profile_bench.sh: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes over the driver's bench command)"
json.dump(tj, open(tj_path, "w"), indent=1)
print("\n".join(lines[:60]))
