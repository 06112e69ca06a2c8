#!/usr/bin/env python3
"""Summarise rocprofv3 output directories (gpurun_out/prof*/...) into profiles/:
  python tools/summarize_prof.py <stats_dir> <fetch_dir> <write_dir> <tag>
writes profiles/<tag>_summary.md and updates profiles/traffic.json (HBM bytes per launch of the
dominant kernel: FETCH_SIZE doubled per MI355X_MICROARCH.md 'HBM' (gfx950 tallies 128-B read
requests at 64 B), WRITE_SIZE as read; both counters are in KiB)."""
import collections
import csv
import glob
import json
import os
import re
import sys

stats_dir, fetch_dir, write_dir, tag = sys.argv[1:5]
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")


def kname(s):
    m = re.search(r"(\w+_kernel|__amd\w+)", s)
    return m.group(1) if m else s[:40]


def counters(d, name):
    f = glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True)[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == name:
            agg[kname(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}, {k: len(v) for k, v in agg.items()}


lines = [f"# rocprofv3 summary {tag}", "", f"## --kernel-trace --stats ({os.environ.get('PROF_CMD', 'python3 bench.py --cpu-sample 0 --steps 50 --warmup 5')})", "",
         "| kernel | calls | total ms | avg us | % |", "|---|---|---|---|---|"]
f = glob.glob(os.path.join(stats_dir, "**", "*_kernel_stats.csv"), recursive=True)[0]
for r in csv.DictReader(open(f)):
    lines.append(f'| {kname(r["Name"])} | {r["Calls"]} | {float(r["TotalDurationNs"])/1e6:.3f} | {float(r["AverageNs"])/1e3:.2f} | {float(r["Percentage"]):.2f} |')
# the dominant kernel over the TIMED launches only (the stats row above includes the warm-up launches and the first-launch outlier),
# next to what bench.py measured with HIP events in the same profiled run (<stats_dir>.log, if it is there)
tr = glob.glob(os.path.join(stats_dir, "**", "*_kernel_trace.csv"), recursive=True)
if tr:
    rows = [r for r in csv.DictReader(open(tr[0])) if "bpr_step_kernel" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
    timed = dur[5:] if len(dur) > 5 else dur
    note = f"bpr_step_kernel over the {len(timed)} timed launches (warm-up of 5 excluded): rocprofv3 mean {sum(timed)/len(timed):.1f} us"
    log = stats_dir.rstrip("/") + ".log"
    if os.path.exists(log):
        m = re.findall(r'"avg_launch_ms": ([0-9.]+)', open(log).read())
        if m:
            note += f"; bench.py's HIP events in the same run: {float(m[-1])*1e3:.1f} us"
    lines += ["", note + "."]
fetch, nf = counters(fetch_dir, "FETCH_SIZE")
write, nw = counters(write_dir, "WRITE_SIZE")
lines += ["", "## --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, KiB per dispatch, mean)", "",
          "| kernel | dispatches | FETCH_SIZE KiB (raw) | WRITE_SIZE KiB | HBM bytes/launch = (2*FETCH + WRITE)*1024 |", "|---|---|---|---|---|"]
for k in sorted(set(fetch) | set(write), key=lambda k: -(fetch.get(k, 0) + write.get(k, 0))):
    tot = (2 * fetch.get(k, 0) + write.get(k, 0)) * 1024
    lines.append(f"| {k} | {nf.get(k, 0)} | {fetch.get(k, 0):.1f} | {write.get(k, 0):.1f} | {tot:.4g} |")
os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
open(os.path.join(ROOT, "profiles", f"{tag}_summary.md"), "w").write("\n".join(lines) + "\n")
tpath = os.path.join(ROOT, "profiles", "traffic.json")
traffic = json.load(open(tpath)) if os.path.exists(tpath) else {}
k = "bpr_step_kernel"
traffic["bpr_step_sgd_K128"] = (2 * fetch[k] + write[k]) * 1024
traffic["slots_per_launch"] = int(float(os.environ.get("PROF_SLOTS", "4.0 M").split()[0]) * 1e6)
traffic["_source"] = f"profiles/{tag}_summary.md (C3, {traffic['slots_per_launch']} slots per launch)"
json.dump(traffic, open(tpath, "w"), indent=1)
print("\n".join(lines))
