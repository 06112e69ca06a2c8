#!/usr/bin/env python3
"""Developer tool: aggregate a rocprofv3 *_kernel_trace.csv by kernel name.  python tools/prof_agg.py <dir-or-csv>"""
import collections
import csv
import glob
import os
import re
import sys

path = sys.argv[1]
files = [path] if path.endswith(".csv") else glob.glob(os.path.join(path, "**", "*_kernel_trace.csv"), recursive=True)
agg = collections.defaultdict(list)
for f in files:
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        m = re.search(r"(\w+_kernel)(<[^>]*>)?", n)
        n = (m.group(1) + (m.group(2) or "")) if m else n[:50]
        agg[n].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print(f"{k:60s} calls {len(v):6d} total {sum(v)/1e3:10.3f} ms avg {sum(v)/len(v):10.1f} us  min {min(v):9.1f} max {max(v):9.1f}")
