#!/usr/bin/env python3
"""Print the per-kernel table of a rocprofv3 --kernel-trace --stats output directory.  python tools/kstats.py <dir> [top]"""
import csv
import glob
import os
import re
import sys

d = sys.argv[1]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 25
f = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)
if not f:
    sys.exit(f"no *kernel_stats.csv under {d}")
rows = list(csv.DictReader(open(f[0])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:top]:
    name = re.sub(r"\(anonymous namespace\)::|cymf::|void ", "", r["Name"])
    name = re.sub(r"\(.*", "", name)
    print(f'{name[:60]:60s} calls {int(r["Calls"]):6d} total {float(r["TotalDurationNs"])/1e6:10.3f} ms avg {float(r["AverageNs"])/1e3:10.1f} us '
          f'min {float(r["MinNs"])/1e3:9.1f} max {float(r["MaxNs"])/1e3:9.1f}  {float(r["Percentage"]):5.1f}%')
