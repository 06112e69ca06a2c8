#!/usr/bin/env python3
"""Summarise the secondary paths' rocprofv3 passes of tools/profile_round.sh into profiles/<tag>_secondary_paths.md:
per-kernel time (--kernel-trace), HBM bytes per launch ((2 * FETCH_SIZE + WRITE_SIZE) KiB, separate --pmc passes; FETCH doubled per
MI355X_MICROARCH.md 'HBM') and, for WMF, the MFMA pipe's busy share (SQ_VALU_MFMA_BUSY_CYCLES over GRBM_GUI_ACTIVE / 8 cycles x 1024 SIMDs).
   python tools/summarize_secondary.py gpurun_out/r02prof r02"""
import collections
import csv
import glob
import os
import re
import sys

root, tag = sys.argv[1], sys.argv[2]
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")


def kname(s):
    m = re.search(r"(\w+_kernel)(<[^>]*>)?", s)
    return (m.group(1) + (m.group(2) or "")) if m else re.sub(r"\(.*", "", s)[:40]


def trace(d):
    f = glob.glob(os.path.join(root, d, "**", "*_kernel_trace.csv"), recursive=True)
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        agg[kname(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    return agg


def counters(d, name):
    f = glob.glob(os.path.join(root, d, "**", "*_counter_collection.csv"), recursive=True)
    agg = collections.defaultdict(list)
    if f:
        for r in csv.DictReader(open(f[0])):
            if r["Counter_Name"] == name:
                agg[kname(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return agg


def section(title, prefix, cmd, mfma=False, note=""):
    out = [f"## {title} (`{cmd}`)", "", "| kernel | calls | total ms | avg us | min | max | FETCH_SIZE KiB (raw, mean) | WRITE_SIZE KiB | HBM bytes/launch |" +
           (" MFMA busy |" if mfma else ""), "|---|---|---|---|---|---|---|---|---|" + ("---|" if mfma else "")]
    t = trace(prefix + "_stats")
    fe, wr = counters(prefix + "_fetch", "FETCH_SIZE"), counters(prefix + "_write", "WRITE_SIZE")
    busy = counters(prefix + "_mfma", "SQ_VALU_MFMA_BUSY_CYCLES") if mfma else {}
    gui = counters(prefix + "_mfma", "GRBM_GUI_ACTIVE") if mfma else {}
    for k, v in sorted(t.items(), key=lambda kv: -sum(kv[1])):
        if sum(v) < 0.05 * 1e3:
            continue
        f_ = sum(fe[k]) / len(fe[k]) if fe.get(k) else float("nan")
        w_ = sum(wr[k]) / len(wr[k]) if wr.get(k) else float("nan")
        row = f"| {k} | {len(v)} | {sum(v)/1e3:.3f} | {sum(v)/len(v):.1f} | {min(v):.1f} | {max(v):.1f} | {f_:.1f} | {w_:.1f} | {(2*f_+w_)*1024:.4g} |"
        if mfma:
            b, g = sum(busy.get(k, [0])), sum(gui.get(k, [0]))
            row += f" {b / (g / 8 * 1024):.3f} |" if g else " |"
        out.append(row)
    if note:
        out += ["", note]
    return out + [""]


lines = [f"# rocprofv3, secondary paths (round {tag[1:]}; tools/profile_round.sh: --kernel-trace --stats, then --pmc FETCH_SIZE, --pmc WRITE_SIZE"
         " (and --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE for WMF), each its own pass)", ""]
lines += section("WMF C4, K=64 then K=128, 4 epochs each", "wmf", "python3 tools/bench_models.py wmf", mfma=True,
                 note="MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs): the share of SIMD cycles in which the matrix pipe "
                      "executes (v_mfma_f32_32x32x2_f32: 64 cycles each).  The long rows' segments (wmf_seg_kernel) and the fill of their scratch "
                      "run on a second stream beside the whole-row kernels of the item sweep: their durations overlap the row kernel's "
                      "(a 7 ms fill is a fill that waited for CUs) and the column does not add up to the epoch.")
lines += section("RelMF 20000 x 8000, K=64, tile schedule: SGD, AdaGrad, Adam, 4 epochs each", "relmf", "python3 tools/relmf_check.py speed")
for pre in ("wmf", "relmf"):
    log = os.path.join(root, pre + "_stats.log")
    if os.path.exists(log):
        lines += [f"`{pre}` run under the profiler printed:", "```"] + [l.rstrip() for l in open(log) if l.startswith(("WMF", "RelMF"))] + ["```", ""]
open(os.path.join(ROOT, "profiles", f"{tag}_secondary_paths.md"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
