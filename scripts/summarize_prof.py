#!/usr/bin/env python
"""Condense a scripts/profile_gpu.sh output directory into a short text summary."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def rows(pattern):
    for f in glob.glob(os.path.join(out, pattern), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                yield r


print("# rocprofv3 summary of", os.path.basename(out))
for line in open(os.path.join(out, "log.txt")):
    if line.startswith("{"):
        print("bench line:", line.strip()[:1500])
        break
print("\n## kernel stats (rocprofv3 --kernel-trace --stats)")
for r in rows("trace/**/*kernel_stats.csv"):
    print("%-90s calls=%s total_ns=%s avg_ns=%s pct=%s" % (r.get("Name", "")[:90], r.get("Calls"), r.get("TotalDurationNs"),
                                                         r.get("AverageNs"), r.get("Percentage")))
print("\n## per-dispatch of the sweep kernel (kernel trace)")
n = 0
for r in rows("trace/**/*kernel_trace.csv"):
    if "godunov" in r.get("Kernel_Name", ""):
        n += 1
        if n <= 3:
            dur = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            print("dur_ns=%d grid=%s wg=%s lds=%s vgpr=%s accum_vgpr=%s sgpr=%s scratch=%s" % (
                dur, r.get("Grid_Size"), r.get("Workgroup_Size"), r.get("LDS_Block_Size"), r.get("VGPR_Count"),
                r.get("Accum_VGPR_Count"), r.get("SGPR_Count"), r.get("Scratch_Size")))
print("sweep dispatches traced:", n)
print("\n## PMC counters, per dispatch of godunov_sweep_kernel (mean over dispatches)")
for d in sorted(glob.glob(os.path.join(out, "pmc_*"))):
    acc = defaultdict(list)
    for r in rows(os.path.basename(d) + "/**/*counter_collection.csv"):
        if "godunov" in r.get("Kernel_Name", ""):
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print("%-10s %-28s mean=%.6g  n=%d" % (os.path.basename(d), k, sum(v) / len(v), len(v)))
