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
print("\n## PMC counters, per dispatch (mean over dispatches)")
vals = {}
for d in sorted(glob.glob(os.path.join(out, "pmc_*"))):
    acc = defaultdict(list)
    for r in rows(os.path.basename(d) + "/**/*counter_collection.csv"):
        kn = r.get("Kernel_Name", "")
        if "godunov" in kn or "courant_kernel" in kn:
            acc[("courant:" if "courant_kernel" in kn else "") + r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        vals[(os.path.basename(d), k)] = sum(v) / len(v)
        print("%-14s %-28s mean=%.6g  n=%d" % (os.path.basename(d), k, sum(v) / len(v), len(v)))

# HBM traffic of the sweep with the FETCH_SIZE calibration (MI355X_MICROARCH.md, HBM section):
# the counter under-reports coalesced streaming reads on gfx950; calibrate on courant_kernel,
# whose algorithmic read volume is exactly nvar*N*8 bytes with the same 8 B/lane row access
import json, re
n = 512
m = re.search(r"--n (\d+)", " ".join(sys.argv[2:]))
if m:
    n = int(m.group(1))
fetch = vals.get(("pmc_fetch", "FETCH_SIZE"))
write = vals.get(("pmc_write", "WRITE_SIZE"))
cal = vals.get(("pmc_cal_fetch", "courant:FETCH_SIZE"))
if fetch and write and cal:
    known = 5 * n ** 3 * 8
    k = known / (cal * 1024.0)
    traffic = k * fetch * 1024.0 + write * 1024.0
    print("\n## HBM traffic per sweep launch")
    print("courant_kernel FETCH_SIZE raw = %.4g KB for %d B algorithmic -> calibration factor %.3f" % (cal, known, k))
    print("sweep: FETCH raw %.4g KB x %.3f + WRITE %.4g KB = %.4g bytes per launch (algorithmic %d)" % (
        fetch, k, write, traffic, n ** 3 * 80))
    json.dump({"n": n, "traffic_bytes_per_launch": traffic, "fetch_raw_kb": fetch, "write_raw_kb": write,
               "fetch_calibration": k, "algorithmic_bytes": n ** 3 * 80},
              open(os.path.join(out, "traffic.json"), "w"))
