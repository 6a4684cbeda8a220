#!/usr/bin/env python
"""Print the top rows of a rocprofv3 --kernel-trace --stats csv directory."""
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:int(sys.argv[2]) if len(sys.argv) > 2 else 10]:
        print(r["Name"][:90], "calls=" + r["Calls"], "total_ns=" + r["TotalDurationNs"], "avg_ns=" + r["AverageNs"], "pct=" + r["Percentage"])
