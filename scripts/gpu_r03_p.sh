#!/bin/bash
# Round-3 GPU session P: overlap probe of the shipped kernel (XCD mapping box by box), HBM traffic of the sweep (FETCH_SIZE /
# WRITE_SIZE passes + calibration), the shim's share of C5's "flag" timer
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 200 python scripts/overlap_probe.py 512 2>/dev/null | grep '^{' > gpurun_out/overlap_probe_p.txt 2>&1
cut -c1-500 gpurun_out/overlap_probe_p.txt
PASSES=traffic timeout 400 bash scripts/profile_gpu.sh r03t --vcycle-level 0 --amr-level 0 --stress-steps 0 --spinup-ms 0 > gpurun_out/profile_r03t.log 2>&1
tail -6 gpurun_out/prof_r03t/summary.txt | cut -c1-300
timeout 200 python scripts/dropin_timing.py c5 7 9 8 prof > gpurun_out/c5_prof.txt 2>&1
cut -c1-1500 gpurun_out/c5_prof.txt
