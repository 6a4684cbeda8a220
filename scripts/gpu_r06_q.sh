#!/bin/bash
# Round-6 GPU session Q: kernel statistics of the sweep of the shell level (19.4 M cells) in tiles
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -rf gpurun_out/prof_q
rocprofv3 --kernel-trace --stats -f csv -d gpurun_out/prof_q -o m -- python scripts/amr_tile_probe.py 9 partial 5 > gpurun_out/prof_q.log 2>&1
tail -3 gpurun_out/prof_q.log | cut -c1-300
python scripts/kstats.py gpurun_out/prof_q 14 | cut -c1-220 | tee gpurun_out/r06_q_partial_kernel_stats.txt
rm -rf gpurun_out/prof_q
