#!/bin/bash
# Round-4 closing record: the default bench line (with the CPU baseline) and the rocprofv3 kernel statistics of the same command
mkdir -p gpurun_out
R=$PWD
export TMPDIR=/tmp
timeout 900 python bench.py > gpurun_out/r04_final_bench_default.json 2> gpurun_out/r04_final_bench_default.err
cut -c1-600 gpurun_out/r04_final_bench_default.json
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -f csv -d $R/gpurun_out/prof_r04_final2 -o t -- python $R/bench.py --no-cpu-baseline --steps 10 --warmup 3 > $R/gpurun_out/prof_r04_final2.log 2>&1
f=$(find $R/gpurun_out/prof_r04_final2 -name "*kernel_stats.csv" | head -1)
cp "$f" $R/gpurun_out/r04_final_bench_kernel_stats.csv
head -14 "$f" | cut -c1-200
