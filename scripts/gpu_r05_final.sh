#!/bin/bash
# Round-5 closing GPU session: the whole -m gpu suite, the default bench line (with the CPU baseline), the rocprofv3 kernel
# statistics of the same bench command, the HBM traffic of the sweep (FETCH_SIZE / WRITE_SIZE passes + calibration).
mkdir -p gpurun_out
R=$PWD
export TMPDIR=/tmp
( time timeout 1300 python -m pytest tests -m gpu -q --timeout 600 --durations=12 ) > gpurun_out/r05_final_pytest_gpu.txt 2>&1; [ "${ONLY_TESTS:-0}" = 1 ] && { tail -8 gpurun_out/r05_final_pytest_gpu.txt; exit 0; }
tail -24 gpurun_out/r05_final_pytest_gpu.txt | cut -c1-300
timeout 600 python bench.py > gpurun_out/r05_final_bench_default.json 2> gpurun_out/r05_final_bench_default.err
cut -c1-400 gpurun_out/r05_final_bench_default.json
cd /tmp
timeout 240 rocprofv3 --kernel-trace --stats -f csv -d $R/gpurun_out/prof_r05_final -o t -- python $R/bench.py --no-cpu-baseline --steps 10 --warmup 3 > $R/gpurun_out/prof_r05_final.log 2>&1
f=$(find $R/gpurun_out/prof_r05_final -name "*kernel_stats.csv" | head -1)
cp "$f" $R/gpurun_out/r05_final_bench_kernel_stats.csv
head -14 "$f" | cut -c1-200
cd $R
if [ "${TRAFFIC:-1}" = 1 ]; then
  PASSES=traffic timeout 500 bash scripts/profile_gpu.sh r05_traffic --vcycle-level 0 --amr-level 0 --stress-steps 0 --mhd-level 0 2>&1 | tail -8
  cp gpurun_out/prof_r05_traffic/traffic.json gpurun_out/r05_sweep_traffic.json 2>/dev/null
  cp gpurun_out/prof_r05_traffic/summary.txt gpurun_out/r05_sweep_traffic_summary.txt 2>/dev/null
fi
