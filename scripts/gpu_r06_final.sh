#!/bin/bash
# Round-6 closing GPU session: the HBM traffic of the sweep FIRST (FETCH_SIZE / WRITE_SIZE passes + calibration, written where
# bench.py reads roofline.traffic from, so the bench line of this session carries this session's counters), the default bench
# line (with the CPU baselines), the rocprofv3 kernel statistics of the same bench command, then the whole -m gpu suite.
mkdir -p gpurun_out
R=$PWD
export TMPDIR=/tmp
if [ "${TRAFFIC:-1}" = 1 ]; then
  PASSES=traffic timeout 500 bash scripts/profile_gpu.sh r06_traffic --vcycle-level 0 --amr-level 0 --stress-steps 0 --mhd-level 0 2>&1 | tail -8 | cut -c1-300
  cp gpurun_out/prof_r06_traffic/traffic.json gpurun_out/r06_sweep_traffic.json 2>/dev/null && cp gpurun_out/r06_sweep_traffic.json profiles/r06_sweep_traffic.json
  cp gpurun_out/prof_r06_traffic/summary.txt gpurun_out/r06_sweep_traffic_summary.txt 2>/dev/null
  rm -rf gpurun_out/prof_r06_traffic/pmc_* gpurun_out/prof_r06_traffic/trace
fi
timeout 900 python bench.py > gpurun_out/r06_final_bench_default.json 2> gpurun_out/r06_final_bench_default.err
cut -c1-600 gpurun_out/r06_final_bench_default.json
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $R/gpurun_out/prof_r06_final -o t -- python $R/bench.py --no-cpu-baseline --steps 10 --warmup 3 > $R/gpurun_out/prof_r06_final.log 2>&1
f=$(find $R/gpurun_out/prof_r06_final -name "*kernel_stats.csv" | head -1)
cp "$f" $R/gpurun_out/r06_final_bench_kernel_stats.csv
head -16 "$f" | cut -c1-200
rm -rf $R/gpurun_out/prof_r06_final
cd $R
[ "${TESTS:-1}" = 1 ] && { ( time timeout 1500 python -m pytest tests -m gpu -q --timeout 600 --durations=12 ) > gpurun_out/r06_final_pytest_gpu.txt 2>&1; tail -24 gpurun_out/r06_final_pytest_gpu.txt | cut -c1-300; }
# the other solvers of the fast build on the final library (scripts/sweep_probe.py)
{ for cfg in "hllc 1" "hllc 2" "hllc 8" "hll 1" "acoustic 1"; do timeout 300 python scripts/sweep_probe.py 512 $cfg 2>&1 | grep -v amdgpu.ids | tail -1; done; } | tee gpurun_out/r06_final_solvers.txt
