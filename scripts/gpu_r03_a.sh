#!/bin/bash
# Round-3 GPU session A: whole -m gpu suite with per-test durations, default bench line, kernel trace of the bench,
# tree-walking sweep probe.   gpurun --timeout 1500 -- 'bash scripts/gpu_r03_a.sh'
mkdir -p gpurun_out
R=$PWD
export TMPDIR=/tmp
nproc > gpurun_out/box.txt; free -g >> gpurun_out/box.txt
( time timeout 1000 python -m pytest tests -m gpu -q --timeout 600 --durations=30 ) > gpurun_out/pytest_gpu.txt 2>&1
tail -45 gpurun_out/pytest_gpu.txt | cut -c1-200
timeout 300 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
cut -c1-1800 gpurun_out/bench_default.json
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats -f csv -d $R/gpurun_out/prof_final -o t -- python $R/bench.py --no-cpu-baseline --steps 10 --warmup 3 > $R/gpurun_out/prof_final.log 2>&1
python $R/scripts/kstats.py $R/gpurun_out/prof_final 14 | cut -c1-220
cd $R
{
  timeout 200 python scripts/amr_probe.py 8 morton 2>&1 | tail -1
  timeout 200 python scripts/amr_probe.py 8 scrambled 2>&1 | tail -1
} > gpurun_out/amr_probe.txt 2>&1
cat gpurun_out/amr_probe.txt
