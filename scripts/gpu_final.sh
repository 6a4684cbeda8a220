#!/bin/bash
# Round-end GPU session: the whole -m gpu suite, the default bench line, and the kernel trace of the same bench command.
#   gpurun --timeout 1200 -- 'bash scripts/gpu_final.sh'
mkdir -p gpurun_out
R=$PWD
export TMPDIR=/tmp
if [ -z "$SKIP_TESTS" ]; then
  ( time timeout 800 python -m pytest tests -m gpu -q --timeout 300 ) > gpurun_out/pytest_gpu.txt 2>&1
  tail -5 gpurun_out/pytest_gpu.txt | cut -c1-300
fi
timeout 400 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
cut -c1-1500 gpurun_out/bench_default.json
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $R/gpurun_out/prof_final -o t -- python $R/bench.py --no-cpu-baseline --steps 10 --warmup 3 > $R/gpurun_out/prof_final.log 2>&1
python $R/scripts/kstats.py $R/gpurun_out/prof_final 14 | cut -c1-220
