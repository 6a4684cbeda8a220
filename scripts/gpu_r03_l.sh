#!/bin/bash
# Round-3 GPU session L (round end): whole -m gpu suite, smoke(), default bench line, kernel trace of the bench.
#   gpurun --timeout 1500 -- 'bash scripts/gpu_r03_l.sh'
mkdir -p gpurun_out
R=$PWD
export TMPDIR=/tmp
nproc > gpurun_out/box.txt; free -g >> gpurun_out/box.txt
( time timeout 1100 python -m pytest tests -m gpu -q --timeout 600 --durations=15 ) > gpurun_out/pytest_gpu.txt 2>&1
tail -26 gpurun_out/pytest_gpu.txt | cut -c1-200
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 300 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
cut -c1-3000 gpurun_out/bench_default.json
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats -f csv -d $R/gpurun_out/prof_final -o t -- python $R/bench.py --no-cpu-baseline --steps 10 --warmup 3 > $R/gpurun_out/prof_final.log 2>&1
python $R/scripts/kstats.py $R/gpurun_out/prof_final 16 | cut -c1-220 | tee $R/gpurun_out/kstats_final.txt
