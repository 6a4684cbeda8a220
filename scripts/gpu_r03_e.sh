#!/bin/bash
# Round-3 GPU session E: whole -m gpu suite, default bench line, kernel trace of the bench, V-cycle A/B of the fused
# restriction, tree-walking sweep probe + counters of the shipped kernel.   gpurun --timeout 1700 -- 'bash scripts/gpu_r03_e.sh'
mkdir -p gpurun_out
R=$PWD
export TMPDIR=/tmp
nproc > gpurun_out/box.txt; free -g >> gpurun_out/box.txt
( time timeout 1100 python -m pytest tests -m gpu -q --timeout 600 --durations=25 ) > gpurun_out/pytest_gpu.txt 2>&1
tail -40 gpurun_out/pytest_gpu.txt | cut -c1-200
timeout 300 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
cut -c1-3000 gpurun_out/bench_default.json
for v in 1 0; do
  echo "== RAMSES_AMD_MG_FUSE_RESTRICT=$v"
  RAMSES_AMD_MG_FUSE_RESTRICT=$v timeout 200 python bench.py --no-cpu-baseline --steps 3 --warmup 1 --amr-level 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(json.dumps(d.get('vcycle'))[:600])"
done 2>&1 | tee gpurun_out/vcycle_ab.txt
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats -f csv -d $R/gpurun_out/prof_final -o t -- python $R/bench.py --no-cpu-baseline --steps 10 --warmup 3 > $R/gpurun_out/prof_final.log 2>&1
python $R/scripts/kstats.py $R/gpurun_out/prof_final 16 | cut -c1-220 | tee $R/gpurun_out/kstats_final.txt
cd $R
{
  timeout 200 python scripts/amr_probe.py 8 morton 2>&1 | tail -1
  timeout 200 python scripts/amr_probe.py 8 scrambled 2>&1 | tail -1
} > gpurun_out/amr_probe.txt 2>&1
cat gpurun_out/amr_probe.txt
bash scripts/pmc_kernel.sh amr_r03 'amr_group_kernel|amr_prep_kernel' -- python scripts/amr_probe.py 8 morton > gpurun_out/pmc_amr_r03.txt 2>&1
cut -c1-200 gpurun_out/pmc_amr_r03.txt
