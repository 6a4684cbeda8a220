#!/bin/bash
# Round-3 GPU session K: XCD-aware block order of the fused smoother (RAMSES_AMD_MG_XCD=0/1) -- same bits? faster?
# (RAMSES_AMD_MG_XCD was an A/B switch of that session only: no gain, removed again, profiles/r03_vcycle_ab.txt)
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 500 python -m pytest tests/test_multigrid_gpu.py tests/test_multigrid_parallel_gpu.py -m gpu -q -x --timeout 300 ) > gpurun_out/pytest_k.txt 2>&1
tail -4 gpurun_out/pytest_k.txt | cut -c1-200
vc() { timeout 200 python bench.py --no-cpu-baseline --steps 3 --warmup 1 --amr-level 0 --stress-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); v=d.get('vcycle'); print(v['ms_per_vcycle'], v['final_error'])"; }
{
for x in 0 1 0 1; do echo "== RAMSES_AMD_MG_XCD=$x"; RAMSES_AMD_MG_XCD=$x vc; done
for z in 64 256; do echo "== RAMSES_AMD_MG_XCD=1 ZCHUNK=$z"; RAMSES_AMD_MG_XCD=1 RAMSES_AMD_MG_ZCHUNK=$z vc; done
} 2>&1 | tee gpurun_out/vcycle_xcd.txt
R=$PWD
cd /tmp
for x in 0 1; do
  RAMSES_AMD_MG_XCD=$x rocprofv3 --kernel-trace --stats -f csv -d /tmp/prof_x$x -o vc -- python $R/bench.py --no-cpu-baseline --steps 3 --warmup 1 --amr-level 0 --stress-steps 0 > /dev/null 2>&1
  echo "== kernel stats, RAMSES_AMD_MG_XCD=$x" >> $R/gpurun_out/vcycle_xcd.txt
  python $R/scripts/kstats.py /tmp/prof_x$x 12 2>&1 | grep -E "mg_|Name|name" | cut -c1-230 >> $R/gpurun_out/vcycle_xcd.txt
done
tail -30 $R/gpurun_out/vcycle_xcd.txt | cut -c1-230
