#!/bin/bash
# Round-3 GPU session J: balanced work decomposition of the fused smoother (RAMSES_AMD_MG_BALANCE: the same number of planes for
# every one of (CUs) workgroups instead of (tile, z chunk) blocks in 2.8 rounds) -- same bits? faster?
# (RAMSES_AMD_MG_BALANCE was an A/B switch of that session only: the balanced decomposition did not pay and was removed again,
#  profiles/r03_vcycle_ab.txt)
mkdir -p gpurun_out
export TMPDIR=/tmp
for b in 1; do
( time RAMSES_AMD_MG_BALANCE=$b timeout 500 python -m pytest tests/test_multigrid_gpu.py tests/test_multigrid_parallel_gpu.py -m gpu -q -x --timeout 300 ) > gpurun_out/pytest_j_b$b.txt 2>&1
tail -4 gpurun_out/pytest_j_b$b.txt | cut -c1-200
done
vc() { timeout 200 python bench.py --no-cpu-baseline --steps 3 --warmup 1 --amr-level 0 --stress-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); v=d.get('vcycle'); print(v['ms_per_vcycle'], v['final_error'])"; }
{
for b in 0 1 2 3 4; do echo "== RAMSES_AMD_MG_BALANCE=$b"; RAMSES_AMD_MG_BALANCE=$b vc; done
echo "== RAMSES_AMD_MG_BALANCE=1 FUSED_MIN=128"; RAMSES_AMD_MG_BALANCE=1 RAMSES_AMD_MG_FUSED_MIN=128 vc
echo "== RAMSES_AMD_MG_BALANCE=2 FUSED_MIN=128"; RAMSES_AMD_MG_BALANCE=2 RAMSES_AMD_MG_FUSED_MIN=128 vc
echo "== RAMSES_AMD_MG_BALANCE=0 again"; RAMSES_AMD_MG_BALANCE=0 vc
} 2>&1 | tee gpurun_out/vcycle_balance.txt
cd /tmp
for b in 0 1; do
  RAMSES_AMD_MG_BALANCE=$b rocprofv3 --kernel-trace --stats -d /tmp/prof_b$b -o vc -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 3 --warmup 1 --amr-level 0 --stress-steps 0 > /dev/null 2>&1
  f=$(find /tmp/prof_b$b -name "*kernel_stats.csv" | head -1)
  echo "== kernel stats, RAMSES_AMD_MG_BALANCE=$b" >> $GRAFT_REPO_ROOT/gpurun_out/vcycle_balance.txt
  grep -E "mg_|Name" "$f" | cut -c1-260 | head -14 >> $GRAFT_REPO_ROOT/gpurun_out/vcycle_balance.txt
done
tail -32 $GRAFT_REPO_ROOT/gpurun_out/vcycle_balance.txt | cut -c1-230
