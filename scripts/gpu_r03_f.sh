#!/bin/bash
# Round-3 GPU session F: multigrid parity tests after the fused prolongation, V-cycle A/B (prolongation / restriction fused or
# not), shell/interior split of the dense sweep re-measured, C4 under MPI with the multigrid levels resident or synced.
mkdir -p gpurun_out
R=$PWD
export TMPDIR=/tmp
( time timeout 600 python -m pytest tests/test_multigrid_gpu.py tests/test_multigrid_parallel_gpu.py tests/test_dropin_gpu.py tests/test_baseline_sizes_gpu.py tests/test_stated_sizes_gpu.py -m gpu -q -x --timeout 400 -k "not c3_" ) > gpurun_out/pytest_f.txt 2>&1
tail -6 gpurun_out/pytest_f.txt | cut -c1-200
for v in "1 1" "0 1" "1 0" "0 0"; do
  set -- $v
  echo "== RAMSES_AMD_MG_FUSE_PROLONG=$1 RAMSES_AMD_MG_FUSE_RESTRICT=$2"
  RAMSES_AMD_MG_FUSE_PROLONG=$1 RAMSES_AMD_MG_FUSE_RESTRICT=$2 timeout 200 python bench.py --no-cpu-baseline --steps 3 --warmup 1 --amr-level 0 --stress-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(json.dumps(d.get('vcycle'))[:330])"
done 2>&1 | tee gpurun_out/vcycle_ab2.txt
timeout 200 python scripts/overlap_probe.py 512 2>&1 | tail -4 | cut -c1-600 | tee gpurun_out/overlap_probe.txt
for np in 2 4 8; do
  timeout 300 python scripts/dropin_timing.py gravmpi 7 4 $np gpu 2>&1 | cut -c1-900
done | tee gpurun_out/dropin_gravmpi.txt
timeout 300 python scripts/dropin_timing.py gravmpi 7 4 8 ref 2>&1 | cut -c1-900 | tee -a gpurun_out/dropin_gravmpi.txt
