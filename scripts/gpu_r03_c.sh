#!/bin/bash
# Round-3 GPU session C: tree-walking sweep after the q-records / XCD order / padded stencil / out-of-line interpolation /
# fused prep launch: parity tests of the AMR sweep, probe (Morton, scrambled), per-kernel breakdown of both.
mkdir -p gpurun_out
R=$PWD
export TMPDIR=/tmp
( time timeout 600 python -m pytest tests/test_amr_godunov_gpu.py tests/test_baseline_sizes_gpu.py tests/test_mpi_amr_resident_gpu.py -m gpu -q -x --timeout 300 ) > gpurun_out/pytest_c.txt 2>&1
tail -8 gpurun_out/pytest_c.txt | cut -c1-220
for order in morton scrambled; do
  timeout 120 python scripts/amr_probe.py 8 $order 2>&1 | tail -1 | cut -c1-150
  cd /tmp
  timeout 200 rocprofv3 --kernel-trace --stats -f csv -d $R/gpurun_out/prof_amr_$order -o t -- python $R/scripts/amr_probe.py 8 $order > $R/gpurun_out/prof_amr_$order.log 2>&1
  cd $R
  python scripts/kstats.py gpurun_out/prof_amr_$order 12 | cut -c1-200
done 2>&1 | tee gpurun_out/amr_breakdown.txt
