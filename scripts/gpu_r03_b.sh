#!/bin/bash
# Round-3 GPU session B: the tests that failed in session A (fixed since), the fast-arithmetic certificate, counters of the
# SHIPPED tree-walking sweep.   gpurun --timeout 1200 -- 'bash scripts/gpu_r03_b.sh'
mkdir -p gpurun_out
R=$PWD
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_rho_fine_gpu.py tests/test_stated_sizes_gpu.py tests/test_fast_certificate_gpu.py -m gpu -q -s --timeout 600 --durations=10 ) > gpurun_out/pytest_b.txt 2>&1
tail -40 gpurun_out/pytest_b.txt | cut -c1-220
bash scripts/pmc_kernel.sh amr_shipped 'amr_group_kernel|amr_pack_kernel|amr_group_walk' -- python scripts/amr_probe.py 8 morton > gpurun_out/pmc_amr_shipped.txt 2>&1
cat gpurun_out/pmc_amr_shipped.txt | cut -c1-200
