#!/bin/bash
# Round-6 GPU session AA: the fused fast HLLC flux (timing at 512^3 against session Z's 5.24 / 6.31 ms, certificates), and the
# strict flagship kernel under the ILP scheduling with the parked / held variants (lottery)
mkdir -p gpurun_out
export TMPDIR=/tmp
{
for st in 1 2; do for r in hllc llf; do timeout 300 python scripts/sweep_probe.py 512 $r $st 2>&1 | grep -v amdgpu.ids | tail -1; done; done
python scripts/ab_sweep.py str_both_ilp str_keep_ilp str_relax 2>&1 | grep -v amdgpu.ids
} | cut -c1-300 | tee gpurun_out/r06_aa_hllc_fast.txt
( timeout 1500 python -m pytest tests/test_baseline_sizes_gpu.py tests/test_fast_certificate_gpu.py tests/test_godunov_gpu.py tests/test_amr_tiles_gpu.py -m gpu -q --timeout 900 -s 2>&1 | grep -E "HLLC|hllc|passed|failed|Error" | cut -c1-300 | tail -30 ) | tee -a gpurun_out/r06_aa_hllc_fast.txt
