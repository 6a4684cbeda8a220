#!/bin/bash
# Round-6 GPU session U: the fast arithmetic of the MHD sweep -- certificate tests, timing at 128^3 / 256^3, kernel statistics
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 1200 python -m pytest tests/test_mhd_fast_certificate_gpu.py tests/test_mhd_gpu.py tests/test_amr_tiles_gpu.py -m gpu -q --timeout 900 -s ) > gpurun_out/r06_u_pytest.txt 2>&1
grep -E "rel-Linf|passed|failed|Error|error" gpurun_out/r06_u_pytest.txt | tail -30 | cut -c1-400
{
for m in 0 1; do for lev in 7 8; do echo "# RAMSES_AMD_MHD_FAST=$m"; RAMSES_AMD_MHD_FAST=$m timeout 300 python scripts/mhd_probe.py $lev 5 2>&1 | grep -v amdgpu.ids | tail -1; done; done
echo "# RAMSES_AMD_MHD_FAST=1 llf/llf"; RAMSES_AMD_MHD_FAST=1 timeout 300 python scripts/mhd_probe.py 8 5 llf llf 2>&1 | grep -v amdgpu.ids | tail -1
} | cut -c1-300 | tee gpurun_out/r06_u_mhd_fast.txt
rm -rf gpurun_out/prof_u
RAMSES_AMD_MHD_FAST=1 rocprofv3 --kernel-trace --stats -f csv -d gpurun_out/prof_u -o m -- python scripts/mhd_probe.py 8 > gpurun_out/prof_u.log 2>&1
python scripts/kstats.py gpurun_out/prof_u 12 | cut -c1-200 | tee gpurun_out/r06_u_mhd_fast_kernel_stats.txt
rm -rf gpurun_out/prof_u gpurun_out/prof_u.log
