#!/bin/bash
# Round-6 GPU session H: the fused ctoprim + edge fields + trace kernel of the MHD sweep: parity, A/B timing, kernel statistics
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 1200 python -m pytest tests/test_mhd_gpu.py tests/test_mhd_dropin_gpu.py -m gpu -q --timeout 900 ) > gpurun_out/r06_h_pytest.txt 2>&1
grep -v "^$" gpurun_out/r06_h_pytest.txt | tail -12 | cut -c1-300
{
for f in 1 0; do for lev in 7 8; do echo "# RAMSES_AMD_MHD_FUSED=$f"; RAMSES_AMD_MHD_FUSED=$f timeout 300 python scripts/mhd_probe.py $lev 2>&1 | grep -v amdgpu.ids | tail -1; done; done
echo "# llf / llf"; timeout 300 python scripts/mhd_probe.py 8 5 llf llf 2>&1 | grep -v amdgpu.ids | tail -1
} | cut -c1-300 | tee gpurun_out/r06_h_mhd_probe.txt
rm -rf gpurun_out/prof_h
rocprofv3 --kernel-trace --stats -f csv -d gpurun_out/prof_h -o m -- python scripts/mhd_probe.py 8 > gpurun_out/prof_h.log 2>&1
python scripts/kstats.py gpurun_out/prof_h 12 | cut -c1-200 | tee gpurun_out/r06_h_mhd_kernel_stats.txt
rm -rf gpurun_out/prof_h gpurun_out/prof_h.log
