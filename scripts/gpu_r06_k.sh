#!/bin/bash
# Round-6 GPU session K: force_fine under MPI with the acceleration left on the device (tests of every self-gravitating MPI
# configuration); A/B of the merged flux + EMF kernel of the MHD sweep
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests/test_mpi_amr_gravity_gpu.py tests/test_mpi_uniform_gravity_gpu.py tests/test_mpi_amr_resident_gpu.py -m gpu -q --timeout 900 -x ) > gpurun_out/r06_k_pytest.txt 2>&1
grep -v "^$" gpurun_out/r06_k_pytest.txt | tail -25 | cut -c1-300
{
for m in 0 1; do for lev in 7 8; do echo "# RAMSES_AMD_MHD_MERGED=$m"; RAMSES_AMD_MHD_MERGED=$m timeout 300 python scripts/mhd_probe.py $lev 5 2>&1 | grep -v amdgpu.ids | tail -1; done; done
} | cut -c1-300 | tee gpurun_out/r06_k_mhd_merged.txt
