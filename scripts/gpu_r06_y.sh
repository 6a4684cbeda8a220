#!/bin/bash
# Round-6 GPU session Y: traced states of the left neighbour through wavefront shifts in the MHD flux / EMF kernels (MHD_XSHARE)
# against the loads of round 5 (mhd_noxs), strict and fast; the MHD parity tests; the y-first tile order of the flagship sweep
mkdir -p gpurun_out
export TMPDIR=/tmp
{
for tag in default mhd_noxs; do
  lib=""; [ $tag != default ] && lib=ramses_amd/lib/ab/libramses_amd_$tag.so
  for m in 0 1; do for lev in 7 8; do echo "# $tag RAMSES_AMD_MHD_FAST=$m"; RAMSES_AMD_LIB=$lib RAMSES_AMD_MHD_FAST=$m timeout 300 python scripts/mhd_probe.py $lev 5 2>&1 | grep -v amdgpu.ids | tail -1; done; done
done
rm -rf gpurun_out/prof_y
rocprofv3 --kernel-trace --stats -f csv -d gpurun_out/prof_y -o m -- python scripts/mhd_probe.py 8 > gpurun_out/prof_y.log 2>&1
python scripts/kstats.py gpurun_out/prof_y 5 | cut -c1-200
rm -rf gpurun_out/prof_y gpurun_out/prof_y.log
python scripts/ab_sweep.py yfast 2>&1 | grep -v amdgpu.ids
} | cut -c1-300 | tee gpurun_out/r06_y_mhd_xshare.txt
( timeout 1200 python -m pytest tests/test_mhd_gpu.py tests/test_mhd_dropin_gpu.py tests/test_mhd_fast_certificate_gpu.py -m gpu -q --timeout 900 2>&1 | tail -3 ) | tee -a gpurun_out/r06_y_mhd_xshare.txt
