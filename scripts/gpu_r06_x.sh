#!/bin/bash
# Round-6 GPU session X: cache policy of the sweep's stores (nt / sc1), z-chunk of the flagship launch on the max-ilp build,
# HBM traffic (FETCH_SIZE / WRITE_SIZE passes) of the default and the nt variant; parity tests on the new default build
mkdir -p gpurun_out
export TMPDIR=/tmp
{
AB_CFGS="default:12,64 default:12,256 default:12,512" python scripts/ab_sweep.py st_nt st_ntsc1 st_sc1 2>&1 | grep -v amdgpu.ids
} | cut -c1-300 | tee gpurun_out/r06_x_store_policy.txt
for tag in default st_nt; do
  lib=""; [ $tag != default ] && lib=$PWD/ramses_amd/lib/ab/libramses_amd_$tag.so
  RAMSES_AMD_LIB=$lib PASSES=traffic timeout 500 bash scripts/profile_gpu.sh r06x_$tag --vcycle-level 0 --amr-level 0 --stress-steps 0 --mhd-level 0 2>&1 | tail -4 | cut -c1-300 | tee -a gpurun_out/r06_x_store_policy.txt
done
( timeout 1200 python -m pytest tests/test_godunov_gpu.py tests/test_amr_tiles_gpu.py tests/test_fast_certificate_gpu.py -m gpu -q --timeout 900 -x 2>&1 | tail -3 ) | tee -a gpurun_out/r06_x_store_policy.txt
rm -rf gpurun_out/prof_r06x_*/pmc_* gpurun_out/prof_r06x_*/trace
