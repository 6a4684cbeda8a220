#!/bin/bash
# Round-6 GPU session AH: the ILP scheduling on the fast moncen (st2) and slope-8 units, now that their HLLC kernels run the fused flux
mkdir -p gpurun_out
export TMPDIR=/tmp
{
for tag in default st2_ilp; do
  lib=""; [ $tag != default ] && lib=ramses_amd/lib/ab/libramses_amd_$tag.so
  for r in llf hllc hll; do echo -n "$tag: "; RAMSES_AMD_LIB=$lib timeout 300 python scripts/sweep_probe.py 512 $r 2 2>&1 | grep -v amdgpu.ids | tail -1; done
done
for tag in default st8_ilp; do
  lib=""; [ $tag != default ] && lib=ramses_amd/lib/ab/libramses_amd_$tag.so
  for r in llf hllc; do echo -n "$tag: "; RAMSES_AMD_LIB=$lib timeout 300 python scripts/sweep_probe.py 512 $r 8 2>&1 | grep -v amdgpu.ids | tail -1; done
done
} | cut -c1-300 | tee gpurun_out/r06_ah_ilp_st2_st8.txt
