#!/bin/bash
# Round-6 GPU session Z: the ILP scheduling of the fast sweep units on the other solvers -- the minmod unit (ships with it) against
# the same unit without (st1_noilp) for hllc / hll / acoustic, the moncen unit with it (st2_ilp) against the default without
mkdir -p gpurun_out
export TMPDIR=/tmp
{
for tag in default st1_noilp; do
  lib=""; [ $tag != default ] && lib=ramses_amd/lib/ab/libramses_amd_$tag.so
  for r in llf hllc hll acoustic; do echo -n "$tag: "; RAMSES_AMD_LIB=$lib timeout 300 python scripts/sweep_probe.py 512 $r 1 2>&1 | grep -v amdgpu.ids | tail -1; done
done
for tag in default st2_ilp; do
  lib=""; [ $tag != default ] && lib=ramses_amd/lib/ab/libramses_amd_$tag.so
  for r in llf hllc hll; do echo -n "$tag: "; RAMSES_AMD_LIB=$lib timeout 300 python scripts/sweep_probe.py 512 $r 2 2>&1 | grep -v amdgpu.ids | tail -1; done
done
} | cut -c1-300 | tee gpurun_out/r06_z_ilp_solvers.txt
