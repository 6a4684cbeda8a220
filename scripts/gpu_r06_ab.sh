#!/bin/bash
# Round-6 GPU session AB: the fused fast HLLC flux after the scratch fix (default) against the generic routine with its divisions
# through ddiv (hllc_generic), 512^3 and the tile sweep of the 256^3 level; certificates on the default
mkdir -p gpurun_out
export TMPDIR=/tmp
{
for tag in default hllc_generic; do
  lib=""; [ $tag != default ] && lib=ramses_amd/lib/ab/libramses_amd_$tag.so
  for st in 1 2; do for r in hllc hll; do echo -n "$tag: "; RAMSES_AMD_LIB=$lib timeout 300 python scripts/sweep_probe.py 512 $r $st 2>&1 | grep -v amdgpu.ids | tail -1; done; done
done
} | cut -c1-300 | tee gpurun_out/r06_ab_hllc_fast.txt
( timeout 1500 python -m pytest tests/test_baseline_sizes_gpu.py tests/test_fast_certificate_gpu.py tests/test_godunov_gpu.py tests/test_amr_tiles_gpu.py -m gpu -q --timeout 900 -s 2>&1 | grep -E "HLLC|hllc|passed|failed|Error" | cut -c1-300 | tail -30 ) | tee -a gpurun_out/r06_ab_hllc_fast.txt
