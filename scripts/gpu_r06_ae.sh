#!/bin/bash
# Round-6 GPU session AE: the fused HLLC flux, second form (side = sign of u*), in kernels that do not hold the plane in registers
# (hllc_f1k: 8 B of scratch like the generic routine, 11 % fewer instructions) against what ships
mkdir -p gpurun_out
export TMPDIR=/tmp
{
for tag in default hllc_f1k; do
  lib=""; [ $tag != default ] && lib=ramses_amd/lib/ab/libramses_amd_$tag.so
  for cfg in "hllc 1" "hllc 2" "hllc 8"; do echo -n "$tag: "; RAMSES_AMD_LIB=$lib timeout 300 python scripts/sweep_probe.py 512 $cfg 2>&1 | grep -v amdgpu.ids | tail -1; done
done
} | cut -c1-300 | tee gpurun_out/r06_ae_hllc_f1k.txt
( RAMSES_AMD_LIB=$PWD/ramses_amd/lib/ab/libramses_amd_hllc_f1k.so timeout 900 python -m pytest tests/test_baseline_sizes_gpu.py -k "hllc" -m gpu -q --timeout 900 -s 2>&1 | grep -E "HLLC|passed|failed|Error" | cut -c1-300 | tail -8 ) | tee -a gpurun_out/r06_ae_hllc_f1k.txt
