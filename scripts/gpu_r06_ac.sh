#!/bin/bash
# Round-6 GPU session AC: what ships (generic HLLC / HLL with ddiv) against the same with the plane held in registers by the LLF
# kernels only (generic_keepllf: the other solvers' 12-row kernels lose their scratch); certificates of what ships
mkdir -p gpurun_out
export TMPDIR=/tmp
{
for tag in default generic_keepllf; do
  lib=""; [ $tag != default ] && lib=ramses_amd/lib/ab/libramses_amd_$tag.so
  for cfg in "hllc 1" "hll 1" "acoustic 1" "hllc 2" "hll 2"; do echo -n "$tag: "; RAMSES_AMD_LIB=$lib timeout 300 python scripts/sweep_probe.py 512 $cfg 2>&1 | grep -v amdgpu.ids | tail -1; done
done
} | cut -c1-300 | tee gpurun_out/r06_ac_keepllf.txt
( timeout 1500 python -m pytest tests/test_baseline_sizes_gpu.py tests/test_fast_certificate_gpu.py tests/test_godunov_gpu.py tests/test_amr_tiles_gpu.py tests/test_uniform_options_resident_gpu.py -m gpu -q --timeout 900 -s 2>&1 | grep -E "HLLC|hllc|hll |passed|failed|Error" | cut -c1-300 | tail -30 ) | tee -a gpurun_out/r06_ac_keepllf.txt
