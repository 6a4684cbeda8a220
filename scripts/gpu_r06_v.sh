#!/bin/bash
# Round-6 GPU session V: code-generation knobs (LLVM AMDGPU scheduler strategies) on the flagship sweep, the multigrid kernels and
# the MHD kernels: one object recompiled per variant (scripts/build_unit_variant.sh), everything else the regular build
mkdir -p gpurun_out
export TMPDIR=/tmp
{
python scripts/ab_sweep.py sw_ilp sw_memcl sw_bias0 sw_bias100 2>&1 | grep -v amdgpu.ids
for tag in default mg_ilp mg_memcl mg_bias100; do
  lib=""; [ $tag != default ] && lib=ramses_amd/lib/ab/libramses_amd_$tag.so
  for rep in 0 1; do
  RAMSES_AMD_LIB=$lib timeout 300 python bench.py --steps 5 --warmup 2 --amr-level 0 --mhd-level 0 --stress-steps 0 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); v = j['vcycle']
        print('$tag rep$rep vcycle ms %.3f frac %.4f err %r | sweep ms %.3f' % (v['ms_per_vcycle'], v['roofline']['frac'], v['final_error'], j['ms_per_step']))
"
  done
done
for tag in default mhd_ilp mhd_memcl; do
  lib=""; [ $tag != default ] && lib=ramses_amd/lib/ab/libramses_amd_$tag.so
  echo "# $tag"; RAMSES_AMD_LIB=$lib timeout 300 python scripts/mhd_probe.py 8 5 2>&1 | grep -v amdgpu.ids | tail -1
done
} | cut -c1-300 | tee gpurun_out/r06_v_codegen.txt
