#!/bin/bash
# Round-6 GPU session AD: the ILP scheduling on capi_amr.o (ghost pre-pass, coarse update, plan kernels: shell level) and
# hydro_misc.o (courant_kernel: kernel statistics of the bench command)
mkdir -p gpurun_out
export TMPDIR=/tmp
{
for tag in default capi_amr_ilp; do
  lib=""; [ $tag != default ] && lib=ramses_amd/lib/ab/libramses_amd_$tag.so
  for cfg in "9 partial" "8 covered"; do
  RAMSES_AMD_LIB=$lib timeout 300 python - $cfg <<'PY' 2>/dev/null | tail -1
import sys, os
sys.path.insert(0, ".")
import torch, bench
torch.cuda.init()
o = bench.amr_resident_bench(int(sys.argv[1]), steps=7, kind=sys.argv[2])
print("%s %s %s: strict %.3f ms (%.4f)  fast %.3f ms (%.4f)" % (os.environ.get("RAMSES_AMD_LIB", "")[-20:] or "default", sys.argv[1], sys.argv[2], o["ms_per_sweep"], o["roofline"]["frac"], o["fast_arithmetic"]["ms_per_sweep"], o["fast_arithmetic"]["frac"]))
PY
  done
done
for tag in default misc_ilp; do
  lib=""; [ $tag != default ] && lib=$PWD/ramses_amd/lib/ab/libramses_amd_$tag.so
  rm -rf gpurun_out/prof_ad
  RAMSES_AMD_LIB=$lib rocprofv3 --kernel-trace --stats -f csv -d gpurun_out/prof_ad -o t -- python bench.py --no-cpu-baseline --steps 5 --warmup 2 --vcycle-level 0 --amr-level 0 --mhd-level 0 --stress-steps 0 > /dev/null 2>&1
  echo "# $tag"; python scripts/kstats.py gpurun_out/prof_ad 6 | grep -i "courant\|godunov" | cut -c1-200
done
rm -rf gpurun_out/prof_ad
} | cut -c1-300 | tee gpurun_out/r06_ad_ilp_misc.txt
