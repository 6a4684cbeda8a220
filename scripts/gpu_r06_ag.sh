#!/bin/bash
# Round-6 GPU session AG: the C5-shaped leg (0.89 ms per coarse step in the first closing run, 1.04 in the second and third) with the
# tree-walking units of the first closing run's sources (old_amr) against the shipped library, twice each
mkdir -p gpurun_out
export TMPDIR=/tmp
{
for rep in 0 1; do for tag in default old_amr; do
  lib=""; [ $tag != default ] && lib=ramses_amd/lib/ab/libramses_amd_$tag.so
  RAMSES_AMD_LIB=$lib timeout 300 python - <<'PY' 2>/dev/null | tail -1
import sys, os
sys.path.insert(0, ".")
import torch, bench
torch.cuda.init()
o = bench.amr_c5_shape_bench()
print("%s: production %.4f ms (host %.4f)  all on tiles %.4f ms" % (os.environ.get("RAMSES_AMD_LIB", "")[-14:] or "default", o["production"]["ms_per_coarse_step"], o["production"]["host_ms_per_coarse_step"], o["all_levels_on_tiles"]["ms_per_coarse_step"]))
PY
done; done
} | tee gpurun_out/r06_ag_c5.txt
