#!/bin/bash
# Round-6 GPU session W: the max-ilp scheduling strategy on the sweep units -- flagship kernel fast + strict, the tile sweep of AMR
# levels, the tree-walking sweep; bit-for-bit tests of the tile sweep with the variant library
mkdir -p gpurun_out
export TMPDIR=/tmp
{
python scripts/ab_sweep.py sw_ilp2 sw_ilp_a sw_ilp_b sw_ilp_c 2>&1 | grep -v amdgpu.ids
for tag in default sw_ilp2; do
  lib=""; [ $tag != default ] && lib=ramses_amd/lib/ab/libramses_amd_$tag.so
  for cfg in "8 full" "8 covered" "9 partial"; do
  RAMSES_AMD_LIB=$lib timeout 300 python - $cfg <<'PY' 2>/dev/null | tail -1
import sys, os
sys.path.insert(0, ".")
import torch, bench
torch.cuda.init()
o = bench.amr_resident_bench(int(sys.argv[1]), steps=7, kind=sys.argv[2])
print("%s %s %s: strict %.3f ms (%.4f)  fast %.3f ms (%.4f)" % (os.environ.get("RAMSES_AMD_LIB", "")[-16:] or "default", sys.argv[1], sys.argv[2], o["ms_per_sweep"], o["roofline"]["frac"], o["fast_arithmetic"]["ms_per_sweep"], o["fast_arithmetic"]["frac"]))
PY
  done
done
for tag in default amr_ilp; do
  lib=""; [ $tag != default ] && lib=ramses_amd/lib/ab/libramses_amd_$tag.so
  RAMSES_AMD_TILE_SWEEP=0 RAMSES_AMD_LIB=$lib timeout 300 python - 8 full <<'PY' 2>/dev/null | tail -1
import sys, os
sys.path.insert(0, ".")
import torch, bench
torch.cuda.init()
o = bench.amr_resident_bench(int(sys.argv[1]), steps=7, kind=sys.argv[2])
print("tree walk %s: %.3f ms (%.4f)" % (os.environ.get("RAMSES_AMD_LIB", "")[-16:] or "default", o["ms_per_sweep"], o["roofline"]["frac"]))
PY
done
} | cut -c1-300 | tee gpurun_out/r06_w_ilp.txt
( RAMSES_AMD_LIB=ramses_amd/lib/ab/libramses_amd_sw_ilp2.so timeout 900 python -m pytest tests/test_amr_tiles_gpu.py tests/test_sweep_gpu.py -m gpu -q --timeout 600 -x 2>&1 | tail -2 ) | tee -a gpurun_out/r06_w_ilp.txt
