#!/bin/bash
# Round-6 GPU session AI: the tile sweep of AMR levels with HLLC (the MASK instantiations of the fused flux; bench.py's legs are LLF)
mkdir -p gpurun_out
export TMPDIR=/tmp
{
for r in llf hllc; do for cfg in "8 full" "8 covered" "9 partial"; do
  RAMSES_AMD_BENCH_AMR_RIEMANN=$r timeout 300 python - $cfg <<'PY' 2>/dev/null | tail -1
import sys, os
sys.path.insert(0, ".")
import torch, bench
torch.cuda.init()
o = bench.amr_resident_bench(int(sys.argv[1]), steps=7, kind=sys.argv[2])
print("%s %s %s: strict %.3f ms (%.4f)  fast %.3f ms (%.4f)" % (os.environ["RAMSES_AMD_BENCH_AMR_RIEMANN"], sys.argv[1], sys.argv[2], o["ms_per_sweep"], o["roofline"]["frac"], o["fast_arithmetic"]["ms_per_sweep"], o["fast_arithmetic"]["frac"]))
PY
done; done
} | tee gpurun_out/r06_ai_tiles_hllc.txt
