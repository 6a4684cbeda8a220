#!/bin/bash
# Round-6 GPU session AJ: the HLLC tile sweep with the generic flux (+/- the plane held in registers) against the fused one that ships
mkdir -p gpurun_out
export TMPDIR=/tmp
{
for tag in hllc_gen_keep hllc_gen_nokeep; do for cfg in "8 full" "9 partial"; do
  RAMSES_AMD_LIB=ramses_amd/lib/ab/libramses_amd_$tag.so RAMSES_AMD_BENCH_AMR_RIEMANN=hllc timeout 300 python - $cfg <<'PY' 2>/dev/null | tail -1
import sys, os
sys.path.insert(0, ".")
import torch, bench
torch.cuda.init()
o = bench.amr_resident_bench(int(sys.argv[1]), steps=7, kind=sys.argv[2])
print("%s hllc %s %s: strict %.3f ms (%.4f)  fast %.3f ms (%.4f)" % (os.environ["RAMSES_AMD_LIB"][-22:], sys.argv[1], sys.argv[2], o["ms_per_sweep"], o["roofline"]["frac"], o["fast_arithmetic"]["ms_per_sweep"], o["fast_arithmetic"]["frac"]))
PY
done; done
} | tee gpurun_out/r06_aj_tiles_hllc_generic.txt
