#!/bin/bash
# Round-6 GPU session R: the surface pass beside the marching kernel (side stream): parity, A/B on the shell level
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_amr_tiles_gpu.py tests/test_fast_certificate_gpu.py -m gpu -q --timeout 900 -k "tiles or amr" ) > gpurun_out/r06_r_pytest.txt 2>&1
grep -v "^$" gpurun_out/r06_r_pytest.txt | tail -8 | cut -c1-250
{
for ov in 1 0; do for cfg in "9 partial" "8 covered"; do
  echo "# RAMSES_AMD_SURFACE_OVERLAP=$ov $cfg"
  RAMSES_AMD_SURFACE_OVERLAP=$ov timeout 300 python - $cfg <<'PY' 2>/dev/null | tail -1
import sys, json
sys.path.insert(0, ".")
import torch, bench
torch.cuda.init()
o = bench.amr_resident_bench(int(sys.argv[1]), steps=7, kind=sys.argv[2])
print(json.dumps({k: o.get(k) for k in ("ms_per_sweep", "fast_arithmetic", "roofline")})[:600])
PY
done; done
} | tee gpurun_out/r06_r_overlap_ab.txt
