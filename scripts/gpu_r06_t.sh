#!/bin/bash
# Round-6 GPU session T: the order the surface pass visits its events in (key of the sort x lane map), shell level
mkdir -p gpurun_out
export TMPDIR=/tmp
{
for ord in face tile oct; do for lanes in e q; do
  echo "# RAMSES_AMD_EVENT_ORDER=$ord RAMSES_AMD_EVENT_LANES=$lanes"
  RAMSES_AMD_EVENT_ORDER=$ord RAMSES_AMD_EVENT_LANES=$lanes timeout 300 python - 9 partial <<'PY' 2>/dev/null | tail -1
import sys, json
sys.path.insert(0, ".")
import torch, bench
torch.cuda.init()
o = bench.amr_resident_bench(int(sys.argv[1]), steps=7, kind=sys.argv[2])
print("strict %.3f ms (%.4f)  fast %.3f ms (%.4f)" % (o["ms_per_sweep"], o["roofline"]["frac"], o["fast_arithmetic"]["ms_per_sweep"], o["fast_arithmetic"]["frac"]))
PY
done; done
} | tee gpurun_out/r06_t_event_order.txt
( timeout 600 python -m pytest tests/test_amr_tiles_gpu.py -m gpu -q --timeout 600 -x 2>&1 | tail -2 )
for ord in tile oct; do RAMSES_AMD_EVENT_ORDER=$ord RAMSES_AMD_EVENT_LANES=q timeout 600 python -m pytest tests/test_amr_tiles_gpu.py -m gpu -q --timeout 600 -x -k "levels_in_tiles or passive" 2>&1 | tail -1; done
