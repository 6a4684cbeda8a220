#!/bin/bash
# Round-6 GPU session AF: the bench line with the new hllc_sweep leg (bench.py changed after the closing run)
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python bench.py --no-cpu-baseline --amr-level 0 --mhd-level 7 --vcycle-level 9 > gpurun_out/r06_af_bench.json 2> gpurun_out/r06_af_bench.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r06_af_bench.json"))
print("sweep", d["ms_per_step"], d["roofline"]["frac"], "hllc", d.get("hllc_sweep"))
PY
