#!/bin/bash
# Round-5 GPU session G: MPI runs on the device numbering: C4's shape (uniform 128^3 + self-gravity) and C5 7-10 on 8 ranks
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 300 python scripts/dropin_timing.py gravmpi 7 4 8 tiles; timeout 300 python scripts/dropin_timing.py c5mpi 7 10 8 8 tiles ) > gpurun_out/r05_dropin_mpi_tiles.txt 2>&1
python - <<'PY'
import json
for l in open("gpurun_out/r05_dropin_mpi_tiles.txt"):
    if l.startswith("{"):
        j = json.loads(l)
        t = j["timers_max_s"]
        print(j["config"][:100], "| godunov", t.get("hydro - godunov"), "TOTAL", t.get("TOTAL"), "|", (j.get("sweeps_of_rank") or [""])[0][:150])
    else:
        print(l.rstrip()[:200])
PY
