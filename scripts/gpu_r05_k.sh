#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest "tests/test_baseline_sizes_gpu.py::test_c5_levels_7_9_checksum" "tests/test_baseline_sizes_gpu.py::test_c5_levels_7_10_on_8_ranks_equals_the_mpi_reference" tests/test_amr_godunov_gpu.py tests/test_amr_remap_gpu.py tests/test_walls_resident_gpu.py -m gpu -q 2>&1 | grep -v "^$" | tail -6 | cut -c1-250
timeout 200 python scripts/dropin_timing.py c5 8 10 6 tiles 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); t=j['timers_s']; print(j['config'][:80],'| godunov',t.get('hydro - godunov'),'courant',t.get('courant'),'flag',t.get('flag'),'TOTAL',t.get('TOTAL'))
"
