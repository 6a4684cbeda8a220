#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_amr_tiles_gpu.py tests/test_amr_covered_gpu.py tests/test_amr_godunov_gpu.py tests/test_mpi_amr_resident_gpu.py tests/test_amr_remap_gpu.py "tests/test_baseline_sizes_gpu.py::test_fast_build_at_the_size_of_the_bench_line" -m gpu -q -s 2>&1 | grep -v "^$" | tail -8 | cut -c1-250
timeout 300 python scripts/dropin_timing.py c5 8 10 6 tiles 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); t=j['timers_s']; print(j['config'][:80],'| godunov',t.get('hydro - godunov'),'courant',t.get('courant'),'set unew',t.get('hydro - set unew'),'set uold',t.get('hydro - set uold'),'upload',t.get('hydro upload fine'),'TOTAL',t.get('TOTAL'))
"
