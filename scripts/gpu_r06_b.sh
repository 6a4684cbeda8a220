#!/bin/bash
# Round-6 GPU session B: (1) crash probe of the MHD sweep of AMR levels with per-stage syncs; (2) the tile tests incl. the regrid
# that outgrows the kept tiles; (3) bench legs (late base-state load in the marching loop) + kernel trace of the AMR legs.
mkdir -p gpurun_out
export TMPDIR=/tmp
( RAMSES_AMD_DEBUG_SYNC=1 timeout 300 python scripts/mhd_amr_probe.py 4 ) > gpurun_out/r06_b_mhd_probe.txt 2>&1
tail -30 gpurun_out/r06_b_mhd_probe.txt | cut -c1-300
( time timeout 1200 python -m pytest tests/test_amr_tiles_gpu.py tests/test_godunov_gpu.py -m gpu -q --timeout 900 --durations=5 -x ) > gpurun_out/r06_b_pytest_tiles.txt 2>&1
tail -15 gpurun_out/r06_b_pytest_tiles.txt | cut -c1-300
( time timeout 600 python bench.py --steps 10 --warmup 3 --vcycle-level 0 --mhd-level 0 --no-cpu-baseline ) > gpurun_out/r06_b_bench.txt 2>&1
python - <<'PY'
import json
l=[x for x in open('gpurun_out/r06_b_bench.txt') if x.startswith('{')]
if l:
    d=json.loads(l[-1])
    print('dense fast frac %.4f ms %.3f strict %.4f' % (d['roofline']['frac'], d['ms_per_step'], d['strict_build']['frac']))
    for k in ('amr_sweep','amr_sweep_partial','amr_sweep_covered'):
        a=d.get(k)
        if a: print(k, 'strict ms %.3f frac %.3f' % (a['ms_per_sweep'], a['roofline']['frac']), 'fast ms %.3f frac %.3f' % (a['fast_arithmetic']['ms_per_sweep'], a['fast_arithmetic']['frac']), 'tree ms %.3f' % a['tree_walking_ms_per_sweep'])
else:
    print(open('gpurun_out/r06_b_bench.txt').read()[-2000:])
PY
cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_b -o amr -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --vcycle-level 0 --mhd-level 0 --no-cpu-baseline > /tmp/prof_b.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(ls /tmp/prof_b/*/*kernel_stats.csv /tmp/prof_b/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] && head -25 "$f" | cut -c1-260 > gpurun_out/r06_b_amr_kernel_stats.csv
cat gpurun_out/r06_b_amr_kernel_stats.csv
