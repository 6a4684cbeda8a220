#!/bin/bash
# Round-6 GPU session C: MHD on AMR levels (probe + live A/B), the ordered norm of the distributed multigrid, NDIM<3 counters,
# tiles with the base state from uold, the AMR bench legs + their kernel trace.
mkdir -p gpurun_out
export TMPDIR=/tmp
( RAMSES_AMD_DEBUG_SYNC=1 timeout 300 python scripts/mhd_amr_probe.py 4 ) > gpurun_out/r06_c_mhd_probe.txt 2>&1
tail -8 gpurun_out/r06_c_mhd_probe.txt | cut -c1-300
( time timeout 1500 python -m pytest tests/test_mhd_amr_gpu.py -m gpu -q --timeout 900 -s -x ) > gpurun_out/r06_c_pytest_mhd_amr.txt 2>&1
tail -30 gpurun_out/r06_c_pytest_mhd_amr.txt | cut -c1-400
( time timeout 1200 python -m pytest tests/test_mgdist_order_gpu.py tests/test_lowdim_dropin_gpu.py tests/test_amr_tiles_gpu.py tests/test_mpi_uniform_gravity_gpu.py \
    -m gpu -q --timeout 900 --durations=5 ) > gpurun_out/r06_c_pytest_misc.txt 2>&1
tail -25 gpurun_out/r06_c_pytest_misc.txt | cut -c1-300
( time timeout 600 python bench.py --steps 10 --warmup 3 --vcycle-level 0 --mhd-level 0 --no-cpu-baseline ) > gpurun_out/r06_c_bench.txt 2>&1
python - <<'PY'
import json
l=[x for x in open('gpurun_out/r06_c_bench.txt') if x.startswith('{')]
if l:
    d=json.loads(l[-1])
    print('dense fast frac %.4f ms %.3f strict %.4f' % (d['roofline']['frac'], d['ms_per_step'], d['strict_build']['frac']))
    for k in ('amr_sweep','amr_sweep_partial','amr_sweep_covered'):
        a=d.get(k)
        if a: print(k, 'strict ms %.3f frac %.3f' % (a['ms_per_sweep'], a['roofline']['frac']), 'fast ms %.3f frac %.3f' % (a['fast_arithmetic']['ms_per_sweep'], a['fast_arithmetic']['frac']), 'tree ms %.3f' % a['tree_walking_ms_per_sweep'])
else:
    print(open('gpurun_out/r06_c_bench.txt').read()[-2000:])
PY
rm -rf gpurun_out/prof_c
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_c -o amr -- python bench.py --steps 3 --warmup 1 --vcycle-level 0 --mhd-level 0 --no-cpu-baseline > gpurun_out/prof_c.log 2>&1
f=$(find gpurun_out/prof_c -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -30 "$f" | cut -c1-200 > gpurun_out/r06_c_amr_kernel_stats.csv
rm -rf gpurun_out/prof_c
cat gpurun_out/r06_c_amr_kernel_stats.csv
