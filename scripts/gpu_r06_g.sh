#!/bin/bash
# Round-6 GPU session G: parity after (i) the XCD-slab cell order of the MHD stencil kernels, (ii) park + keep in the fast plain
# kernel, (iii) the single-oct shortcut for small levels and the deferred error check; the gravity certificates; bench with the
# MHD leg.
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 2400 python -m pytest tests/test_mhd_gpu.py tests/test_mhd_dropin_gpu.py tests/test_mhd_amr_gpu.py tests/test_godunov_gpu.py tests/test_amr_godunov_gpu.py \
    tests/test_amr_tiles_gpu.py tests/test_baseline_sizes_gpu.py tests/test_fast_certificate_gpu.py tests/test_mpi_amr_resident_gpu.py tests/test_amr_remap_gpu.py \
    -m gpu -q --timeout 1200 --durations=8 ) > gpurun_out/r06_g_pytest.txt 2>&1
grep -v "^$" gpurun_out/r06_g_pytest.txt | tail -40 | cut -c1-300
( time timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline ) > gpurun_out/r06_g_bench.txt 2>&1
python - <<'PY'
import json
l=[x for x in open('gpurun_out/r06_g_bench.txt') if x.startswith('{')]
if l:
    d=json.loads(l[-1])
    print('dense fast frac %.4f ms %.3f strict %.4f' % (d['roofline']['frac'], d['ms_per_step'], d['strict_build']['frac']))
    for k in ('amr_sweep','amr_sweep_partial','amr_sweep_covered'):
        a=d.get(k)
        if a: print(k, 'strict ms %.3f frac %.3f' % (a['ms_per_sweep'], a['roofline']['frac']), 'fast ms %.3f frac %.3f' % (a['fast_arithmetic']['ms_per_sweep'], a['fast_arithmetic']['frac']), 'tree ms %.3f' % a['tree_walking_ms_per_sweep'])
    print('c5', d['amr_c5_shape'].get('production'), d['amr_c5_shape'].get('all_levels_on_tiles'))
    print('mhd', json.dumps(d.get('mhd_sweep'))[:900])
    print('vcycle', d['vcycle']['ms_per_vcycle'], d['vcycle']['roofline']['frac'])
else:
    print(open('gpurun_out/r06_g_bench.txt').read()[-3000:])
PY
for lev in 7 8; do timeout 300 python scripts/mhd_probe.py $lev 2>&1 | grep -v amdgpu.ids | tail -3; done | cut -c1-300 | tee gpurun_out/r06_g_mhd_probe.txt
