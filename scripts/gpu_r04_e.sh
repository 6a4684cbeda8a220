#!/bin/bash
# Round-4 GPU session E: the in-place dense sweep of covered levels (index + mask bricks), the compact MHD trace
# (47 numbers per cell instead of 144), bench.py on two ranks after the census fix.
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 1200 python -m pytest tests/test_amr_covered_gpu.py tests/test_mhd_gpu.py tests/test_mhd_dropin_gpu.py tests/test_bench_multirank_gpu.py \
    tests/test_amr_godunov_gpu.py tests/test_baseline_sizes_gpu.py tests/test_dropin_gpu.py tests/test_amr_remap_gpu.py tests/test_godunov_gpu.py \
    -m gpu -q --timeout 600 --durations=6 ) > gpurun_out/r04_e_pytest.txt 2>&1
tail -30 gpurun_out/r04_e_pytest.txt | cut -c1-300
timeout 600 python bench.py --no-cpu-baseline --vcycle-level 0 --stress-steps 0 --steps 5 --warmup 2 > gpurun_out/r04_e_bench.json 2> gpurun_out/r04_e_bench.err
python - <<'PY'
import json
d = json.loads([l for l in open('gpurun_out/r04_e_bench.json') if l.startswith('{')][-1])
print('sweep', d['roofline']['kernel_ms'], d['roofline']['frac'], 'strict', d['strict_build']['kernel_ms'])
for k in ('amr_sweep', 'amr_sweep_partial', 'amr_sweep_covered', 'mhd_sweep'):
    v = d.get(k, {})
    print(k, {kk: v.get(kk) for kk in ('value', 'ms_per_sweep', 'tree_walking_ms_per_sweep', 'dense_sweeps_taken', 'error')}, v.get('roofline', {}).get('frac'))
PY
