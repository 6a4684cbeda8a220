#!/bin/bash
# Round-4 GPU session D: fused gather / scatter of covered levels, the defrag hook (nremap > 0, mid-run snapshots), uniform
# self-gravity under MPI on the AMR-resident path, the two-rank bench line (nccl-shared: keep the whole stderr).
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 1200 python -m pytest tests/test_amr_covered_gpu.py tests/test_amr_remap_gpu.py tests/test_mpi_uniform_gravity_gpu.py \
    tests/test_bench_multirank_gpu.py tests/test_baseline_sizes_gpu.py tests/test_dropin_gpu.py tests/test_mpi_amr_gravity_gpu.py \
    -m gpu -q --timeout 600 --durations=6 ) > gpurun_out/r04_d_pytest.txt 2>&1
tail -40 gpurun_out/r04_d_pytest.txt | cut -c1-300
timeout 600 python bench.py --no-cpu-baseline --vcycle-level 0 --mhd-level 0 --stress-steps 0 --steps 5 --warmup 2 > gpurun_out/r04_d_bench.json 2> gpurun_out/r04_d_bench.err
python - <<'PY'
import json
d = json.loads([l for l in open('gpurun_out/r04_d_bench.json') if l.startswith('{')][-1])
for k in ('amr_sweep', 'amr_sweep_partial', 'amr_sweep_covered'):
    v = d.get(k, {})
    print(k, {kk: v.get(kk) for kk in ('value', 'ms_per_sweep', 'tree_walking_ms_per_sweep', 'dense_sweeps_taken', 'error')}, v.get('roofline', {}).get('frac'))
PY
