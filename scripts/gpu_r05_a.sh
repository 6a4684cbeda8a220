#!/bin/bash
# Round-5 GPU session A: the device's own oct numbering (tiles) and the dense sweep of AMR levels on it.
# (1) the new parity tests against the oracle, the covered-level test, rho_fine (oct centres in device numbers), the live
# patched-program runs that go through the resident path; (2) the AMR legs of bench.py.
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests/test_amr_tiles_gpu.py tests/test_amr_covered_gpu.py tests/test_rho_fine_gpu.py \
    "tests/test_amr_godunov_gpu.py" tests/test_amr_remap_gpu.py \
    -m gpu -q --timeout 900 --durations=10 ) > gpurun_out/r05_a_pytest.txt 2>&1
tail -60 gpurun_out/r05_a_pytest.txt | cut -c1-400
( time timeout 600 python bench.py --steps 5 --warmup 2 --vcycle-level 0 --mhd-level 0 --no-cpu-baseline ) > gpurun_out/r05_a_bench.txt 2>&1
tail -5 gpurun_out/r05_a_bench.txt | cut -c1-6000
