#!/bin/bash
# Round-6 GPU session A: (1) the tile sweep without scratch (2-slot LDS ring, parked partial update, surface pass instead of
# in-loop flux records) against the ORACLE + the dense brick sweep's own parity tests (the ring change touches every variant);
# (2) the MHD sweep of AMR levels, live A/B; (3) the AMR legs of bench.py (strict + fast); (4) kernel trace of the AMR legs.
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 1200 python -m pytest tests/test_amr_tiles_gpu.py tests/test_godunov_gpu.py tests/test_embedded_ndim_gpu.py \
    -m gpu -q --timeout 900 --durations=8 -x ) > gpurun_out/r06_a_pytest_tiles.txt 2>&1
tail -25 gpurun_out/r06_a_pytest_tiles.txt | cut -c1-300
( time timeout 1200 python -m pytest tests/test_mhd_amr_gpu.py -m gpu -q --timeout 900 -s ) > gpurun_out/r06_a_pytest_mhd_amr.txt 2>&1
tail -40 gpurun_out/r06_a_pytest_mhd_amr.txt | cut -c1-400
( time timeout 600 python bench.py --steps 10 --warmup 3 --vcycle-level 0 --mhd-level 0 --no-cpu-baseline ) > gpurun_out/r06_a_bench.txt 2>&1
tail -3 gpurun_out/r06_a_bench.txt | cut -c1-7000
