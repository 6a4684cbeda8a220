#!/bin/bash
# Round-6 GPU session M: passive scalars / the Newton solver on tiles; the sweep split into one translation unit per slope type;
# the uniform self-gravitating MPI runs with the acceleration resident
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 1700 python -m pytest tests/test_amr_tiles_gpu.py tests/test_mpi_uniform_gravity_gpu.py tests/test_godunov_gpu.py tests/test_fast_certificate_gpu.py tests/test_uniform_options_resident_gpu.py tests/test_amr_godunov_gpu.py -m gpu -q --timeout 900 --durations=8 ) > gpurun_out/r06_m_pytest.txt 2>&1
grep -v "^$" gpurun_out/r06_m_pytest.txt | tail -40 | cut -c1-300
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-1500 | tee gpurun_out/r06_m_bench.txt
