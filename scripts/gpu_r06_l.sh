#!/bin/bash
# Round-6 GPU session L: force_fine under MPI with the acceleration left on the device -- every self-gravitating MPI test
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 1700 python -m pytest tests/test_mpi_amr_gravity_gpu.py tests/test_mpi_uniform_gravity_gpu.py tests/test_mpi_amr_resident_gpu.py -m gpu -q --timeout 900 --durations=8 ) > gpurun_out/r06_l_pytest.txt 2>&1
grep -v "^$" gpurun_out/r06_l_pytest.txt | tail -40 | cut -c1-300
