#!/bin/bash
# Round-6 GPU session O: one-level self-gravitating MPI runs with rho, phi and f resident
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 1700 python -m pytest tests/test_mpi_uniform_gravity_gpu.py -m gpu -q --timeout 900 --durations=8 ) > gpurun_out/r06_o_pytest.txt 2>&1
grep -v "^$" gpurun_out/r06_o_pytest.txt | tail -40 | cut -c1-300
