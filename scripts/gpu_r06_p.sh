#!/bin/bash
# Round-6 GPU session P: the whole -m gpu suite on the code of the evening; C4's shape under MPI with the bytes of the Poisson
# level arrays over PCIe (steady state on / off)
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -q --timeout 900 --durations=25 ) > gpurun_out/r06_p_pytest_gpu.txt 2>&1
grep -v "^$" gpurun_out/r06_p_pytest_gpu.txt | tail -45 | cut -c1-250
( for np in 2 8; do timeout 900 python scripts/dropin_timing.py gravmpi 7 4 $np pcie; done ) > gpurun_out/r06_p_gravmpi_pcie.txt 2>&1
cut -c1-600 gpurun_out/r06_p_gravmpi_pcie.txt
