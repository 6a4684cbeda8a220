#!/bin/bash
# Round-4 GPU session A: (1) the gravity step of an MPI run on the GPUs -- rho_fine's deposit with its exchanges and force_fine of
# the AMR levels under MPI, config C5 WITH multigrid live against the MPI reference; (2) A/B of the sweep with the z neighbours in
# registers (SWEEP_ZREG) on 12-, 14- and 16-row tiles; (3) stall-breakdown PMC of the shipped sweep kernel.
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_rho_fine_gpu.py tests/test_mpi_amr_gravity_gpu.py "tests/test_baseline_sizes_gpu.py::test_c5_with_multigrid_equals_the_mpi_reference" \
    -m gpu -q --timeout 600 --durations=8 ) > gpurun_out/r04_a_pytest.txt 2>&1
tail -25 gpurun_out/r04_a_pytest.txt | cut -c1-300
AB_CFGS="zreg:12,128 zreg:14,86 zreg:16,103 zreg:16,57 zreg:16,128 base:12,64" timeout 900 python scripts/ab_sweep.py base zreg 2>&1 | tee gpurun_out/r04_ab_sweep.txt
timeout 900 scripts/pmc_sweep.sh r04_shipped --amr-level 0 --stress-steps 0 > gpurun_out/r04_sweep_pmc.txt 2>&1
tail -70 gpurun_out/r04_sweep_pmc.txt | cut -c1-200
