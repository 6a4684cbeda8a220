#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 700 python -m pytest tests/test_mpi_uniform_gravity_gpu.py tests/test_bench_multirank_gpu.py tests/test_multigrid_parallel_gpu.py tests/test_amr_remap_gpu.py tests/test_amr_tiles_gpu.py -m gpu -q 2>&1 | grep -v "^$" | tail -8 | cut -c1-250
