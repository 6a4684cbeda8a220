#!/bin/bash
# Round-4 GPU session B: (1) the MHD sweep against the compiled reference; the in-library RCCL leg of the distributed
# multigrid; bench.py's N>1 path with two ranks on the one GPU; the fast-mode certificate over the solver matrix;
# (2) A/B of the sweep variants (z neighbours in registers, unrolled by the ring period, 14/16-row tiles).
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 1200 python -m pytest tests/test_mhd_gpu.py tests/test_rccl_gpu.py tests/test_bench_multirank_gpu.py \
    "tests/test_fast_certificate_gpu.py::test_default_mode_solver_matrix_live_ab_at_64" tests/test_godunov_gpu.py \
    -m gpu -q --timeout 600 --durations=8 ) > gpurun_out/r04_b_pytest.txt 2>&1
tail -40 gpurun_out/r04_b_pytest.txt | cut -c1-300
AB_CFGS="zreg:12,128 zreg:14,86 zreg:16,103 zreg:16,57 zreg2:12,128 zreg2:16,103 base:12,64" timeout 900 python scripts/ab_sweep.py base zreg zreg2 2>&1 | tee gpurun_out/r04_ab_sweep.txt
