#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_lowdim_dropin_gpu.py tests/test_embedded_ndim_gpu.py tests/test_patch_ab_switch.py -m gpu -q 2>&1 | grep -v "^$" | tail -40 | cut -c1-300
