#!/bin/bash
# Round-6 GPU session N: the tile tests again (Newton solver tolerance, the live NVAR = 7 A/B)
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 1200 python -m pytest tests/test_amr_tiles_gpu.py -m gpu -q --timeout 900 --durations=5 ) > gpurun_out/r06_n_pytest.txt 2>&1
grep -v "^$" gpurun_out/r06_n_pytest.txt | tail -30 | cut -c1-300
