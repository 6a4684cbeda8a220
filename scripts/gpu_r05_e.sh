#!/bin/bash
# Round-5 GPU session E: the two tests that failed in D, the C5 checksum test with the tile counters, drop-in timing of C5 on the
# three layouts, kernel statistics of the AMR legs of bench.py
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 400 python -m pytest tests/test_rccl_gpu.py "tests/test_baseline_sizes_gpu.py::test_c5_levels_7_9_checksum" tests/test_amr_tiles_gpu.py -m gpu -q -s 2>&1 | grep -v "^$" | tail -12 | cut -c1-250
( timeout 300 python scripts/dropin_timing.py c5 7 9 8 tiles; timeout 500 python scripts/dropin_timing.py c5 8 10 6 tiles ) > gpurun_out/r05_dropin_c5_tiles.txt 2>&1
cut -c1-1500 gpurun_out/r05_dropin_c5_tiles.txt
