#!/bin/bash
# Round-3 GPU session M: the overlapped sweep as ONE launch (shell blocks first, "shell done" flag, polling lane on the
# communication stream): same bits? what does the schedule cost on one GPU?  MPI-resident runs with it.
# (the one-launch overlapped sweep -- godunov_fine_overlap, ramses_amd_signal_* -- was measured in sessions M-O, was not faster than the
#  shell + interior launches and was removed again: profiles/r03_overlap_probe.txt; what stayed is the XCD mapping box by box)
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 600 python -m pytest tests/test_halo_gpu.py tests/test_godunov_gpu.py -m gpu -q -x --timeout 300 ) > gpurun_out/pytest_m1.txt 2>&1
tail -5 gpurun_out/pytest_m1.txt | cut -c1-200
timeout 300 python scripts/overlap_probe.py 512 > gpurun_out/overlap_probe_m.txt 2>&1
cut -c1-600 gpurun_out/overlap_probe_m.txt
( time timeout 600 python -m pytest tests/test_mpi_resident_gpu.py -m gpu -q -x --timeout 300 ) > gpurun_out/pytest_m2.txt 2>&1
tail -5 gpurun_out/pytest_m2.txt | cut -c1-200
( time timeout 600 python -m pytest tests/test_stated_sizes_gpu.py -m gpu -q -x --timeout 400 -k "half_size" ) > gpurun_out/pytest_m3.txt 2>&1
tail -5 gpurun_out/pytest_m3.txt | cut -c1-200
