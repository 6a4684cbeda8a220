#!/bin/bash
# Round-3 GPU session O: the one-launch overlapped sweep with the shell blocks spread round-robin over the XCDs (long
# blocks first): same bits? cost of the schedule on one GPU?  MPI-resident runs with it.
# (see scripts/gpu_r03_m.sh: a record of a session whose subject was removed again)
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 600 python -m pytest tests/test_halo_gpu.py -m gpu -q -x --timeout 300 ) > gpurun_out/pytest_o1.txt 2>&1
tail -5 gpurun_out/pytest_o1.txt | cut -c1-200
timeout 300 python scripts/overlap_probe.py 512 2>/dev/null | grep '^{' > gpurun_out/overlap_probe_o.txt 2>&1
cut -c1-700 gpurun_out/overlap_probe_o.txt
( time timeout 600 python -m pytest tests/test_mpi_resident_gpu.py -m gpu -q -x --timeout 300 ) > gpurun_out/pytest_o2.txt 2>&1
tail -5 gpurun_out/pytest_o2.txt | cut -c1-200
( time timeout 600 python -m pytest tests/test_stated_sizes_gpu.py -m gpu -q -x --timeout 400 -k "half_size" ) > gpurun_out/pytest_o3.txt 2>&1
tail -5 gpurun_out/pytest_o3.txt | cut -c1-200
