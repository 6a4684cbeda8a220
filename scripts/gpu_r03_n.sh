#!/bin/bash
# Round-3 GPU session N: the one-launch overlapped sweep without the per-block L2 write-back (the polling kernel's end does
# it once): same bits? cost?
# (see scripts/gpu_r03_m.sh: a record of a session whose subject was removed again; RAMSES_AMD_OVERLAP_FENCE existed in that session only)
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 600 python -m pytest tests/test_halo_gpu.py -m gpu -q -x --timeout 300 ) > gpurun_out/pytest_n1.txt 2>&1
tail -5 gpurun_out/pytest_n1.txt | cut -c1-200
{
echo "== default (no per-block fence)"; timeout 300 python scripts/overlap_probe.py 512 2>/dev/null | grep '^{'
echo "== RAMSES_AMD_OVERLAP_FENCE=1"; RAMSES_AMD_OVERLAP_FENCE=1 timeout 300 python scripts/overlap_probe.py 512 2>/dev/null | grep '^{' | head -1
} > gpurun_out/overlap_probe_n.txt 2>&1
cut -c1-700 gpurun_out/overlap_probe_n.txt
( time timeout 600 python -m pytest tests/test_mpi_resident_gpu.py -m gpu -q -x --timeout 300 ) > gpurun_out/pytest_n2.txt 2>&1
tail -5 gpurun_out/pytest_n2.txt | cut -c1-200
( time timeout 600 python -m pytest tests/test_stated_sizes_gpu.py -m gpu -q -x --timeout 400 -k "half_size" ) > gpurun_out/pytest_n3.txt 2>&1
tail -5 gpurun_out/pytest_n3.txt | cut -c1-200
