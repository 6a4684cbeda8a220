#!/bin/bash
# One GPU-box session: the whole -m gpu suite (log kept), then the tree-walking sweep probe (Z-order and scrambled oct
# numbering).  Everything lands in gpurun_out/.
#   gpurun --timeout 900 -- 'bash scripts/gpu_check.sh'
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 700 python -m pytest tests -m gpu -q --timeout 300 ) > gpurun_out/pytest_gpu.txt 2>&1
tail -5 gpurun_out/pytest_gpu.txt | cut -c1-300
{
  echo "# tree-walking sweep, fully refined 256^3 tree (scripts/amr_probe.py 8)"
  timeout 200 python scripts/amr_probe.py 8 morton 2>&1 | tail -1
  timeout 200 python scripts/amr_probe.py 8 scrambled 2>&1 | tail -1
} > gpurun_out/amr_probe.txt 2>&1
cat gpurun_out/amr_probe.txt
