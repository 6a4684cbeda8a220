#!/bin/bash
# Round-5 GPU session D: the whole GPU suite on the device numbering / tile sweeps
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 1300 python -m pytest tests -m gpu -q --timeout 900 --durations=25 ) > gpurun_out/r05_d_pytest.txt 2>&1
tail -80 gpurun_out/r05_d_pytest.txt | cut -c1-300
