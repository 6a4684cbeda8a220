#!/bin/bash
# Round-6 GPU session D: MHD on AMR levels live A/B (all cases), the fast certificate on an AMR run, kernel trace of the AMR legs.
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 2400 python -m pytest tests/test_mhd_amr_gpu.py -m gpu -q --timeout 1200 -s ) > gpurun_out/r06_d_pytest_mhd_amr.txt 2>&1
grep -v "^$" gpurun_out/r06_d_pytest_mhd_amr.txt | tail -30 | cut -c1-400
( time timeout 2400 python -m pytest "tests/test_fast_certificate_gpu.py::test_default_mode_amr_run_live_ab" -m gpu -q --timeout 2000 -s ) > gpurun_out/r06_d_pytest_fastamr.txt 2>&1
grep -v "^$" gpurun_out/r06_d_pytest_fastamr.txt | tail -15 | cut -c1-400
rm -rf gpurun_out/prof_d
rocprofv3 --kernel-trace --stats -f csv -d gpurun_out/prof_d -o amr -- python bench.py --steps 3 --warmup 1 --vcycle-level 0 --mhd-level 0 --no-cpu-baseline > gpurun_out/prof_d.log 2>&1
python scripts/kstats.py gpurun_out/prof_d 30 > gpurun_out/r06_d_amr_kernel_stats.txt 2>&1
rm -rf gpurun_out/prof_d gpurun_out/prof_d.log
cat gpurun_out/r06_d_amr_kernel_stats.txt | cut -c1-230
