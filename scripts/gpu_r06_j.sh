#!/bin/bash
# Round-6 GPU session J: A/B of the MHD sweep's first kernel (loads batched, tile shapes) and of the EMF kernel at three waves per SIMD
mkdir -p gpurun_out
export TMPDIR=/tmp
{
echo "# default (loads of the conversion phase batched)"; timeout 300 python scripts/mhd_probe.py 8 5 2>&1 | grep -v amdgpu.ids | tail -1
for v in nobatch emf3 t1644 t3242 t1648 t1684; do
  echo "# variant $v"; RAMSES_AMD_LIB=$PWD/ramses_amd/lib/ab/libramses_amd_$v.so timeout 300 python scripts/mhd_probe.py 8 5 2>&1 | grep -v amdgpu.ids | tail -1
done
echo "# default, level 7"; timeout 300 python scripts/mhd_probe.py 7 5 2>&1 | grep -v amdgpu.ids | tail -1
} | cut -c1-300 | tee gpurun_out/r06_j_mhd_ab.txt
( time timeout 1200 python -m pytest tests/test_mhd_gpu.py tests/test_mhd_dropin_gpu.py -m gpu -q --timeout 900 ) > gpurun_out/r06_j_pytest.txt 2>&1
grep -v "^$" gpurun_out/r06_j_pytest.txt | tail -6 | cut -c1-300
