mkdir -p gpurun_out
R=$PWD
for v in "$@"; do
  echo "== $v"
  RAMSES_AMD_LIB=$R/ramses_amd/lib/ab/libramses_amd_amr_$v.so timeout 120 python scripts/amr_probe.py 8 morton 2>&1 | tail -1 | cut -c1-140
done > gpurun_out/ab_amr.txt 2>&1
cat gpurun_out/ab_amr.txt
