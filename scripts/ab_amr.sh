# A/B of tree-walking sweep variants built by scripts/build_ab_amr.py:  bash scripts/ab_amr.sh TAG [TAG...]
mkdir -p gpurun_out
R=$PWD
for v in "$@"; do
  for order in morton scrambled; do
    echo "== $v $order"
    RAMSES_AMD_LIB=$R/ramses_amd/lib/ab/libramses_amd_amr_$v.so timeout 120 python scripts/amr_probe.py 8 $order 2>&1 | tail -1 | cut -c1-140
  done
done > gpurun_out/ab_amr.txt 2>&1
cat gpurun_out/ab_amr.txt
