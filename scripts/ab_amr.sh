mkdir -p gpurun_out
R=$PWD
for v in w4 w5 w6 w7; do
  echo "== $v"
  RAMSES_AMD_LIB=$R/ramses_amd/lib/ab/libramses_amd_amr_$v.so timeout 120 python scripts/amr_probe.py 8 morton 2>&1 | tail -1 | cut -c1-140
done > gpurun_out/ab_amr.txt 2>&1
cd /tmp; export TMPDIR=/tmp
RAMSES_AMD_LIB=$R/ramses_amd/lib/ab/libramses_amd_amr_w5.so rocprofv3 --kernel-trace --stats -f csv -d $R/gpurun_out/prof_amr -o t -- python $R/scripts/amr_probe.py 8 morton > $R/gpurun_out/prof_amr.log 2>&1
python $R/scripts/kstats.py $R/gpurun_out/prof_amr 6 >> $R/gpurun_out/ab_amr.txt
cat $R/gpurun_out/ab_amr.txt | cut -c1-200
