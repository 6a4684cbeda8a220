#!/bin/bash
# MHD sweep: per-kernel times and counters (round 4)
mkdir -p gpurun_out
R=$PWD
export TMPDIR=/tmp
O=$R/gpurun_out/${OUTNAME:-r04_mhd_prof.txt}
: > $O
setvar() { # variant name -> environment (the variants of round 4 -- tiles, planes, emf3 -- were measured and removed)
  export RAMSES_AMD_MHD_VARIANT=$1
}
for v in ${VARIANTS:-default}; do
  setvar $v
  for l in 7 8; do python scripts/mhd_probe.py $l 5 >> $O 2>&1; done
  python scripts/mhd_probe.py 7 5 llf llf >> $O 2>&1
done
cd /tmp
for v in ${VARIANTS:-default}; do
  rm -rf $R/gpurun_out/prof_mhd_$v
  setvar $v
  timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $R/gpurun_out/prof_mhd_$v/trace -o t -- python $R/scripts/mhd_probe.py 8 5 > $R/gpurun_out/prof_mhd_$v.log 2>&1
  f=$(find $R/gpurun_out/prof_mhd_$v/trace -name "*kernel_stats.csv" | head -1)
  echo "== kernel stats, variant $v, level 8 ==" >> $O
  grep -i "mhd\|Name" "$f" | sed "s/(anonymous namespace):://g" | cut -d, -f1-5 | cut -c1-200 >> $O
done
if [ "${PMC:-1}" = 1 ]; then
setvar ${PMC_VARIANT:-default}
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $grp | cut -d' ' -f1)
  rm -rf $R/gpurun_out/prof_mhd_pmc_$tag
  timeout 300 rocprofv3 --pmc $grp --kernel-include-regex mhd -f csv -d $R/gpurun_out/prof_mhd_pmc_$tag -o c -- python $R/scripts/mhd_probe.py 8 2 >> $R/gpurun_out/prof_mhd_pmc.log 2>&1
  f=$(find $R/gpurun_out/prof_mhd_pmc_$tag -name "*counter_collection.csv" | head -1)
  echo "== pmc $grp (level 8, per launch mean) ==" >> $O
  python - "$f" >> $O <<'PY'
import csv, re, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    m = re.search(r"mhd_\w+(<\w+>)?", r["Kernel_Name"])
    acc[(m.group(0) if m else r["Kernel_Name"][:28], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(acc.items()):
    print("  %-30s %-22s mean=%.6g n=%d" % (k, c, sum(v) / len(v), len(v)))
PY
done
fi
cat $O
