#!/bin/bash
# Round-5 GPU session B: where does a sweep of a level in tiles spend its time?  rocprofv3 kernel statistics of the covered and
# the partial case, the regrid test again.
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp
R=$GRAFT_REPO_ROOT
for kind in covered partial; do
  lvl=8; [ $kind = partial ] && lvl=9
  rm -rf /tmp/prof_$kind
  timeout 100 rocprofv3 --kernel-trace --stats -f csv -d /tmp/prof_$kind -o p -- python $R/scripts/amr_tile_probe.py $lvl $kind 5 > $R/gpurun_out/r05_b_probe_$kind.txt 2>&1
  tail -3 $R/gpurun_out/r05_b_probe_$kind.txt | cut -c1-400
  python $R/scripts/kstats.py /tmp/prof_$kind 14 | tee $R/gpurun_out/r05_b_kstats_$kind.txt
done
cd $R

