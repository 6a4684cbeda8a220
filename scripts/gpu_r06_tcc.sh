#!/bin/bash
# L2 behaviour of the masked kernel (level in tiles) next to the plain kernel of the same size: TCC hits / misses per dispatch
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/tcc
mkdir -p $OUT
cd /tmp
( timeout 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-include-regex 'godunov_sweep_kernel' -f csv -d $OUT/mask -o c -- python $R/scripts/amr_tile_probe.py 8 full 3 ) > $OUT/mask.log 2>&1
( timeout 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-include-regex 'godunov_sweep_kernel' -f csv -d $OUT/plain -o c -- python $R/bench.py --n 256 --steps 3 --warmup 1 --vcycle-level 0 --mhd-level 0 --amr-level 0 --no-cpu-baseline ) > $OUT/plain.log 2>&1
cd $R
python3 - <<'PY'
import csv,glob,collections
for tag in ('mask','plain'):
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob('gpurun_out/tcc/%s/**/*counter_collection.csv'%tag,recursive=True):
        for r in csv.DictReader(open(f)):
            agg[r['Kernel_Name'][:95]][r['Counter_Name']].append(float(r['Counter_Value']))
    for kn,cs in sorted(agg.items()):
        print(tag,kn,' '.join('%s=%.4g(n=%d)'%(k,sum(v)/len(v),len(v)) for k,v in sorted(cs.items())))
    if not agg: print(tag,'NO DATA',open('gpurun_out/tcc/%s.log'%tag).read()[-400:])
PY
