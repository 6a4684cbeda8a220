#!/bin/bash
# PMC characterisation of one kernel family (counter passes only, never combined with tracing):
#   scripts/pmc_kernel.sh TAG KERNEL_REGEX -- <command that runs it>
# e.g. scripts/pmc_kernel.sh vcycle mg_smooth_fused -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline
set -u
TAG=$1; KREGEX=$2; shift 3
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
CMD=("$@")
cd /tmp && export TMPDIR=/tmp
pass() {
  local name=$1; shift
  (cd $REPO && timeout ${PMC_TIMEOUT:-300} rocprofv3 --pmc "$@" --kernel-include-regex "$KREGEX" -f csv -d $OUT/$name -o c -- "${CMD[@]}") > $OUT/$name.log 2>&1
}
pass a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES
pass c SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS
pass f FETCH_SIZE
pass w WRITE_SIZE
pass g GRBM_GUI_ACTIVE
python3 - "$OUT" <<'PY'
import csv,glob,sys,collections
out=sys.argv[1]
for d in sorted(glob.glob(out+'/?')):
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d+'/**/*counter_collection.csv',recursive=True):
        for r in csv.DictReader(open(f)):
            agg[r['Kernel_Name'][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
    for kn,cs in sorted(agg.items()):
        for k,v in sorted(cs.items()):
            print(d[-1],kn,k,'mean=%.6g'%(sum(v)/len(v)),'max=%.6g'%max(v),'n=%d'%len(v))
    if not agg:
        print(d[-1],'NO DATA:',open(d+'.log').read()[-300:].replace('\n',' | '))
PY
