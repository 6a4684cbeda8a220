#!/bin/bash
# Stall breakdown of the sweep kernel: PMC passes only (no tracing), one pass per
# counter group.  scripts/pmc_sweep.sh TAG [bench args...]
set -u
TAG=${1:-pmc}; shift || true
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py --no-cpu-baseline --vcycle-level 0 $*"
rocprofv3 -L > $OUT/counters_list.txt 2>&1
pass() {
  local name=$1; shift
  timeout 300 rocprofv3 --pmc "$@" --kernel-include-regex godunov -f csv -d $OUT/$name -o c -- $BENCH --steps 2 --warmup 1 > $OUT/$name.log 2>&1
}
pass a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES
pass b SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM
pass c SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT
pass d SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_THREAD_CYCLES_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_FMA_F64
pass e SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_BRANCH SQ_INSTS_SENDMSG
pass f SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES
pass g TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TA_TCP_STATE_READ_sum TA_BUSY_avr TA_TA_BUSY_sum
python3 - "$OUT" <<'PY'
import csv,glob,sys,collections
out=sys.argv[1]
for d in sorted(glob.glob(out+'/?')):
    agg=collections.defaultdict(list)
    for f in glob.glob(d+'/**/*counter_collection.csv',recursive=True):
        for r in csv.DictReader(open(f)):
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in sorted(agg.items()):
        print(d[-1],k,'mean=%.6g'%(sum(v)/len(v)),'n=%d'%len(v))
    if not agg:
        print(d[-1],'NO DATA:',open(d+'.log').read()[-300:].replace('\n',' | '))
PY
