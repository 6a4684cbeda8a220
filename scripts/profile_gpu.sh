#!/bin/bash
# Profile the bench on the GPU box with rocprofv3 (run through gpurun):
#   scripts/profile_gpu.sh TAG [bench args...]
# Writes raw output under gpurun_out/prof_TAG/ and a compact summary
# gpurun_out/prof_TAG/summary.txt (copy that into profiles/).
# Counter passes are separate runs with --pmc only (never combined with
# sys/runtime/hip/hsa tracing).
set -u
TAG=${1:-run}; shift || true
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py --no-cpu-baseline $*"
echo "== bench (unprofiled) ==" > $OUT/log.txt
$BENCH --steps 10 --warmup 3 >> $OUT/log.txt 2>&1
echo "== kernel trace ==" >> $OUT/log.txt
rocprofv3 --kernel-trace --stats -f csv -d $OUT/trace -o t -- $BENCH --steps 10 --warmup 3 >> $OUT/log.txt 2>&1
pass() { # name counters...
  local name=$1; shift
  echo "== pmc $name: $* ==" >> $OUT/log.txt
  timeout 600 rocprofv3 --pmc "$@" --kernel-include-regex "${KREGEX:-godunov}" -f csv -d $OUT/$name -o c -- $BENCH --steps 2 --warmup 1 >> $OUT/log.txt 2>&1
}
if [ "${PASSES:-all}" = traffic ]; then
  # only what roofline.traffic needs: FETCH_SIZE, WRITE_SIZE and the calibration pass
  pass pmc_fetch FETCH_SIZE
  pass pmc_write WRITE_SIZE
  KREGEX=courant pass pmc_cal_fetch FETCH_SIZE
  python $REPO/scripts/summarize_prof.py $OUT "$@" > $OUT/summary.txt 2>&1
  cat $OUT/summary.txt
  exit 0
fi
pass pmc_sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
pass pmc_sq2 SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS
pass pmc_fetch FETCH_SIZE
pass pmc_write WRITE_SIZE
pass pmc_grbm GRBM_GUI_ACTIVE GRBM_COUNT
pass pmc_tcc TCC_HIT_sum TCC_MISS_sum
# FETCH_SIZE calibration on a kernel of known traffic with the same access shape
# (8 B/lane coalesced rows): courant_kernel reads exactly nvar*N*8 bytes
KREGEX=courant pass pmc_cal_fetch FETCH_SIZE
python $REPO/scripts/summarize_prof.py $OUT "$@" > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
