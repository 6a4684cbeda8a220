#!/bin/bash
# Round-4 GPU session C: (1) the new paths -- dense masked sweep of covered AMR levels, the MHD drop-in, bench.py on two ranks,
# the slope_type-3 strict default -- then the WHOLE -m gpu suite; (2) the bench line with the new legs (mhd_sweep,
# amr_sweep_covered); (3) kernel statistics + PMC of the final V-cycle.
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_amr_covered_gpu.py tests/test_mhd_dropin_gpu.py tests/test_bench_multirank_gpu.py \
    "tests/test_fast_certificate_gpu.py::test_slope_type_3_runs_strict_by_default" -m gpu -q --timeout 600 --durations=6 ) > gpurun_out/r04_c_pytest_new.txt 2>&1
tail -40 gpurun_out/r04_c_pytest_new.txt | cut -c1-300
( time timeout 1500 python -m pytest tests -m gpu -q --timeout 600 --durations=15 ) > gpurun_out/r04_c_pytest_gpu.txt 2>&1
tail -30 gpurun_out/r04_c_pytest_gpu.txt | cut -c1-300
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/r04_c_bench.json 2> gpurun_out/r04_c_bench.err
python - <<'PY'
import json
d = json.loads([l for l in open('gpurun_out/r04_c_bench.json') if l.startswith('{')][-1])
print('sweep', d['roofline']['kernel_ms'], d['roofline']['frac'], 'strict', d['strict_build']['kernel_ms'])
for k in ('vcycle', 'amr_sweep', 'amr_sweep_partial', 'amr_sweep_covered', 'mhd_sweep'):
    v = d.get(k, {})
    print(k, {kk: v.get(kk) for kk in ('value', 'ms_per_sweep', 'ms_per_vcycle', 'tree_walking_ms_per_sweep', 'dense_sweeps_taken', 'error')}, v.get('roofline', {}).get('frac'))
PY
REPO=$GRAFT_REPO_ROOT
cd /tmp
rocprofv3 --kernel-trace --stats -f csv -d /tmp/prof_vc -o vc -- python $REPO/bench.py --no-cpu-baseline --steps 3 --warmup 1 --amr-level 0 --stress-steps 0 --mhd-level 0 > /dev/null 2>&1
f=$(find /tmp/prof_vc -name "*kernel_stats.csv" | head -1)
cp "$f" $REPO/gpurun_out/r04_vcycle_kernel_stats.csv
head -14 "$f" | cut -c1-200
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --kernel-include-regex "mg_smooth_fused|mg_interp|mg_restrict|mg_gs_kernel|mg_residual" -f csv -d /tmp/pmc_vc_$i -o c -- python $REPO/bench.py --no-cpu-baseline --steps 2 --warmup 1 --amr-level 0 --stress-steps 0 --mhd-level 0 > /tmp/pmc_vc_$i.log 2>&1
done
python3 - <<'PY' > $REPO/gpurun_out/r04_vcycle_pmc_raw.txt
import csv, glob, collections
for d in sorted(glob.glob('/tmp/pmc_vc_*/')):
    agg = collections.defaultdict(list)
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            grid = r.get('Grid_Size', '')
            agg[(r['Kernel_Name'][:48], grid, r['Counter_Name'])].append(float(r['Counter_Value']))
    for k, v in sorted(agg.items()):
        print(k[0], 'grid', k[1], k[2], 'mean=%.6g' % (sum(v) / len(v)), 'n=%d' % len(v))
PY
cut -c1-200 $REPO/gpurun_out/r04_vcycle_pmc_raw.txt | head -70
