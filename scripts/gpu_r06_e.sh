#!/bin/bash
# Round-6 GPU session E: the whole -m gpu suite (regression after the tile / surface / MHD / multigrid changes), the cost model of
# the tile sweep's work items (per-iteration vs per-workgroup: RAMSES_AMD_TILE_ZRUN), the AMR bench legs.
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -q --timeout 900 --durations=15 ) > gpurun_out/r06_e_pytest_gpu.txt 2>&1
tail -32 gpurun_out/r06_e_pytest_gpu.txt | cut -c1-250
{
  for kind in full partial; do
    lev=8; [ $kind = partial ] && lev=9
    for z in 16 32 64 128 256; do
      echo "# kind=$kind zrun=$z"
      RAMSES_AMD_TILE_ZRUN=$z timeout 300 python scripts/amr_tile_probe.py $lev $kind 5 2>&1 | grep -v amdgpu.ids | tail -2 | cut -c1-400
    done
  done
} > gpurun_out/r06_e_zrun.txt 2>&1
cat gpurun_out/r06_e_zrun.txt
( time timeout 600 python bench.py --steps 10 --warmup 3 --vcycle-level 0 --mhd-level 0 --no-cpu-baseline ) > gpurun_out/r06_e_bench.txt 2>&1
python - <<'PY'
import json
l=[x for x in open('gpurun_out/r06_e_bench.txt') if x.startswith('{')]
if l:
    d=json.loads(l[-1])
    print('dense fast frac %.4f ms %.3f strict %.4f' % (d['roofline']['frac'], d['ms_per_step'], d['strict_build']['frac']))
    for k in ('amr_sweep','amr_sweep_partial','amr_sweep_covered'):
        a=d.get(k)
        if a: print(k, 'strict ms %.3f frac %.3f' % (a['ms_per_sweep'], a['roofline']['frac']), 'fast ms %.3f frac %.3f' % (a['fast_arithmetic']['ms_per_sweep'], a['fast_arithmetic']['frac']), 'tree ms %.3f' % a['tree_walking_ms_per_sweep'])
else:
    print(open('gpurun_out/r06_e_bench.txt').read()[-2000:])
PY
