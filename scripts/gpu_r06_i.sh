#!/bin/bash
# Round-6 GPU session I: counter passes of the MHD sweep's kernels and of the fused multigrid smoother; the smoother's tile shapes
# again with the XCD-aware tile order in place
mkdir -p gpurun_out
export TMPDIR=/tmp PMC_TIMEOUT=200
R=$PWD
bash scripts/pmc_kernel.sh mhd 'mhd_prim_trace|mhd_flux|mhd_emf|mhd_update' -- python $R/scripts/mhd_probe.py 8 3 > gpurun_out/r06_mhd_pmc.txt 2>&1
tail -80 gpurun_out/r06_mhd_pmc.txt | cut -c1-170
for t in 1 4 24 16; do
  timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --mg-tune $t 2>/dev/null | python -c "import sys,json; [print('mg-tune $t:', json.dumps(json.loads(l)['vcycle'])[:500]) for l in sys.stdin if l.startswith('{')]"
done | tee gpurun_out/r06_i_mgtune.txt
bash scripts/pmc_kernel.sh vcycle 'mg_smooth_fused' -- python $R/scripts/mgtune_probe9.py > gpurun_out/r06_vcycle_pmc.txt 2>&1
tail -40 gpurun_out/r06_vcycle_pmc.txt | cut -c1-170
