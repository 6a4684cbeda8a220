#!/bin/bash
# Round-5 PMC record of the sweep of a level in tiles (counter passes only): the masked dense kernel, the ghost pre-pass, the replay
mkdir -p gpurun_out
export PMC_TIMEOUT=90
R=$PWD
IFS=";" read -ra LIST <<< "${CFGS:-8 covered}"      # CFGS="8 covered;9 partial"
for cfg in "${LIST[@]}"; do
  set -- $cfg
  bash scripts/pmc_kernel.sh tile_$2 'godunov_sweep_kernel.*true|plan_ghost_fill|tile_coarse_update' -- python $R/scripts/amr_tile_probe.py $1 $2 3 > gpurun_out/r05_tile_pmc_$2.txt 2>&1
  tail -40 gpurun_out/r05_tile_pmc_$2.txt | cut -c1-170
done
