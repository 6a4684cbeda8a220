#!/bin/bash
# Round-6 PMC record of the sweep of a level in tiles (counter passes only, one group per pass): the masked dense kernel (strict
# and fast), the surface pass, the ghost pre-pass, the replay -- on the complete level, the covered level and the SHELL level.
mkdir -p gpurun_out
export PMC_TIMEOUT=150
R=$PWD
IFS=";" read -ra LIST <<< "${CFGS:-8 full;8 covered;9 partial}"
for cfg in "${LIST[@]}"; do
  set -- $cfg
  bash scripts/pmc_kernel.sh tile_$2 'godunov_sweep_kernel.*true|surface_flux|plan_ghost_fill|tile_coarse_update' -- python $R/scripts/amr_tile_probe.py $1 $2 3 > gpurun_out/r06_tile_pmc_$2.txt 2>&1
  tail -60 gpurun_out/r06_tile_pmc_$2.txt | cut -c1-170
done
