#!/bin/bash
# Round-6 GPU session S: counters of the surface pass on the shell level (events sorted by face and device oct)
mkdir -p gpurun_out
export TMPDIR=/tmp PMC_TIMEOUT=150
R=$PWD
bash scripts/pmc_kernel.sh surf 'surface_flux' -- python $R/scripts/amr_tile_probe.py 9 partial 3 > gpurun_out/r06_s_surface_pmc.txt 2>&1
grep -E "strictmode" gpurun_out/r06_s_surface_pmc.txt | cut -c1-170
