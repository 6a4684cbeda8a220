#!/bin/bash
# Round-5 GPU session F: where the dense sweep on tiles overtakes the tree-walking sweep (fully refined levels of 64^3 .. 256^3)
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export RAMSES_AMD_TILE_MIN_OCTS=0
for cfg in "6 full" "7 full" "7 covered" "8 partial" "8 full"; do
  echo "== $cfg: $(timeout 120 python $R/scripts/amr_tile_probe.py $cfg 8 2>&1 | tail -2 | tr '\n' ' ' | cut -c1-230)"
done 2>&1 | tee $R/gpurun_out/r05_f_crossover.txt
