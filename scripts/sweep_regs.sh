#!/bin/bash
# Compile hydro_sweep.hip (fast and strict) with the given -D knobs and print VGPR use / spills of the
# flagship instantiation (LLF + minmod, 12-row tiles, no gravity):  scripts/sweep_regs.sh "-DX=1 -DY=2" ...
cd "$(dirname "$0")/../ramses_amd/csrc"
i=0
for knobs in "$@"; do
  for mode in fast strict; do
    if [ $mode = fast ]; then F="-ffp-contract=off -DRAMSES_AMD_FAST=1"; else F="-ffp-contract=off"; fi
    ( hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-fast-math -I ../../include $F $knobs -c hydro_sweep.hip \
        -o /tmp/sweepregs_${i}_$mode.o -Rpass-analysis=kernel-resource-usage 2> /tmp/sweepregs_${i}_$mode.log ) &
  done
  i=$((i+1))
done
wait
i=0
for knobs in "$@"; do
  for mode in fast strict; do
    echo "== [$knobs] $mode: $(grep -A12 'godunov_sweep_kernelILi1ELi0ELi12ELb0ELi0ELi5E' /tmp/sweepregs_${i}_$mode.log | grep -i ' VGPRs:\|VGPRs Spill\|error' | sed 's/.*remark: *//;s/\[-R.*//' | tr '\n' ' ')"
    grep -i "error" /tmp/sweepregs_${i}_$mode.log | head -3
  done
  i=$((i+1))
done
