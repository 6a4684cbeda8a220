#!/bin/bash
# scripts/build_variant.sh TAG "-DFLAG ..." : ramses_amd/lib/ab/libramses_amd_TAG.so with the LLF + minmod instantiations of the
# sweep only (a ten-second compile; for kernel tuning with scripts/amr_tile_probe.py, RAMSES_AMD_LIB=...), the other objects
# from the regular build
set -e
cd "$(dirname "$0")/.."
tag=$1; shift
mkdir -p ramses_amd/lib/ab ramses_amd/build/ab
C="hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-fast-math --offload-compress -I include -ffp-contract=off -DSWEEP_FLAGSHIP_ONLY $*"
$C -c ramses_amd/csrc/hydro_sweep.hip -o ramses_amd/build/ab/sweep_${tag}_strict.o &
$C -DRAMSES_AMD_FAST=1 -c ramses_amd/csrc/hydro_sweep.hip -o ramses_amd/build/ab/sweep_${tag}_fast.o &
wait
others=$(ls ramses_amd/build/*.o | grep -v hydro_sweep_)
hipcc --offload-arch=gfx950 -shared -fPIC -o ramses_amd/lib/ab/libramses_amd_$tag.so ramses_amd/build/ab/sweep_${tag}_strict.o ramses_amd/build/ab/sweep_${tag}_fast.o $others -ldl
echo built ramses_amd/lib/ab/libramses_amd_$tag.so
