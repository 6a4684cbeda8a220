#!/bin/bash
# scripts/build_mhd_variant.sh TAG "-DFLAG ..." : ramses_amd/lib/ab/libramses_amd_TAG.so with mhd_sweep.hip compiled with the extra
# flags, the other objects from the regular build (for A/B runs of scripts/mhd_probe.py, RAMSES_AMD_LIB=...)
set -e
cd "$(dirname "$0")/.."
tag=$1; shift
mkdir -p ramses_amd/lib/ab ramses_amd/build/ab
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-fast-math --offload-compress -I include -ffp-contract=off $* -c ramses_amd/csrc/mhd_sweep.hip -o ramses_amd/build/ab/mhd_sweep_${tag}.o
others=$(ls ramses_amd/build/*.o | grep -v /mhd_sweep.o)
hipcc --offload-arch=gfx950 -shared -fPIC -o ramses_amd/lib/ab/libramses_amd_$tag.so ramses_amd/build/ab/mhd_sweep_${tag}.o $others -ldl
echo built ramses_amd/lib/ab/libramses_amd_$tag.so
