#!/bin/bash
# Round-5 GPU sessions C: A/B of the tile-sweep kernel variants (ramses_amd/lib/ab; TAGS="default a b", TESTS=1 runs the parity tests first)
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
[ "${TESTS:-1}" = 1 ] && timeout 300 python -m pytest tests/test_amr_tiles_gpu.py tests/test_amr_covered_gpu.py -m gpu -q 2>&1 | tail -3
for tag in ${TAGS:-default}; do
  for cfg in "8 covered" "9 partial"; do
    if [ $tag = default ]; then unset RAMSES_AMD_LIB; else export RAMSES_AMD_LIB=$R/ramses_amd/lib/ab/libramses_amd_$tag.so; fi
    echo "== $tag $cfg: $(timeout 120 python $R/scripts/amr_tile_probe.py $cfg 5 2>&1 | tail -2 | tr '\n' ' ' | cut -c1-200)"
  done
done 2>&1 | tee -a $R/gpurun_out/r05_c_ab.txt
