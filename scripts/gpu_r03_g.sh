#!/bin/bash
# Round-3 GPU session G: CG under MPI with the halo of p on the device; V-cycle A/B of the fused smoother on the 128^3 / 64^3 levels
mkdir -p gpurun_out
R=$PWD
export TMPDIR=/tmp
( time timeout 500 python -m pytest tests/test_mpi_amr_gravity_gpu.py -m gpu -q -x --timeout 300 -k "cg_levels" ) > gpurun_out/pytest_g.txt 2>&1
tail -6 gpurun_out/pytest_g.txt | cut -c1-200
vc() { timeout 200 python bench.py --no-cpu-baseline --steps 3 --warmup 1 --amr-level 0 --stress-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); v=d.get('vcycle'); print(v['ms_per_vcycle'], v['final_error'])"; }
{
echo "== default"; vc
for fm in 128 64; do for zc in 16 32 64; do
  echo "== FUSED_MIN=$fm ZCHUNK_SMALL=$zc"; RAMSES_AMD_MG_FUSED_MIN=$fm RAMSES_AMD_MG_ZCHUNK_SMALL=$zc vc
done; done
for zc in 32 96; do echo "== ZCHUNK(256^3+)=$zc"; RAMSES_AMD_MG_ZCHUNK=$zc vc; done
} 2>&1 | tee gpurun_out/vcycle_ab3.txt
