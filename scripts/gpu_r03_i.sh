#!/bin/bash
# Round-3 GPU session I: multigrid_fine of a uniform level under MPI through the distributed dense driver (Fortran shim ->
# ramses_amd_mgdist_*), 2 / 4 / 8 ranks against the MPI reference; its poisson timer next to the multigrid of AMR levels;
# bench.py --gpus 2 / 4 in smoke mode (gloo: ranks share the GPU, messages through the host) with the V-cycle block
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 600 python -m pytest tests/test_mpi_uniform_gravity_gpu.py -m gpu -q -x --timeout 300 ) > gpurun_out/pytest_i.txt 2>&1
tail -12 gpurun_out/pytest_i.txt | cut -c1-300
{
for np in 2 8; do timeout 300 python scripts/dropin_timing.py gravmpi 7 4 $np gpu 2>&1 | grep '^{' ; done
timeout 200 python scripts/dropin_timing.py gravmpi 7 4 8 ref 2>&1 | grep '^{'
} > gpurun_out/dropin_gravmpi_i.txt 2>&1
cut -c1-700 gpurun_out/dropin_gravmpi_i.txt
{
for N in 2 4; do
  echo "== bench.py --gpus $N, gloo smoke mode, 128^3 hydro bricks, V-cycle level_local 8"
  RAMSES_AMD_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 2950$N \
    bench.py --gpus $N --steps 2 --warmup 1 --cells 128 --vcycle-level 8 --spinup-ms 0 2>&1 | tail -1 | cut -c1-3500
done
} > gpurun_out/mgdist_i.txt 2>&1
cut -c1-2500 gpurun_out/mgdist_i.txt
