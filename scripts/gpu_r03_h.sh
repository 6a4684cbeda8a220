#!/bin/bash
# Round-3 GPU session H: the distributed multigrid behind the C ABI (csrc/mg_dist.hip) on bricks that are not cubes
# (2 and 4 virtual ranks on one GPU), the dense V-cycle after the smoother's signature change, bench.py --gpus 2 / 4
# in smoke mode (gloo: ranks share the GPU, messages through the host) with its V-cycle block
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 600 python -m pytest tests/test_multigrid_parallel_gpu.py tests/test_multigrid_gpu.py tests/test_capi_symbols.py -q -x --timeout 300 ) > gpurun_out/pytest_h.txt 2>&1
tail -8 gpurun_out/pytest_h.txt | cut -c1-220
{
echo "== single GPU V-cycle (512^3)"
timeout 300 python bench.py --no-cpu-baseline --steps 3 --warmup 1 --amr-level 0 --stress-steps 0 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(json.dumps(d.get('vcycle')))"
for N in 2 4; do
  echo "== bench.py --gpus $N, gloo smoke mode, 128^3 hydro bricks, V-cycle level_local 8"
  RAMSES_AMD_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 2950$N \
    bench.py --gpus $N --steps 2 --warmup 1 --n 128 --vcycle-level 8 --spinup-ms 0 2>&1 | tail -1 | cut -c1-3000
done
} > gpurun_out/mgdist_h.txt 2>&1
cat gpurun_out/mgdist_h.txt | cut -c1-1500
