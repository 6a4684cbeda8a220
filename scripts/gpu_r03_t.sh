#!/bin/bash
# Round-3 GPU session T: the spin-up with the SAME kernel on a scratch level of the SAME size (bench.py default now) against none
mkdir -p gpurun_out
export TMPDIR=/tmp
{
for sp in 150 0 150; do
  echo "== --spinup-ms $sp (same kernel, same size)"
  RAMSES_AMD_BENCH_STEPS=1 timeout 120 python bench.py --no-cpu-baseline --vcycle-level 0 --amr-level 0 --stress-steps 0 --spinup-ms $sp 2> gpurun_out/steps.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel_ms'], d['roofline']['frac'], d['config']['spinup'][:60])"
  grep "ms per step" gpurun_out/steps.err | cut -c1-260
done
} > gpurun_out/spinup_ab2.txt 2>&1
cat gpurun_out/spinup_ab2.txt
