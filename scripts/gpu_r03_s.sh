#!/bin/bash
# Round-3 GPU session S: does the sweep still speed up during the 20 timed steps (clock ramp)?  Per-step kernel times with the
# default 150 ms of spin-up, with 600 ms and with 1500 ms, on one box.
mkdir -p gpurun_out
export TMPDIR=/tmp
{
for sp in 150 600 1500 150; do
  echo "== --spinup-ms $sp"
  RAMSES_AMD_BENCH_STEPS=1 timeout 120 python bench.py --no-cpu-baseline --vcycle-level 0 --amr-level 0 --stress-steps 0 --spinup-ms $sp 2> gpurun_out/steps.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel_ms'], d['roofline']['frac'], d['config']['spinup'][:40])"
  grep "ms per step" gpurun_out/steps.err | cut -c1-260
done
} > gpurun_out/spinup_ab.txt 2>&1
cat gpurun_out/spinup_ab.txt
