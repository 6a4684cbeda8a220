#!/bin/bash
# Round-6 GPU session F: L2 counters of the masked kernel; the library built with --offload-compress (17 MB instead of 110): does
# it load and pass; the fast certificate with self-gravity; the C5-shaped leg and the V-cycle CPU baseline of bench.py.
mkdir -p gpurun_out
export TMPDIR=/tmp
bash scripts/gpu_r06_tcc.sh > gpurun_out/r06_f_tcc.txt 2>&1
cat gpurun_out/r06_f_tcc.txt | cut -c1-250
( export RAMSES_AMD_LIB=$PWD/ramses_amd/lib/ab/libramses_amd_z.so
  time python -c "
import __graft_entry__ as g, time
t=time.time(); g.smoke(); print('smoke with the compressed library: %.1f s' % (time.time()-t))"
  time timeout 900 python -m pytest tests/test_godunov_gpu.py tests/test_amr_tiles_gpu.py tests/test_mhd_gpu.py -m gpu -q --timeout 600 2>&1 | tail -3 ) > gpurun_out/r06_f_compressed.txt 2>&1
cat gpurun_out/r06_f_compressed.txt | grep -v amdgpu.ids | cut -c1-200
( time timeout 1200 python -m pytest "tests/test_fast_certificate_gpu.py::test_default_mode_amr_self_gravity_live_ab" tests/test_mhd_amr_gpu.py -m gpu -q --timeout 1000 -s ) > gpurun_out/r06_f_pytest.txt 2>&1
grep -v "^$" gpurun_out/r06_f_pytest.txt | tail -20 | cut -c1-400
( time timeout 900 python bench.py --steps 10 --warmup 3 --mhd-level 0 ) > gpurun_out/r06_f_bench.txt 2>&1
python - <<'PY'
import json
l=[x for x in open('gpurun_out/r06_f_bench.txt') if x.startswith('{')]
if l:
    d=json.loads(l[-1])
    print('dense fast frac %.4f ms %.3f strict %.4f' % (d['roofline']['frac'], d['ms_per_step'], d['strict_build']['frac']))
    print('vcycle', {k: d['vcycle'].get(k) for k in ('ms_per_vcycle','value')}, d['vcycle']['roofline']['frac'], d['vcycle'].get('cpu_baseline'))
    print('cpu_baseline', d.get('cpu_baseline'))
    print('c5', json.dumps(d.get('amr_c5_shape'))[:1500])
else:
    print(open('gpurun_out/r06_f_bench.txt').read()[-3000:])
PY
