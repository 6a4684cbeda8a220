#!/usr/bin/env python
"""Debug helper: an AMR case of tests/test_mpi_amr_resident_gpu.py under rocgdb (backtrace of a crash).
    python scripts/dbg_mpi_amr.py [case index 0..3]"""
import importlib.util
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("t", os.path.join(ROOT, "tests", "test_mpi_amr_resident_gpu.py"))
t = importlib.util.module_from_spec(spec)
spec.loader.exec_module(t)
CASES = [(2, 3, 5, "1,1,2,2", "llf", 1, 6, None, False), (4, 4, 6, "1,1,2,2", "hllc", 2, 5, t.OFF_CENTRE, False),
         (2, 5, 7, "10*2", "llf", 1, 4, t.OFF_CENTRE, False), (2, 3, 5, "1,1,2,2", "hllc", 1, 5, None, True)]
c = CASES[int(sys.argv[1]) if len(sys.argv) > 1 else 1]
nml = t._namelist(*c[1:])
work = tempfile.mkdtemp(prefix="dbg_")
open(os.path.join(work, "run.nml"), "w").write(nml)
binary = os.path.join(ROOT, "oracle", "_ref", "ramses3d_mpi_patch")
env = dict(os.environ, RAMSES_AMD="1")
cmd = ["/opt/conda/bin/mpiexec", "-n", str(c[0]), "/opt/rocm/bin/rocgdb", "-batch", "-ex", "run", "-ex", "bt", "--args", binary, "run.nml"]
out = subprocess.run(cmd, cwd=work, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=500)
lines = out.stdout.splitlines()
keep = [l for l in lines if l.startswith("#") or "SIG" in l or "ramses_amd" in l or "rror" in l]
print("\n".join(keep[-120:]))
print("---- tail ----")
print("\n".join(lines[-40:]))
