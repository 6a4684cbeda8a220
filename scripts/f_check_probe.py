"""Diagnostic run: the self-gravitating AMR namelist of tests/test_mpi_amr_gravity_gpu.py on 2 ranks with RAMSES_AMD_F_CHECK=1
(the resident acceleration beside the path through the host array, cell by cell after every force_fine)."""
import importlib.util
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ramses_snapshot as rs  # noqa: E402

spec = importlib.util.spec_from_file_location("mka", os.path.join(ROOT, "tests", "golden", "make_golden_amr.py"))
m = importlib.util.module_from_spec(spec)
spec.loader.exec_module(m)
nml = m.selfgrav_namelist().replace("ngridtot=6000 !", "ngridtot=60000 !")
os.environ.update({"RAMSES_AMD": "1", "RAMSES_AMD_STATS": "1", "RAMSES_AMD_F_CHECK": "1"})
work, out = rs.run_reference(nml, binary=os.path.join(ROOT, "oracle", "_ref", "ramses3d_mpi_patch"), nproc=int(sys.argv[1]) if len(sys.argv) > 1 else 2)
shutil.rmtree(work, ignore_errors=True)
lines = [l for l in out.splitlines() if "f check" in l or "acceleration f" in l or "Fine step" in l or "Main step" in l]
print("\n".join(lines[:120]))
