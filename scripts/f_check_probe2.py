import importlib.util, os, shutil, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ramses_snapshot as rs
spec = importlib.util.spec_from_file_location("mka", os.path.join(ROOT, "tests", "golden", "make_golden_amr.py"))
m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
nml = m.selfgrav_namelist().replace("ngridtot=6000 !", "ngridtot=60000 !")
def run(binary, env):
    for k in ("RAMSES_AMD", "RAMSES_AMD_F_RESIDENT", "RAMSES_AMD_STATS", "RAMSES_AMD_F_CHECK"): os.environ.pop(k, None)
    os.environ.update(env)
    work, out = rs.run_reference(nml, binary=os.path.join(ROOT, "oracle", "_ref", binary), nproc=2)
    print(binary, env, sorted(os.listdir(work)))
    for o in sorted(d for d in os.listdir(work) if d.startswith("output_")):
        s = rs.load_leaf_cells(os.path.join(work, o), with_grav=True)
        lv, cnt = np.unique(s["level"], return_counts=True)
        print("  ", o, dict(zip(lv.tolist(), cnt.tolist())), "sum|f|", float(np.abs(s["grav"]).sum()), "sum rho", float(s["prim"][0].sum()))
    print("\n".join(l for l in out.splitlines() if "Main step" in l or "fatal" in l.lower() or "error" in l.lower())[-1500:])
    shutil.rmtree(work, ignore_errors=True)
run("ramses3d_mpi", {})
run("ramses3d_mpi_patch", {"RAMSES_AMD": "1", "RAMSES_AMD_F_RESIDENT": "0"})
run("ramses3d_mpi_patch", {"RAMSES_AMD": "1", "RAMSES_AMD_F_RESIDENT": "1"})
run("ramses3d_mpi_patch", {"RAMSES_AMD": "1", "RAMSES_AMD_F_RESIDENT": "1", "RAMSES_AMD_F_CHECK": "1"})
