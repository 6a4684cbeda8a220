import importlib.util, os, re, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from oracle import ramses_snapshot as rs
spec = importlib.util.spec_from_file_location("mka", os.path.join(ROOT, "tests", "golden", "make_golden_amr.py"))
mka = importlib.util.module_from_spec(spec); spec.loader.exec_module(mka)
z = np.load(os.path.join(ROOT, "tests", "golden", "amr_godunov_ref.npz"))
patched = os.path.join(ROOT, "oracle", "_ref", "ramses3d_patch")
os.environ["RAMSES_AMD"] = "1"; os.environ["RAMSES_AMD_MG_SYNC"] = "1"
names = {1: "gs_fine", 2: "res_fine", 4: "norm_fine", 8: "restrict_fine", 16: "interp_fine", 32: "gs_coarse", 64: "res_coarse", 128: "restrict_coarse", 256: "interp_coarse"}
for mask in [511] + [511 - b for b in names] + [0]:
    os.environ["RAMSES_AMD_MG_HOST"] = str(mask)
    work, out = rs.run_reference(mka.selfgrav_namelist(), binary=patched)
    solves = [(int(a), int(b), c) for a, b, c in re.findall(r"==> Level=\s*(\d+) Step=\s*(\d+) Error=\s*(\S+)", out)]
    snap = rs.load_leaf_cells(os.path.join(work, "output_00002"), with_grav=True)
    order = np.lexsort((snap["x"][:, 0], snap["x"][:, 1], snap["x"][:, 2], snap["level"]))
    ok = np.array_equal(snap["grav"][:, order], z["sg_grav"])
    on = [names[b] for b in names if not mask & b]
    print("device:", on if len(on) < 9 else "ALL", "| level-5 iters", [s[1] for s in solves if s[0] == 5][:2], [s[2] for s in solves if s[0] == 5][:1], "grav equal", ok, flush=True)
    shutil.rmtree(work, ignore_errors=True)
