"""make_golden_baseline.py -- TEST INFRASTRUCTURE ONLY.

Checksums of the UNMODIFIED reference program (oracle/_ref/ramses3d, serial) on the
BASELINE.json configurations at sizes too large for per-cell fixtures:

  c4_128   config C4 stand-in: hydro + self-gravity on a uniform 128^3
           periodic level, rho = 1 background + 'square' over-density rho = 10 of side 0.25,
           P = 1, epsilon = 1d-6, three coarse steps; digest of (prim, phi, f) and the
           V-cycle counts of every solve
  c4_256   the same at BASELINE's stated size for config C4, uniform 256^3 (levelmin = levelmax = 8)
  c5_79    config C5 at levels 7-9 (sedov3d.nml, levelmin=7 levelmax=9, interpol_var=0,
           interpol_type=2, err_grad_p=0.1; 8 coarse steps); digest of the sorted leaf data

The serial reference takes minutes on these, so the goldens are made here (this container,
CPU) and the GPU tests compare the patched program with them; the uniform hydro-only
configurations C2 (128^3, 256^3) are cheap with the MPI reference and are A/B-ed live on
the GPU box instead (tests/test_baseline_sizes_gpu.py).

  amr_grav_68  AMR levels 6-8 with self-gravity (blob + blast), 3 coarse steps: sha256 of the sorted leaf data
           (level, x, prim, phi, f), cells per level, V-cycle counts
Run:  python tests/golden/make_golden_baseline.py [c4_128] [c4_256] [c5_79] [amr_grav_68]
Writes tests/golden/baseline_sizes.json (merged with what is there).
"""
import hashlib
import json
import os
import re
import shutil
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ramses_snapshot as rs  # noqa: E402

PATH = os.path.join(ROOT, "tests", "golden", "baseline_sizes.json")

C4_INIT = """nregion=2
region_type(1)='square'
region_type(2)='square'
x_center=0.5,0.5
y_center=0.5,0.5
z_center=0.5,0.5
length_x=10.0,0.25
length_y=10.0,0.25
length_z=10.0,0.25
exp_region=10.0,10.0
d_region=1.0,10.0
u_region=0.0,0.0
v_region=0.0,0.0
p_region=1.0,1.0"""

REFINE = """&REFINE_PARAMS
interpol_var=0
interpol_type=2
err_grad_p=0.1
/
"""


def c4_namelist(level=7, nstep=3):
    """hydro + self-gravity, uniform 2^level cells per direction, boxlen 1, outputs at steps 0 and nstep"""
    return rs.sedov3d_namelist(level=level, nstepmax=nstep, foutput=nstep, boxlen=1.0, poisson=True, init=C4_INIT,
                               extra="&POISSON_PARAMS\nepsilon=1d-6\n/\n")


def c5_namelist(lmin=7, lmax=9, nstep=8, ngridtot=900000):
    nml = rs.sedov3d_namelist(level=lmin, nstepmax=nstep, foutput=nstep, extra=REFINE)
    return nml.replace("levelmax=%d" % lmin, "levelmax=%d" % lmax).replace("ngridtot=", "ngridtot=%d !" % ngridtot)


def amr_grav_namelist(lmin=6, lmax=8, nstep=3, foutput=None):
    """AMR + self-gravity: the blob + blast setup of tests/golden/make_golden_amr.py (levels 3-5) moved to
    levels lmin..lmax -- every level takes multigrid_fine, force_fine, rho_fine, synchro_hydro_fine and the
    gravity terms of courant_fine / godunov_fine / set_uold, with regridding every coarse step"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("mka", os.path.join(os.path.dirname(os.path.abspath(__file__)), "make_golden_amr.py"))
    mka = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mka)
    nml = mka.selfgrav_namelist().replace("levelmin=3", "levelmin=%d" % lmin).replace("levelmax=5", "levelmax=%d" % lmax)
    nml = nml.replace("nstepmax=%d" % mka.SELFGRAV_NSTEP, "nstepmax=%d" % nstep).replace("ngridtot=6000 !", "ngridtot=600000 !")
    return nml.replace("foutput=%d" % mka.SELFGRAV_NSTEP, "foutput=%d" % (nstep if foutput is None else foutput))


def digest_leaves_grav(snap):
    order = np.lexsort((snap["x"][:, 0], snap["x"][:, 1], snap["x"][:, 2], snap["level"]))
    h = hashlib.sha256()
    h.update(np.ascontiguousarray(snap["level"][order].astype(np.int8)).tobytes())
    h.update(np.ascontiguousarray(snap["x"][order]).tobytes())
    h.update(np.ascontiguousarray(snap["prim"][:, order]).tobytes())
    g = snap["grav"]
    g = g[1:] if g.shape[0] == 5 else g
    h.update(np.ascontiguousarray(g[:, order]).tobytes())
    return h.hexdigest()


def digest_uniform(snap):
    h = hashlib.sha256()
    h.update(np.ascontiguousarray(snap["prim"]).tobytes())
    if snap.get("grav") is not None:
        g = snap["grav"]
        g = g[1:] if g.shape[0] == 5 else g          # a -DOUTPUT_PARTICLE_DENSITY build also writes rho
        h.update(np.ascontiguousarray(g).tobytes())
    return h.hexdigest()


def digest_leaves(snap):
    order = np.lexsort((snap["x"][:, 0], snap["x"][:, 1], snap["x"][:, 2], snap["level"]))
    h = hashlib.sha256()
    h.update(np.ascontiguousarray(snap["level"][order].astype(np.int8)).tobytes())
    h.update(np.ascontiguousarray(snap["x"][order]).tobytes())
    h.update(np.ascontiguousarray(snap["prim"][:, order]).tobytes())
    return h.hexdigest()


def solves(log):
    return [[int(a), int(b)] for a, b, _ in re.findall(r"==> Level=\s*(\d+) Step=\s*(\d+) Error=\s*(\S+)", log)]


def main():
    todo = sys.argv[1:] or ["c4_128", "c5_79", "amr_grav_68"]
    out = json.load(open(PATH)) if os.path.exists(PATH) else {}
    if "c4_128" in todo:
        work, log = rs.run_reference(c4_namelist())
        try:
            snap = rs.load_uniform_level(os.path.join(work, "output_00002"), 7, with_grav=True)
            out["c4_128"] = dict(sha256=digest_uniform(snap), solves=solves(log), rho_tot=snap["info"]["rho_tot"],
                                 t=snap["info"]["t"])
            print("c4_128", out["c4_128"])
        finally:
            shutil.rmtree(work, ignore_errors=True)
    if "c4_256" in todo:
        work, log = rs.run_reference(c4_namelist(level=8))
        try:
            snap = rs.load_uniform_level(os.path.join(work, "output_00002"), 8, with_grav=True)
            out["c4_256"] = dict(sha256=digest_uniform(snap), solves=solves(log), rho_tot=snap["info"]["rho_tot"],
                                 t=snap["info"]["t"])
            print("c4_256", out["c4_256"])
        finally:
            shutil.rmtree(work, ignore_errors=True)
    if "c5_79" in todo:
        work, log = rs.run_reference(c5_namelist())
        try:
            snap = rs.load_leaf_cells(os.path.join(work, "output_00002"))
            out["c5_79"] = dict(sha256=digest_leaves(snap), ncell=[int((snap["level"] == l).sum()) for l in (7, 8, 9)],
                                t=snap["info"]["t"])
            print("c5_79", out["c5_79"])
        finally:
            shutil.rmtree(work, ignore_errors=True)
    if "amr_grav_68" in todo:
        work, log = rs.run_reference(amr_grav_namelist())
        try:
            snap = rs.load_leaf_cells(os.path.join(work, "output_00002"), with_grav=True)
            out["amr_grav_68"] = dict(sha256=digest_leaves_grav(snap), ncell=[int((snap["level"] == l).sum()) for l in (6, 7, 8)],
                                      solves=solves(log), t=snap["info"]["t"])
            print("amr_grav_68", out["amr_grav_68"])
        finally:
            shutil.rmtree(work, ignore_errors=True)
    with open(PATH, "w") as fh:
        json.dump(out, fh, indent=1, sort_keys=True)
    print("wrote", PATH)


if __name__ == "__main__":
    main()
