#!/usr/bin/env python
"""Golden vectors of the AMR godunov_fine (reference run HERE, vectors committed).

Runs the UNMODIFIED reference program wrapped by oracle/dump_patch (which only
dumps the arrays around its godunov_fine) on an AMR Sedov namelist and stores,
for a few calls of godunov_fine(ilevel) on partially refined levels:
the tree (son, nbor, father, active list), uold, unew before and unew after.

    oracle/build_ref.sh ramses 3 serial oracle/dump_patch     # -> oracle/_ref/ramses3d_dump_patch
    python tests/golden/make_golden_amr.py                    # -> tests/golden/amr_godunov_ref.npz
"""
import os
import shutil
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ramses_snapshot as rs  # noqa: E402

REFINE = """&REFINE_PARAMS
interpol_var={ivar}
interpol_type={itype}
err_grad_p=0.1
/
"""
# (tag, levelmin, levelmax, nsubcycle, riemann, slope, interpol_var, interpol_type, nstep, calls to dump)
CASES = [
    ("a", 3, 5, "1,1,2,2", "llf", 1, 0, 2, 6, (8, 9, 13)),
    ("b", 3, 5, "10*1", "hllc", 2, 1, 1, 4, (5, 6, 7)),
    # analytic point-mass gravity (gravity_type=2: no Poisson solve): the predictor of ctoprim
    # with f of existing cells and the father cell's f for interpolated ones
    ("g", 3, 5, "1,1,2,2", "hll", 7, 0, 1, 4, (7, 11)),
    # pressure_fix: divu/enew updated with cmpflxm's normal-velocity and internal-energy fluxes
    ("p", 3, 5, "1,1,2,2", "hllc", 1, 0, 2, 5, (7, 15)),
]
GRAVITY = """&POISSON_PARAMS
gravity_type=2
gravity_params=3.0,0.02,0.2,0.3,0.1
/
"""


def read_dump(work, k):
    fi = os.path.join(work, "godunov_%04d_in.bin" % k)
    fo = os.path.join(work, "godunov_%04d_out.bin" % k)
    with open(fi, "rb") as fh:
        hdr = np.fromfile(fh, np.int32, 11)
        ilevel, ngrid, ngridmax, ncoarse, nvar, nvector, nlevelmax, ivar, itype, ipoisson, ipfix = [int(x) for x in hdr]
        dx, dt, gamma, smallr, smallc = np.fromfile(fh, np.float64, 5)
        igrid = np.fromfile(fh, np.int32, ngrid)
        ncell = ncoarse + 8 * ngridmax
        son = np.fromfile(fh, np.int32, ncell)
        nbor = np.fromfile(fh, np.int32, ngridmax * 6).reshape(6, ngridmax)
        father = np.fromfile(fh, np.int32, ngridmax)
        uold = np.fromfile(fh, np.float64, ncell * nvar).reshape(nvar, ncell)
        unew = np.fromfile(fh, np.float64, ncell * nvar).reshape(nvar, ncell)
        f = np.fromfile(fh, np.float64, ncell * 3).reshape(3, ncell) if ipoisson else np.zeros((0, 0))
        divu = np.fromfile(fh, np.float64, ncell) if ipfix else np.zeros(0)
        enew = np.fromfile(fh, np.float64, ncell) if ipfix else np.zeros(0)
        assert fh.read() == b""
    with open(fo, "rb") as fh:
        unew_out = np.fromfile(fh, np.float64, ncell * nvar).reshape(nvar, ncell)
        divu_out = np.fromfile(fh, np.float64, ncell) if ipfix else np.zeros(0)
        enew_out = np.fromfile(fh, np.float64, ncell) if ipfix else np.zeros(0)
        assert fh.read() == b""
    return dict(meta=np.array([ilevel, ngrid, ngridmax, ncoarse, nvar, nvector, nlevelmax, ivar, itype], np.int64),
                real=np.array([dx, dt, gamma, smallr, smallc]), igrid=igrid, son=son, nbor=nbor, father=father,
                uold=uold, unew=unew, unew_out=unew_out, f=f, divu=divu, enew=enew, divu_out=divu_out, enew_out=enew_out)


def amr_namelist(lmin, lmax, nsub, riemann, slope, ivar, itype, nstep, foutput=1, tag=""):
    grav = tag == "g"
    if tag == "p":
        riemann = riemann + "'\npressure_fix=.true.\nbeta_fix=0.5\n!'"
    nml = rs.sedov3d_namelist(level=lmin, nstepmax=nstep, foutput=foutput, riemann=riemann, slope_type=slope,
                              extra=REFINE.format(ivar=ivar, itype=itype) + (GRAVITY if grav else ""), mem_factor=1.0,
                              poisson=grav)
    nml = nml.replace("levelmax=%d" % lmin, "levelmax=%d" % lmax).replace("nsubcycle=10*1", "nsubcycle=" + nsub)
    return nml.replace("ngridtot=", "ngridtot=3000 !")


# self-gravity on AMR levels: multigrid_fine on partially refined levels (masks, Dirichlet
# boundaries interpolated from the coarser level, scan flags, the per-solve multigrid hierarchy)
SELFGRAV_INIT = """nregion=3
region_type(1)='square'
region_type(2)='point'
region_type(3)='square'
x_center=0.5,0.0,0.3
y_center=0.5,0.0,0.2
z_center=0.5,0.0,0.25
length_x=10.0,1.0,0.12
length_y=10.0,1.0,0.12
length_z=10.0,1.0,0.12
exp_region=10.0,10.0,10.0
d_region=1.0,0.0,20.0
u_region=0.0,0.0,0.0
v_region=0.0,0.0,0.0
p_region=1e-5,0.4,1e-3"""
SELFGRAV_NSTEP = 3


def selfgrav_namelist(eps="1e-5"):
    extra = """&REFINE_PARAMS
interpol_var=0
interpol_type=1
err_grad_p=0.1
err_grad_d=0.2
/
&POISSON_PARAMS
epsilon=%s
/
""" % eps
    nml = rs.sedov3d_namelist(level=3, nstepmax=SELFGRAV_NSTEP, foutput=SELFGRAV_NSTEP, riemann="llf", slope_type=1,
                              extra=extra, init=SELFGRAV_INIT, poisson=True)
    return nml.replace("levelmax=3", "levelmax=5").replace("nsubcycle=10*1", "nsubcycle=1,1,2,2").replace(
        "ngridtot=", "ngridtot=6000 !")


# reflexive walls on all six faces (nboundary=6): boundary octs are ordinary neighbours of the
# tree-walking sweep, make_boundary_hydro stays the reference's host code
WALLS = """&BOUNDARY_PARAMS
nboundary=6
ibound_min=-1,+1,-1,-1,-1,-1
ibound_max=-1,+1,+1,+1,+1,+1
jbound_min= 0, 0,-1,+1,-1,-1
jbound_max= 0, 0,-1,+1,+1,+1
kbound_min= 0, 0, 0, 0,-1,+1
kbound_max= 0, 0, 0, 0,-1,+1
bound_type= 1, 1, 1, 1, 1, 1
/
"""
WALLS_NSTEP = 12


def walls_namelist():
    nml = rs.sedov3d_namelist(level=3, nstepmax=WALLS_NSTEP, foutput=WALLS_NSTEP, riemann="hllc", slope_type=2,
                              extra=REFINE.format(ivar=0, itype=2) + WALLS)
    return nml.replace("levelmax=3", "levelmax=5").replace("ngridtot=", "ngridtot=20000 !")


def walls_selfgrav_namelist():
    """Self-gravity in a box with six reflexive walls: Dirichlet boundaries of phi enter the
    multigrid through the masks and the right-hand side on EVERY level (levelmin included)."""
    return selfgrav_namelist().replace("&POISSON_PARAMS", WALLS + "&POISSON_PARAMS").replace("ngridtot=6000 !", "ngridtot=20000 !")


# two passive scalars (NVAR=7 builds): interpolation, sweep and coarse corrections of the scalars
V7_INIT = SELFGRAV_INIT + """
var_region(1,1)=0.1
var_region(1,2)=0.2
var_region(3,1)=0.9
var_region(3,2)=0.5"""
V7_NSTEP = 5


def v7_namelist():
    nml = rs.sedov3d_namelist(level=3, nstepmax=V7_NSTEP, foutput=V7_NSTEP, riemann="hllc", slope_type=1,
                              extra=REFINE.format(ivar=1, itype=1).replace("err_grad_p=0.1", "err_grad_p=0.1\nerr_grad_d=0.2"),
                              init=V7_INIT)
    return nml.replace("levelmax=3", "levelmax=5").replace("nsubcycle=10*1", "nsubcycle=1,1,2,2").replace(
        "ngridtot=", "ngridtot=8000 !")


def difmag_namelist():
    """case "b" with artificial diffusion (cmpdivu + consup): every level, uniform or not, takes the tree-walking sweep"""
    b = [c for c in CASES if c[0] == "b"][0]
    _, lmin, lmax, nsub, riemann, slope, ivar, itype, nstep, _ = b
    return amr_namelist(lmin, lmax, nsub, riemann, slope, ivar, itype, nstep).replace("riemann='hllc'", "riemann='hllc'\ndifmag=0.1")


C5_NSTEP = 8


def c5_namelist():
    nml = rs.sedov3d_namelist(level=6, nstepmax=C5_NSTEP, foutput=C5_NSTEP, extra=REFINE.format(ivar=0, itype=2))
    return nml.replace("levelmax=6", "levelmax=8").replace("ngridtot=", "ngridtot=150000 !")


def c5_digest(snap, order):
    import hashlib
    h = hashlib.sha256()
    h.update(np.ascontiguousarray(snap["level"][order].astype(np.int8)).tobytes())
    h.update(np.ascontiguousarray(snap["x"][order]).tobytes())
    h.update(np.ascontiguousarray(snap["prim"][:, order]).tobytes())
    return h.hexdigest()


def MPI_CASES():
    a = [c for c in CASES if c[0] == "a"][0]
    _, lmin, lmax, nsub, riemann, slope, ivar, itype, nstep, _ = a
    amr = amr_namelist(lmin, lmax, nsub, riemann, slope, ivar, itype, nstep).replace("ngridtot=3000 !", "ngridtot=8000 !")
    uni = rs.sedov3d_namelist(level=4, nstepmax=4, foutput=1, riemann="hllc", slope_type=2, mem_factor=4.0)
    return [("mpi2_amr", 2, amr, nstep), ("mpi4_uniform", 4, uni, 4)]


def main():
    binary = os.path.join(ROOT, "oracle", "_ref", "ramses3d_dump_patch")
    out = {}
    for tag, lmin, lmax, nsub, riemann, slope, ivar, itype, nstep, calls in CASES:
        nml = amr_namelist(lmin, lmax, nsub, riemann, slope, ivar, itype, nstep, foutput=1000, tag=tag)
        os.environ["RAMSES_DUMP_CALLS"] = ",".join(str(c) for c in calls)
        work, log = rs.run_reference(nml, binary=binary)
        try:
            for c in calls:
                d = read_dump(work, c)
                print(tag, c, "level", d["meta"][0], "ngrid", d["meta"][1], "changed cells",
                      int((d["unew"] != d["unew_out"]).any(0).sum()))
                for k, v in d.items():
                    out["%s%d_%s" % (tag, c, k)] = v
            out[tag + "_riemann"] = np.array(riemann)
            out[tag + "_slope"] = np.array(slope)
            out[tag + "_calls"] = np.array(calls)
        finally:
            shutil.rmtree(work, ignore_errors=True)
    # end-to-end: leaf cells of the snapshots of the untouched reference program
    for tag, lmin, lmax, nsub, riemann, slope, ivar, itype, nstep, calls in CASES:
        nml = amr_namelist(lmin, lmax, nsub, riemann, slope, ivar, itype, nstep, tag=tag)
        work, log = rs.run_reference(nml)
        try:
            for k in (1, nstep + 1):
                snap = rs.load_leaf_cells(os.path.join(work, "output_%05d" % k))
                order = np.lexsort((snap["x"][:, 0], snap["x"][:, 1], snap["x"][:, 2], snap["level"]))
                out["%s_e2e%d_level" % (tag, k)] = snap["level"][order].astype(np.int8)
                out["%s_e2e%d_x" % (tag, k)] = snap["x"][order]
                out["%s_e2e%d_prim" % (tag, k)] = snap["prim"][:, order]
            print(tag, "e2e leaf cells", snap["level"].size, "levels", np.unique(snap["level"]))
        finally:
            shutil.rmtree(work, ignore_errors=True)
    # MPI: the unmodified reference on several ranks (AMR case "a" on 2 ranks, a uniform
    # level-4 run on 4 ranks); goldens = leaf cells of all ranks' snapshot files
    for tag, nproc, nml, nstep in MPI_CASES():
        work, log = rs.run_reference(nml, nproc=nproc, binary=os.path.join(ROOT, "oracle", "_ref", "ramses3d_mpi"))
        try:
            snap = rs.load_leaf_cells(os.path.join(work, "output_%05d" % (nstep + 1)))
            order = np.lexsort((snap["x"][:, 0], snap["x"][:, 1], snap["x"][:, 2], snap["level"]))
            out["%s_level" % tag] = snap["level"][order].astype(np.int8)
            out["%s_x" % tag] = snap["x"][order]
            out["%s_prim" % tag] = snap["prim"][:, order]
            print(tag, "mpi leaf cells", snap["level"].size, "levels", np.unique(snap["level"]))
        finally:
            shutil.rmtree(work, ignore_errors=True)
    # AMR with difmag
    work, log = rs.run_reference(difmag_namelist())
    try:
        nst = [c for c in CASES if c[0] == "b"][0][8]
        snap = rs.load_leaf_cells(os.path.join(work, "output_%05d" % (nst + 1)))
        order = np.lexsort((snap["x"][:, 0], snap["x"][:, 1], snap["x"][:, 2], snap["level"]))
        out["difmag_level"] = snap["level"][order].astype(np.int8)
        out["difmag_prim"] = snap["prim"][:, order]
        print("difmag leaf cells", snap["level"].size, "differs from case b:",
              not np.array_equal(out["difmag_prim"], out.get("b_e2e%d_prim" % (nst + 1))))
    finally:
        shutil.rmtree(work, ignore_errors=True)
    # AMR with two passive scalars
    work, log = rs.run_reference(v7_namelist(), binary=os.path.join(ROOT, "oracle", "_ref", "ramses3d_v7"))
    try:
        snap = rs.load_leaf_cells(os.path.join(work, "output_00002"))
        order = np.lexsort((snap["x"][:, 0], snap["x"][:, 1], snap["x"][:, 2], snap["level"]))
        out["v7_level"] = snap["level"][order].astype(np.int8)
        out["v7_prim"] = snap["prim"][:, order]
        print("v7 leaf cells", snap["level"].size, "levels", np.unique(snap["level"]), "nvar", snap["prim"].shape[0],
              "scalar range", snap["prim"][5].min(), snap["prim"][5].max())
    finally:
        shutil.rmtree(work, ignore_errors=True)
    # AMR in a box with reflexive walls
    work, log = rs.run_reference(walls_namelist())
    try:
        snap = rs.load_leaf_cells(os.path.join(work, "output_00002"))
        order = np.lexsort((snap["x"][:, 0], snap["x"][:, 1], snap["x"][:, 2], snap["level"]))
        out["walls_level"] = snap["level"][order].astype(np.int8)
        out["walls_x"] = snap["x"][order]
        out["walls_prim"] = snap["prim"][:, order]
        print("walls leaf cells", snap["level"].size, "levels", np.unique(snap["level"]),
              "x range", snap["x"].min(), snap["x"].max())
    finally:
        shutil.rmtree(work, ignore_errors=True)
    # AMR + self-gravity
    work, log = rs.run_reference(selfgrav_namelist())
    try:
        import re
        solves = re.findall(r"==> Level=\s*(\d+) Step=\s*(\d+) Error=\s*(\S+)", log)
        out["sg_solves"] = np.array([[int(a), int(b)] for a, b, _ in solves])
        snap = rs.load_leaf_cells(os.path.join(work, "output_00002"), with_grav=True)
        order = np.lexsort((snap["x"][:, 0], snap["x"][:, 1], snap["x"][:, 2], snap["level"]))
        out["sg_level"] = snap["level"][order].astype(np.int8)
        out["sg_x"] = snap["x"][order]
        out["sg_prim"] = snap["prim"][:, order]
        out["sg_grav"] = snap["grav"][:, order]
        print("selfgrav leaf cells", snap["level"].size, "levels", np.unique(snap["level"]), "solves", len(solves),
              "levels solved", sorted(set(int(a) for a, _, _ in solves)))
    finally:
        shutil.rmtree(work, ignore_errors=True)
    # AMR + self-gravity inside six walls
    work, log = rs.run_reference(walls_selfgrav_namelist())
    try:
        import re
        solves = re.findall(r"==> Level=\s*(\d+) Step=\s*(\d+) Error=\s*(\S+)", log)
        out["wg_solves"] = np.array([[int(a), int(b)] for a, b, _ in solves])
        snap = rs.load_leaf_cells(os.path.join(work, "output_00002"), with_grav=True)
        order = np.lexsort((snap["x"][:, 0], snap["x"][:, 1], snap["x"][:, 2], snap["level"]))
        out["wg_level"] = snap["level"][order].astype(np.int8)
        out["wg_x"] = snap["x"][order]
        out["wg_prim"] = snap["prim"][:, order]
        out["wg_grav"] = snap["grav"][:, order]
        print("walls+selfgrav leaf cells", snap["level"].size, "levels", np.unique(snap["level"]), "solves", len(solves))
    finally:
        shutil.rmtree(work, ignore_errors=True)
    # BASELINE config C5 at 1/8 linear size (sedov3d.nml + levelmin=6, levelmax=8 + the C5 refine
    # parameters): too many cells for a fixture, so the golden is a checksum of the sorted leaf data
    import hashlib
    work, log = rs.run_reference(c5_namelist())
    try:
        snap = rs.load_leaf_cells(os.path.join(work, "output_00002"))   # outputs at step 0 and C5_NSTEP
        order = np.lexsort((snap["x"][:, 0], snap["x"][:, 1], snap["x"][:, 2], snap["level"]))
        out["c5_sha256"] = np.array(c5_digest(snap, order))
        out["c5_ncell"] = np.array([int((snap["level"] == l).sum()) for l in (6, 7, 8)])
        print("c5 leaf cells per level", out["c5_ncell"], out["c5_sha256"])
    finally:
        shutil.rmtree(work, ignore_errors=True)
    path = os.path.join(ROOT, "tests", "golden", "amr_godunov_ref.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
