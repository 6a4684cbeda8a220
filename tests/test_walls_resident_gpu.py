"""GPU tests of AMR residency in a box with physical boundaries (missing #6): the patched program keeps
uold / unew and the tree on the GPU through a run with &BOUNDARY_PARAMS, and make_boundary_hydro (hydro/hydro_boundary.f90:5-269;
callers amr/amr_step.f90:70,293,514) fills the boundary octs on the device (csrc/capi_amr.hip:
ramses_amd_amrres_boundary_hydro) -- reflexive walls, free boundaries, the no_inflow clamp, imposed states, regions processed in the
reference's order (corners read what an earlier region wrote).  Live against the untouched reference program: leaf cells
bit for bit, on one rank and under MPI, on AMR levels and on a single uniform level."""
import importlib.util
import os
import shutil

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref", "ramses3d")
PATCHED = os.path.join(ROOT, "oracle", "_ref", "ramses3d_patch")
REF_MPI = os.path.join(ROOT, "oracle", "_ref", "ramses3d_mpi")
PATCHED_MPI = os.path.join(ROOT, "oracle", "_ref", "ramses3d_mpi_patch")

# the blast three cells off the corner x = y = z = 0: the shock meets three boundaries within a few steps, the refined
# patch leans on them (boundary octs on every level)
CORNER = """nregion=2
region_type(1)='square'
region_type(2)='point'
x_center=0.5,0.15
y_center=0.5,0.2
z_center=0.5,0.1
length_x=10.0,1.0
length_y=10.0,1.0
length_z=10.0,1.0
exp_region=10.0,10.0
d_region=1.0,0.0
u_region=0.0,0.0
v_region=0.0,0.0
p_region=1e-5,0.4"""

BOUNDS = """&BOUNDARY_PARAMS
nboundary=6
ibound_min=-1,+1,-1,-1,-1,-1
ibound_max=-1,+1,+1,+1,+1,+1
jbound_min= 0, 0,-1,+1,-1,-1
jbound_max= 0, 0,-1,+1,+1,+1
kbound_min= 0, 0, 0, 0,-1,+1
kbound_max= 0, 0, 0, 0,-1,+1
bound_type= %s
no_inflow=%s
/
"""


def _mka():
    spec = importlib.util.spec_from_file_location("mka", os.path.join(ROOT, "tests", "golden", "make_golden_amr.py"))
    mka = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mka)
    return mka


def _namelist(lmin, lmax, nsub, riemann, slope, nstep, types, no_inflow):
    from oracle import ramses_snapshot as rs
    mka = _mka()
    extra = mka.REFINE.format(ivar=0, itype=2) + BOUNDS % (types, ".true." if no_inflow else ".false.")
    nml = rs.sedov3d_namelist(level=lmin, nstepmax=nstep, foutput=nstep, riemann=riemann, slope_type=slope, extra=extra,
                              init=CORNER, mem_factor=1.0)
    nml = nml.replace("levelmax=%d" % lmin, "levelmax=%d" % lmax).replace("nsubcycle=10*1", "nsubcycle=" + nsub)
    return nml.replace("ngridtot=", "ngridtot=30000 !")


def _run(nml, binary, nproc, env):
    from oracle import ramses_snapshot as rs
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        return rs.run_reference(nml, binary=binary, nproc=nproc)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _leaves(work, k=2):
    from oracle import ramses_snapshot as rs
    snap = rs.load_leaf_cells(os.path.join(work, "output_%05d" % k))
    order = np.lexsort((snap["x"][:, 0], snap["x"][:, 1], snap["x"][:, 2], snap["level"]))
    return snap["level"][order], snap["x"][order], snap["prim"][:, order], snap["info"]["t"]


def _ab(nml, nproc, env, expect_resident=True, min_levels=2):
    ref_bin, pat_bin = (REF, PATCHED) if nproc == 1 else (REF_MPI, PATCHED_MPI)
    if not (os.path.exists(ref_bin) and os.path.exists(pat_bin)):
        pytest.skip("oracle/_ref programs not built")
    e = {"RAMSES_AMD": "1", "RAMSES_AMD_PROFILE": "1"}
    e.update(env)
    workp, outp = _run(nml, pat_bin, nproc, e)
    try:
        assert ("AMR levels stay resident on the GPU" in outp) == expect_resident, outp[-3000:]
        assert ("make_boundary_hydro (device)" in outp) == expect_resident, outp[-3000:]
        got = _leaves(workp)
    finally:
        shutil.rmtree(workp, ignore_errors=True)
    workr, outr = _run(nml, ref_bin, nproc, {})
    try:
        ref = _leaves(workr)
    finally:
        shutil.rmtree(workr, ignore_errors=True)
    assert got[3] == ref[3]
    assert len(set(int(l) for l in ref[0])) >= min_levels, "the run must have refined"
    assert np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1])
    assert np.array_equal(got[2].view(np.int64), ref[2].view(np.int64)), np.abs(got[2] - ref[2]).max()


@pytest.mark.parametrize("types,no_inflow,riemann,slope,nsub", [
    ("1, 1, 1, 1, 1, 1", False, "hllc", 2, "1,1,2,2"),
    ("2, 2, 2, 2, 2, 2", False, "llf", 1, "1,1,2,2"),
    ("2, 2, 2, 2, 2, 2", True, "hllc", 1, "10*1"),
    ("1, 2, 2, 1, 1, 2", True, "hll", 2, "1,2,2,2"),
], ids=["reflexive", "free", "free-no_inflow", "mixed-no_inflow-subcycled"])
def test_amr_run_between_boundaries_stays_resident_and_equals_the_reference(gpu_lib, types, no_inflow, riemann, slope, nsub):
    _ab(_namelist(3, 5, nsub, riemann, slope, 16, types, no_inflow), 1, {})


def test_single_uniform_level_between_walls_takes_the_resident_path(gpu_lib):
    """levelmin = levelmax with walls: no brick path (those are periodic), the level stays resident as cell vectors"""
    _ab(_namelist(4, 4, "10*1", "hllc", 2, 10, "1, 2, 1, 2, 2, 1", True), 1, {}, min_levels=1)


def test_walls_off_switch_keeps_the_staged_path(gpu_lib):
    _ab(_namelist(3, 5, "1,1,2,2", "hllc", 2, 8, "1, 2, 1, 2, 2, 1", False), 1, {"RAMSES_AMD_RESIDENT_WALLS": "0"},
        expect_resident=False)


@pytest.mark.parametrize("nproc", [2, 4])
def test_amr_run_between_boundaries_under_mpi(gpu_lib, nproc):
    _ab(_namelist(3, 5, "1,1,2,2", "hllc", 2, 10, "1, 2, 2, 1, 1, 2", True), nproc, {})


def test_imposed_boundaries_are_resident_too(gpu_lib):
    """bound_type = 3: the shim evaluates the reference's boundana for the cells of the region, the device stores the states"""
    nml = _namelist(3, 5, "1,1,2,2", "llf", 1, 12, "3, 1, 2, 3, 1, 1", False)
    nml = nml.replace("no_inflow=.false.", "no_inflow=.false.\nd_bound=1.0,0,0,2.0\nu_bound=0.1,0,0,0.0\nv_bound=0.0,0,0,-0.2\nw_bound=0.0\np_bound=1e-5,0,0,2e-5")
    _ab(nml, 1, {})


# ---- regions more than one oct deep: the reference's in-place loop decides what is read (hydro/hydro_boundary.f90:119-262) ----
IND_REF = {1: (2, 1, 4, 3, 6, 5, 8, 7), 2: (2, 1, 4, 3, 6, 5, 8, 7), 3: (3, 4, 1, 2, 7, 8, 5, 6), 4: (3, 4, 1, 2, 7, 8, 5, 6),
           5: (5, 6, 7, 8, 1, 2, 3, 4), 6: (5, 6, 7, 8, 1, 2, 3, 4),
           11: (1, 1, 3, 3, 5, 5, 7, 7), 12: (2, 2, 4, 4, 6, 6, 8, 8), 13: (1, 2, 1, 2, 5, 6, 5, 6), 14: (3, 4, 3, 4, 7, 8, 7, 8),
           15: (1, 2, 3, 4, 1, 2, 3, 4), 16: (5, 6, 7, 8, 5, 6, 7, 8)}


def _reference_loop(uold, son, nbor, ncoarse, ngridmax, btype, octs, nvector, no_inflow, smallr):
    """make_boundary_hydro's loop over one region as written: chunks of nvector octs, inside a chunk cell index by cell
    index, gather of the whole chunk before its scatter, in place (uold: [nvar, ncell], 0-based storage of 1-based indices)"""
    bdir, kind = btype % 10, btype // 10
    inbor = {1: 2, 2: 1, 3: 4, 4: 3, 5: 6, 6: 5}[bdir]
    axis = (bdir - 1) // 2
    for s in range(0, len(octs), nvector):
        chunk = octs[s:s + nvector]
        ref = [int(son[nbor[inbor - 1, g - 1] - 1]) for g in chunk]
        for ind in range(1, 9):
            cells = np.array([ncoarse + (ind - 1) * ngridmax + g for g in chunk]) - 1
            cref = np.array([ncoarse + (IND_REF[btype][ind - 1] - 1) * ngridmax + r for r in ref]) - 1
            uu = uold[:, cref].copy()
            if kind == 0:
                uu[1 + axis] = uu[1 + axis] * -1.0
            else:
                for _ in range(1):
                    d = np.maximum(uu[0], smallr)
                    ekin = np.zeros_like(d)
                    for k in range(3):
                        v = uu[1 + k] / d
                        ekin = ekin + 0.5 * d * (v * v)
                    uu[4] = uu[4] - ekin
                    if no_inflow:
                        uu[1 + axis] = np.minimum(0.0, uu[1 + axis]) if bdir % 2 == 1 else np.maximum(0.0, uu[1 + axis])
                    ekin = np.zeros_like(d)
                    for k in range(3):
                        v = uu[1 + k] / d
                        ekin = ekin + 0.5 * d * (v * v)
                    uu[4] = uu[4] + ekin
            uold[:, cells] = uu


@pytest.mark.parametrize("nvector", [1, 2, 32])
@pytest.mark.parametrize("btype", [1, 2, 3, 4, 5, 6, 11, 12, 13, 14, 15, 16])
def test_deep_boundary_regions_follow_the_references_in_place_loop(gpu_lib, btype, nvector):
    """a row of 7 octs along the wall's normal, the outer three of them boundary octs of one region, in every list order:
    what an oct reads from an oct of its own region (new or old state) must be what the reference's chunked loop reads"""
    import ctypes as C
    import itertools
    from ramses_amd import _capi
    lib = _capi.lib()
    nvar, nocts, ncoarse = 5, 7, 7
    ngridmax = nocts
    ncell = ncoarse + 8 * ngridmax
    son = np.zeros(ncell, dtype=np.int32)
    son[:ncoarse] = np.arange(1, nocts + 1)                 # coarse cell c holds oct c
    nbor = np.zeros((6, ngridmax), dtype=np.int32)
    for g in range(1, nocts + 1):
        for axis in range(3):
            nbor[2 * axis, g - 1] = g - 1                   # towards the low end of the row: coarse cell g-1 (0: none)
            nbor[2 * axis + 1, g - 1] = g + 1 if g < nocts else 0
    father = np.arange(1, nocts + 1, dtype=np.int32)
    high = (btype % 10) % 2 == 0
    region = [7, 6, 5] if high else [1, 2, 3]
    rng = np.random.default_rng(100 + btype)
    for no_inflow in (False, True):
        for order in itertools.permutations(region):
            u0 = np.empty((nvar, ncell))
            u0[0] = rng.uniform(0.5, 2.0, ncell)
            u0[1:4] = rng.normal(0.0, 1.0, (3, ncell))
            u0[4] = rng.uniform(3.0, 6.0, ncell)
            want = u0.copy()
            _reference_loop(want, son, nbor, ncoarse, ngridmax, btype, list(order), nvector, no_inflow, 1e-10)
            got = u0.copy()
            _capi.check(lib.ramses_amd_amrres_invalidate())
            _capi.check(lib.ramses_amd_amrres_load(nvar, ngridmax, ncoarse, got.ctypes.data_as(C.c_void_p), son.ctypes.data_as(C.c_void_p),
                                                   nbor.ctypes.data_as(C.c_void_p), father.ctypes.data_as(C.c_void_p)))
            bt = np.array([btype], dtype=np.int32)
            ng = np.array([len(order)], dtype=np.int32)
            ig = np.array(order, dtype=np.int32)
            _capi.check(lib.ramses_amd_amrres_boundary_hydro(1, bt.ctypes.data_as(C.c_void_p), ng.ctypes.data_as(C.c_void_p),
                                                             ig.ctypes.data_as(C.c_void_p), int(no_inflow), 1e-10, nvector, None))
            _capi.check(lib.ramses_amd_amrres_sync_all(got.ctypes.data_as(C.c_void_p)))
            _capi.check(lib.ramses_amd_amrres_invalidate())
            assert np.array_equal(got.view(np.int64), want.view(np.int64)), (btype, nvector, order, no_inflow)


def test_self_gravity_between_walls_stays_resident(gpu_lib):
    """hydro + self-gravity inside six reflexive walls (levels 3-5): the hydro state stays on the device (make_boundary_hydro
    there, f of the boundary octs mirrored after make_boundary_force), the Dirichlet solve keeps the reference's driver;
    leaf cells, phi and f equal the reference program bit for bit"""
    from oracle import ramses_snapshot as rs
    if not (os.path.exists(REF) and os.path.exists(PATCHED)):
        pytest.skip("oracle/_ref programs not built")
    nml = _mka().walls_selfgrav_namelist()
    workp, outp = _run(nml, PATCHED, 1, {"RAMSES_AMD": "1", "RAMSES_AMD_PROFILE": "1"})
    try:
        assert "AMR levels stay resident on the GPU" in outp, outp[-3000:]
        assert "make_boundary_hydro (device)" in outp, outp[-3000:]
        got = rs.load_leaf_cells(os.path.join(workp, "output_00002"))
    finally:
        shutil.rmtree(workp, ignore_errors=True)
    workr, outr = _run(nml, REF, 1, {})
    try:
        ref = rs.load_leaf_cells(os.path.join(workr, "output_00002"))
    finally:
        shutil.rmtree(workr, ignore_errors=True)
    assert np.array_equal(got["level"], ref["level"]) and np.array_equal(got["x"], ref["x"])
    assert np.array_equal(got["prim"].view(np.int64), ref["prim"].view(np.int64)), np.abs(got["prim"] - ref["prim"]).max()
    if "grav" in ref:
        assert np.array_equal(np.asarray(got["grav"]).view(np.int64), np.asarray(ref["grav"]).view(np.int64))


def test_self_gravity_between_walls_under_mpi_stays_resident(gpu_lib):
    """the same on 2 ranks: hydro state, tree and communicators resident on every rank, make_boundary_hydro on the device; the
    Dirichlet solves, rho_fine and force_fine + make_boundary_force keep the reference's MPI routines"""
    from oracle import ramses_snapshot as rs
    if not (os.path.exists(REF_MPI) and os.path.exists(PATCHED_MPI)):
        pytest.skip("oracle/_ref MPI programs not built")
    nml = _mka().walls_selfgrav_namelist().replace("ngridtot=20000 !", "ngridtot=60000 !")
    workp, outp = _run(nml, PATCHED_MPI, 2, {"RAMSES_AMD": "1", "RAMSES_AMD_PROFILE": "1"})
    try:
        assert "AMR levels stay resident on the GPU" in outp, outp[-3000:]
        assert "make_boundary_hydro (device)" in outp, outp[-3000:]
        got = rs.load_leaf_cells(os.path.join(workp, "output_00002"))
    finally:
        shutil.rmtree(workp, ignore_errors=True)
    workr, outr = _run(nml, REF_MPI, 2, {})
    try:
        ref = rs.load_leaf_cells(os.path.join(workr, "output_00002"))
    finally:
        shutil.rmtree(workr, ignore_errors=True)
    og = np.lexsort((got["x"][:, 0], got["x"][:, 1], got["x"][:, 2], got["level"]))
    orf = np.lexsort((ref["x"][:, 0], ref["x"][:, 1], ref["x"][:, 2], ref["level"]))
    assert np.array_equal(got["level"][og], ref["level"][orf]) and np.array_equal(got["x"][og], ref["x"][orf])
    assert np.array_equal(got["prim"][:, og].view(np.int64), ref["prim"][:, orf].view(np.int64))
