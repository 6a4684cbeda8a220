"""GPU tests of the drop-in boundary.

1. ramses_amd_godunov_fine_host on a synthetic RAMSES-layout octree (octs in
   random slot order, cell vectors (1:ncell,1:nvar)) against the oracle.
2. The REAL thing: the reference program linked with the patch directory
   ramses_amd/patch (oracle/_ref/ramses3d_patch, built by
   `oracle/build_ref.sh ramses 3 serial ramses_amd/patch`): same namelist in,
   snapshots out, compared bit-for-bit with the goldens produced by the
   untouched reference (tests/golden/sedov3d_ref_runs.npz)."""
import ctypes as C
import os
import shutil

import numpy as np
import pytest

from helpers import random_brick

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATCHED = os.path.join(ROOT, "oracle", "_ref", "ramses3d_patch")
GOLD = os.path.join(ROOT, "tests", "golden", "sedov3d_ref_runs.npz")


def _octree_layout(u, level, rng, ngridmax, ncoarse=1):
    """Scatter brick u[nvar,n,n,n] into RAMSES cell vectors with octs of the
    level placed in random slots; returns (cellvec[nvar,ncell], igrid, xg)."""
    nvar, n = u.shape[0], u.shape[1]
    no = n // 2
    ngrid = no ** 3
    ncell = ncoarse + 8 * ngridmax
    slots = rng.permutation(ngridmax)[:ngrid] + 1            # 1-based oct slots
    igrid = rng.permutation(slots).astype(np.int32)          # active list order
    xg = np.zeros((3, ngridmax))
    vec = rng.normal(size=(nvar, ncell))                     # garbage elsewhere
    # oct g of the active list sits at a random oct position
    pos = rng.permutation(ngrid)
    oz, oy, ox = np.unravel_index(pos, (no, no, no))
    for d, o in enumerate((ox, oy, oz)):
        xg[d, igrid - 1] = (2 * o + 1) / n
    for ind in range(8):
        ix, iy, iz = ind & 1, (ind >> 1) & 1, (ind >> 2) & 1
        icell = ncoarse + ind * ngridmax + (igrid - 1)
        vec[:, icell] = u[:, 2 * oz + iz, 2 * oy + iy, 2 * ox + ix]
    return vec, igrid, xg, (ox, oy, oz)


def test_host_entry_on_synthetic_octree(gpu_lib, oracle):
    import ramses_amd
    rng = np.random.default_rng(3)
    level, n = 4, 16
    u = random_brick(n, n, n, seed=9)
    dx, dt = 1.0 / n, 0.003
    ngridmax = 700
    uold, igrid, xg, (ox, oy, oz) = _octree_layout(u, level, rng, ngridmax)
    unew = uold.copy()
    other = unew.copy()
    p = ramses_amd.make_params()
    rc = gpu_lib.ramses_amd_godunov_fine_host(C.byref(p), level, len(igrid), igrid.ctypes.data_as(C.c_void_p),
                                              xg.ctypes.data_as(C.c_void_p), ngridmax, 1, 1,
                                              uold.ctypes.data_as(C.c_void_p), unew.ctypes.data_as(C.c_void_p),
                                              None, dx, dt)
    assert rc == 0, gpu_lib.ramses_amd_last_error()
    ref = oracle.godunov_uniform(oracle.make_params(), u, dx, dt)
    touched = np.zeros(unew.shape[1], bool)
    for ind in range(8):
        ix, iy, iz = ind & 1, (ind >> 1) & 1, (ind >> 2) & 1
        icell = 1 + ind * ngridmax + (igrid - 1)
        touched[icell] = True
        assert np.array_equal(unew[:, icell], ref[:, 2 * oz + iz, 2 * oy + iy, 2 * ox + ix])
    # cells of other octs / levels are untouched
    assert np.array_equal(unew[:, ~touched], other[:, ~touched])
    # a level that is not fully refined is refused loudly
    rc = gpu_lib.ramses_amd_godunov_fine_host(C.byref(p), level, len(igrid) - 1, igrid.ctypes.data_as(C.c_void_p),
                                              xg.ctypes.data_as(C.c_void_p), ngridmax, 1, 1,
                                              uold.ctypes.data_as(C.c_void_p), unew.ctypes.data_as(C.c_void_p),
                                              None, dx, dt)
    assert rc == -2


def test_resident_level_entry_points(gpu_lib, oracle):
    """courant_fine -> godunov_fine -> set_uold on the device-resident level,
    then sync to the host array: two steps against the oracle, bit for bit."""
    import ramses_amd
    rng = np.random.default_rng(5)
    level, n = 4, 16
    u = random_brick(n, n, n, seed=21)
    dx = 1.0 / n
    ngridmax = 700
    uold, igrid, xg, (ox, oy, oz) = _octree_layout(u, level, rng, ngridmax)
    before = uold.copy()
    p = ramses_amd.make_params(courant_factor=0.8)
    po = oracle.make_params()
    args = (C.byref(p), level, len(igrid), igrid.ctypes.data_as(C.c_void_p), xg.ctypes.data_as(C.c_void_p),
            ngridmax, 1, 1, uold.ctypes.data_as(C.c_void_p))
    out4 = np.zeros(4)
    uo = u.copy()
    assert gpu_lib.ramses_amd_resident_invalidate() == 0
    for step in range(2):
        rc = gpu_lib.ramses_amd_resident_courant_f90(*args, dx, 1e30, out4.ctypes.data_as(C.c_void_p))
        assert rc == 0, gpu_lib.ramses_amd_last_error()
        dto = oracle.courant_uniform(po, uo, dx, 0.8)
        assert out4[0] == dto
        vol = dx ** 3
        assert abs(out4[1] - uo[0].sum() * vol) <= 1e-13 * abs(out4[1])
        assert abs(out4[2] - uo[4].sum() * vol) <= 1e-13 * abs(out4[2])
        eint = (uo[4] - 0.5 * (uo[1] ** 2 + uo[2] ** 2 + uo[3] ** 2) / np.maximum(uo[0], 1e-10)).sum() * vol
        assert abs(out4[3] - eint) <= 1e-12 * abs(eint)
        # set_uold before godunov_fine is an error, not a silent swap
        if step == 0:
            assert gpu_lib.ramses_amd_resident_set_uold_f90(level) == -1
        rc = gpu_lib.ramses_amd_resident_godunov_f90(*args, dx, dto)
        assert rc == 0, gpu_lib.ramses_amd_last_error()
        assert gpu_lib.ramses_amd_resident_set_uold_f90(level) == 0
        uo = oracle.godunov_uniform(po, uo, dx, dto)
        # the host array is untouched until it is synced
        assert np.array_equal(uold, before)
    # forgetting a level whose host copy is stale is refused
    assert gpu_lib.ramses_amd_resident_invalidate() == -1
    assert gpu_lib.ramses_amd_resident_sync_host_f90(uold.ctypes.data_as(C.c_void_p)) == 0
    touched = np.zeros(uold.shape[1], bool)
    for ind in range(8):
        ix, iy, iz = ind & 1, (ind >> 1) & 1, (ind >> 2) & 1
        icell = 1 + ind * ngridmax + (igrid - 1)
        touched[icell] = True
        assert np.array_equal(uold[:, icell], uo[:, 2 * oz + iz, 2 * oy + iy, 2 * ox + ix])
    assert np.array_equal(uold[:, ~touched], before[:, ~touched])
    assert gpu_lib.ramses_amd_resident_invalidate() == 0


@pytest.mark.parametrize("resident", [1, 0])
@pytest.mark.parametrize("riemann,slope", [("llf", 1), ("hllc", 2), ("hll", 7), ("acoustic", 8)])
def test_patched_reference_program_reproduces_goldens(gpu_lib, riemann, slope, resident):
    """resident=1: the state stays on the GPU across courant_fine / set_unew /
    godunov_fine / set_uold and reaches the host only for the snapshots;
    resident=0: staged in and out around every sweep.  Same snapshots."""
    if not os.path.exists(PATCHED):
        pytest.skip("oracle/_ref/ramses3d_patch not built (needs the reference tree at build time)")
    from oracle import ramses_snapshot as rs
    z = np.load(GOLD)
    key = "%s_s%d_muscl" % (riemann, slope)
    nml = rs.sedov3d_namelist(level=4, nstepmax=4, foutput=1, riemann=riemann, slope_type=slope)
    env_before = {k: os.environ.get(k) for k in ("RAMSES_AMD", "RAMSES_AMD_RESIDENT")}
    os.environ["RAMSES_AMD"] = "1"
    os.environ["RAMSES_AMD_RESIDENT"] = str(resident)
    try:
        work, out = rs.run_reference(nml, binary=PATCHED)
    finally:
        for k, v in env_before.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    assert ("stays resident on the GPU" in out) == bool(resident)
    try:
        for k in range(1, 5):
            snap = rs.load_uniform_level(os.path.join(work, "output_%05d" % k), 4)
            assert np.array_equal(snap["prim"], z["%s_prim%d" % (key, k - 1)]), (key, k)
    finally:
        shutil.rmtree(work, ignore_errors=True)


@pytest.mark.parametrize("case", [0, 1, 2])
def test_patched_program_self_gravity_reproduces_goldens(gpu_lib, case):
    """hydro + poisson: the patched program's multigrid_fine (device V-cycles)
    and the reference's own force_fine on top of it give the phi and f of the
    untouched reference, bit for bit; the hydro sweep then runs with gravity."""
    if not os.path.exists(PATCHED):
        pytest.skip("oracle/_ref/ramses3d_patch not built")
    import importlib.util
    from oracle import ramses_snapshot as rs
    spec = importlib.util.spec_from_file_location("mkp", os.path.join(ROOT, "tests", "golden", "make_golden_poisson.py"))
    mkp = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mkp)
    key, level, boxlen, eps, blob = mkp.CASES[case]
    z = np.load(os.path.join(ROOT, "tests", "golden", "poisson_ref_runs.npz"))
    nml = rs.sedov3d_namelist(level=level, nstepmax=2, foutput=1, boxlen=boxlen, poisson=True,
                              init=mkp.BLOB.format(**blob), extra="&POISSON_PARAMS\nepsilon=%s\n/\n" % eps)
    os.environ["RAMSES_AMD"] = "1"
    work, out = rs.run_reference(nml, binary=PATCHED)
    try:
        import re
        m = re.search(r"==> Level=\s*(\d+) Step=\s*(\d+) Error=\s*(\S+)", out)
        assert int(m.group(2)) == int(z[key + "_meta"][3])
        snap = rs.load_uniform_level(os.path.join(work, "output_00002"), level, with_grav=True)
        g = snap["grav"]
        phi, f = (g[1], g[2:5]) if g.shape[0] == 5 else (g[0], g[1:4])
        assert np.array_equal(phi, z[key + "_phi"])
        assert np.array_equal(f, z[key + "_f"])
        # the hydro step of that coarse step ran with gravity (ctoprim predictor on the
        # device, the reference's own source-term routines on the host)
        assert np.array_equal(snap["prim"], z[key + "_prim2"])
    finally:
        shutil.rmtree(work, ignore_errors=True)


@pytest.mark.parametrize("resident_grav", ["0", "1"])
def test_patched_program_self_gravity_three_steps(gpu_lib, resident_grav):
    """Three coarse steps of hydro + self-gravity on a uniform 32^3 level through the patched
    reference program: the gravity branch of amr_step three times over (synchro_hydro_fine with the
    old and the new force, multigrid_fine, force_fine, courant_fine / godunov_fine / set_uold with
    gravity).  resident_grav = 0: arrays staged around every device call; 1
    (RAMSES_AMD_RESIDENT_GRAV=1): the level and its acceleration stay on the GPU, rho_fine gets
    the density back.  Either way the snapshot equals the untouched reference bit for bit."""
    if not os.path.exists(PATCHED):
        pytest.skip("oracle/_ref/ramses3d_patch not built")
    import importlib.util
    import re
    from oracle import ramses_snapshot as rs
    spec = importlib.util.spec_from_file_location("mkp", os.path.join(ROOT, "tests", "golden", "make_golden_poisson.py"))
    mkp = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mkp)
    key, level, boxlen, eps, blob = mkp.CASES[-1]
    z = np.load(os.path.join(ROOT, "tests", "golden", "poisson_ref_runs.npz"))
    nml = rs.sedov3d_namelist(level=level, nstepmax=4, foutput=3, boxlen=boxlen, poisson=True,
                              init=mkp.BLOB.format(**blob), extra="&POISSON_PARAMS\nepsilon=%s\n/\n" % eps)
    os.environ["RAMSES_AMD"] = "1"
    os.environ["RAMSES_AMD_RESIDENT_GRAV"] = resident_grav
    try:
        work, out = rs.run_reference(nml, binary=PATCHED)
    finally:
        os.environ.pop("RAMSES_AMD_RESIDENT_GRAV", None)
    try:
        assert ("stays resident on the GPU" in out) == (resident_grav == "1")
        iters = np.array([int(b) for _, b, _ in re.findall(r"==> Level=\s*(\d+) Step=\s*(\d+) Error=\s*(\S+)", out)])
        assert np.array_equal(iters, z[key + "_s3_iters"])
        snap = rs.load_uniform_level(os.path.join(work, "output_00002"), level, with_grav=True)
        gold = z[key + "_s3_grav"]
        gold = gold[1:] if gold.shape[0] == 5 else gold          # the golden run also wrote rho (its own -DOUTPUT_PARTICLE_DENSITY)
        got = snap["grav"][1:] if snap["grav"].shape[0] == 5 else snap["grav"]
        assert np.array_equal(got, gold), np.abs(got - gold).max()                      # phi, f(1:3)
        assert np.array_equal(snap["prim"], z[key + "_s3_prim"]), np.abs(snap["prim"] - z[key + "_s3_prim"]).max()
    finally:
        shutil.rmtree(work, ignore_errors=True)
