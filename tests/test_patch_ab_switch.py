"""RAMSES_AMD=0: the patched reference program (oracle/_ref/ramses3d_patch = the reference built with
ramses_amd/patch as its PATCH= directory) must run the untouched reference routines behind every
shim (godunov_fine, set_unew/set_uold, courant_fine, synchro_hydro_fine, rho_fine, multigrid_fine and
its compute routines, phi_fine_cg, force_fine, backup_hydro) -- the A/B switch of INTEGRATION.md.
No GPU is touched, so this runs in the CPU suite: snapshots must equal the reference's goldens."""
import importlib.util
import os
import re
import shutil
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATCHED = os.path.join(ROOT, "oracle", "_ref", "ramses3d_patch")


def _load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _run(nml):
    sys.path.insert(0, ROOT)
    from oracle import ramses_snapshot as rs
    old = os.environ.get("RAMSES_AMD")
    os.environ["RAMSES_AMD"] = "0"
    try:
        return rs, rs.run_reference(nml, binary=PATCHED)
    finally:
        if old is None:
            os.environ.pop("RAMSES_AMD", None)
        else:
            os.environ["RAMSES_AMD"] = old


@pytest.mark.skipif(not os.path.exists(PATCHED), reason="oracle/_ref/ramses3d_patch not built")
def test_uniform_self_gravity_three_steps_through_the_reference_routines():
    mkp = _load(os.path.join(ROOT, "tests", "golden", "make_golden_poisson.py"), "mkp")
    key, level, boxlen, eps, blob = mkp.CASES[-1]
    z = np.load(os.path.join(ROOT, "tests", "golden", "poisson_ref_runs.npz"))
    sys.path.insert(0, ROOT)
    from oracle import ramses_snapshot as rs0
    nml = rs0.sedov3d_namelist(level=level, nstepmax=4, foutput=3, boxlen=boxlen, poisson=True,
                               init=mkp.BLOB.format(**blob), extra="&POISSON_PARAMS\nepsilon=%s\n/\n" % eps)
    rs, (work, out) = _run(nml)
    try:
        assert "MI355X" not in out and "resident on the GPU" not in out
        snap = rs.load_uniform_level(os.path.join(work, "output_00002"), level, with_grav=True)
        gold = z[key + "_s3_grav"]
        gold = gold[1:] if gold.shape[0] == 5 else gold
        got = snap["grav"][1:] if snap["grav"].shape[0] == 5 else snap["grav"]
        assert np.array_equal(got, gold)
        assert np.array_equal(snap["prim"], z[key + "_s3_prim"])
    finally:
        shutil.rmtree(work, ignore_errors=True)


@pytest.mark.skipif(not os.path.exists(PATCHED), reason="oracle/_ref/ramses3d_patch not built")
def test_amr_self_gravity_with_cg_levels_through_the_reference_routines():
    mk = _load(os.path.join(ROOT, "tests", "golden", "make_golden_cg.py"), "mkcg")
    z = np.load(os.path.join(ROOT, "tests", "golden", "cg_ref.npz"))
    rs, (work, out) = _run(mk.cg_namelist())
    try:
        solves = re.findall(r"==> Level=\s*(\d+) Step=\s*(\d+) Error=\s*(\S+)[ \t]+(\S+)[ \t]*\n", out)
        assert np.array_equal(np.array([[int(a), int(b)] for a, b, _, _ in solves]), z["solves"])
        snap = rs.load_leaf_cells(os.path.join(work, "output_00002"), with_grav=True)
        order = np.lexsort((snap["x"][:, 0], snap["x"][:, 1], snap["x"][:, 2], snap["level"]))
        assert np.array_equal(snap["level"][order], z["level"])
        assert np.array_equal(snap["grav"][:, order], z["grav"])
        assert np.array_equal(snap["prim"][:, order], z["prim"])
    finally:
        shutil.rmtree(work, ignore_errors=True)


MPI_PATCHED = os.path.join(ROOT, "oracle", "_ref", "ramses3d_mpi_patch")


@pytest.mark.skipif(not (os.path.exists(MPI_PATCHED) and os.path.exists("/opt/conda/bin/mpiexec")),
                    reason="oracle/_ref/ramses3d_mpi_patch or mpiexec not available")
def test_mpi_build_through_the_reference_routines():
    """The MPI build of the patched program on two ranks with RAMSES_AMD=0 (AMR run): the snapshot of
    the unmodified MPI reference."""
    mka = _load(os.path.join(ROOT, "tests", "golden", "make_golden_amr.py"), "mka")
    tag, nproc, nml, nstep = [c for c in mka.MPI_CASES() if c[1] == 2][0]
    z = np.load(os.path.join(ROOT, "tests", "golden", "amr_godunov_ref.npz"))
    sys.path.insert(0, ROOT)
    from oracle import ramses_snapshot as rs
    old = os.environ.get("RAMSES_AMD")
    os.environ["RAMSES_AMD"] = "0"
    try:
        work, out = rs.run_reference(nml, nproc=nproc, binary=MPI_PATCHED)
    finally:
        if old is None:
            os.environ.pop("RAMSES_AMD", None)
        else:
            os.environ["RAMSES_AMD"] = old
    try:
        snap = rs.load_leaf_cells(os.path.join(work, "output_%05d" % (nstep + 1)))
        order = np.lexsort((snap["x"][:, 0], snap["x"][:, 1], snap["x"][:, 2], snap["level"]))
        assert np.array_equal(snap["level"][order], z[tag + "_level"])
        assert np.array_equal(snap["prim"][:, order], z[tag + "_prim"])
    finally:
        shutil.rmtree(work, ignore_errors=True)


@pytest.mark.skipif(not (os.path.exists(os.path.join(ROOT, "oracle", "_ref", "ramses3d_mpi_patch"))
                         and os.path.exists("/opt/conda/bin/mpiexec")), reason="MPI build or mpiexec missing")
def test_mpi_amr_run_with_load_balancing_through_the_reference_routines():
    """RAMSES_AMD=0 under MPI on an AMR run with nremap=1: the shims of load_balance, build_comm,
    make_virtual_fine_dp / make_virtual_reverse_dp, refine_fine, set_unew and courant_fine all take their
    reference branch -- leaf cells equal the untouched MPI reference (live, 2 ranks)."""
    t = _load(os.path.join(ROOT, "tests", "test_mpi_amr_resident_gpu.py"), "tmpi")
    nml = t._namelist(3, 5, "1,1,2,2", "llf", 1, 5, t.OFF_CENTRE, nremap=1)
    workp, outp = t._run(nml, t.PATCHED_MPI, 2, {"RAMSES_AMD": "0"})
    try:
        assert "resident on the GPU" not in outp and "Load balancing" in outp
        got = t._leaves(workp, 2)
    finally:
        shutil.rmtree(workp, ignore_errors=True)
    workr, outr = t._run(nml, t.REF_MPI, 2, {})
    try:
        ref = t._leaves(workr, 2)
    finally:
        shutil.rmtree(workr, ignore_errors=True)
    assert got[3] == ref[3] and len(set(int(l) for l in ref[0])) >= 2
    assert np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1]) and np.array_equal(got[2], ref[2])


def test_mhd_patch_with_the_switch_off_is_the_reference_program():
    """SOLVER=mhd, PATCH=ramses_amd/patch_mhd with RAMSES_AMD=0: the shim hands every level to the reference's own
    godunov_fine -- same snapshot as the unmodified SOLVER=mhd program (CPU only)."""
    import shutil
    import sys
    ref = os.path.join(ROOT, "oracle", "_ref", "ramses3d_mhd")
    pat = os.path.join(ROOT, "oracle", "_ref", "ramses3d_patch_mhd_mhd")
    if not (os.path.exists(ref) and os.path.exists(pat)):
        pytest.skip("oracle/_ref/ramses3d_mhd / ramses3d_patch_mhd_mhd not built")
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from mhd_common import mhd_namelist
    from oracle import ramses_snapshot as rs
    old = os.environ.get("RAMSES_AMD")
    os.environ["RAMSES_AMD"] = "0"
    try:
        snaps = []
        for binary in (ref, pat):
            work, out = rs.run_reference(mhd_namelist(4, 6, "hlld", "hlld", 2), binary=binary)
            try:
                snaps.append(rs.load_uniform_level(os.path.join(work, "output_00002"), 4)["prim"])
            finally:
                shutil.rmtree(work, ignore_errors=True)
    finally:
        if old is None:
            os.environ.pop("RAMSES_AMD", None)
        else:
            os.environ["RAMSES_AMD"] = old
    assert snaps[0].shape[0] == 11 and np.array_equal(snaps[0], snaps[1])


@pytest.mark.skipif(not os.path.exists(PATCHED), reason="oracle/_ref/ramses3d_patch not built")
def test_amr_run_between_walls_through_the_reference_routines():
    """RAMSES_AMD=0 behind the round-4 shims too: hydro_boundary.f90 (make_boundary_hydro) and the boundary octs in the
    level lists -- the walls golden of tests/golden/make_golden_amr.py"""
    mka = _load(os.path.join(ROOT, "tests", "golden", "make_golden_amr.py"), "mka")
    z = np.load(os.path.join(ROOT, "tests", "golden", "amr_godunov_ref.npz"))
    rs, (work, out) = _run(mka.walls_namelist())
    try:
        assert "MI355X" not in out and "resident on the GPU" not in out
        snap = rs.load_leaf_cells(os.path.join(work, "output_00002"))
        order = np.lexsort((snap["x"][:, 0], snap["x"][:, 1], snap["x"][:, 2], snap["level"]))
        assert np.array_equal(snap["level"][order], z["walls_level"])
        assert np.array_equal(snap["x"][order], z["walls_x"])
        assert np.array_equal(snap["prim"][:, order], z["walls_prim"])
    finally:
        shutil.rmtree(work, ignore_errors=True)
