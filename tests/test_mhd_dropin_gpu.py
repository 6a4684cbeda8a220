"""The MHD drop-in end to end: the reference program built with SOLVER=mhd, NVAR=8 and
PATCH=ramses_amd/patch_mhd (oracle/_ref/ramses3d_patch_mhd_mhd: the level RESIDENT on the device -- courant_fine,
godunov_fine, set_uold there, backup_hydro fetches it -- or, with RAMSES_AMD_MHD_RESIDENT=0, godunov_fine staged through
ramses_amd_mhd_godunov_fine_f90; everything else the reference's) against the unmodified SOLVER=mhd program
(oracle/_ref/ramses3d_mhd) on the same namelist -- a blast in a magnetised medium on a uniform 32^3 / 64^3 level:
density, velocity, the six face fields and the pressure of every cell after the run, bit for bit."""
import os
import shutil
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
REF = os.path.join(ROOT, "oracle", "_ref", "ramses3d_mhd")
PATCHED = os.path.join(ROOT, "oracle", "_ref", "ramses3d_patch_mhd_mhd")


@pytest.mark.parametrize("level,nstep,riemann,riemann2d,slope_type,resident", [
    (5, 12, "llf", "llf", 1, "1"), (5, 12, "hlld", "hlld", 2, "1"), (6, 10, "hlld", "hlld", 1, "1"), (5, 8, "hll", "hll", 8, "1"),
    (5, 8, "hlld", "llf", 7, "1"), (5, 12, "hlld", "hlld", 2, "0"), (5, 8, "hll", "hll", 8, "0"),
])
def test_patched_mhd_program_equals_the_reference(gpu_lib, monkeypatch, level, nstep, riemann, riemann2d, slope_type, resident):
    if not (os.path.exists(REF) and os.path.exists(PATCHED)):
        pytest.skip("oracle/_ref/ramses3d_mhd / ramses3d_patch_mhd_mhd not built")
    from mhd_common import mhd_namelist
    from oracle import ramses_snapshot as rs
    nml = mhd_namelist(level, nstep, riemann, riemann2d, slope_type)
    monkeypatch.setenv("RAMSES_AMD", "1")
    monkeypatch.setenv("RAMSES_AMD_MHD_RESIDENT", resident)
    work, out = rs.run_reference(nml, binary=PATCHED)
    try:
        assert ("the MHD level stays resident on the GPU" in out) == (resident == "1"), out[-1500:]
        assert ("MHD godunov_fine of fully refined levels on the MI355X (staged)" in out) == (resident == "0"), out[-1500:]
        # the time steps come from the device's courant_fine: the run must take the reference's steps (checked below through t)
        got = rs.load_uniform_level(os.path.join(work, "output_00002"), level)
    finally:
        shutil.rmtree(work, ignore_errors=True)
    monkeypatch.setenv("RAMSES_AMD", "0")
    work, out = rs.run_reference(nml, binary=REF)
    try:
        ref = rs.load_uniform_level(os.path.join(work, "output_00002"), level)
    finally:
        shutil.rmtree(work, ignore_errors=True)
    assert got["prim"].shape == (11, 2 ** level, 2 ** level, 2 ** level)
    assert int(np.ravel(got["info"]["nstep"])[0]) == int(np.ravel(ref["info"]["nstep"])[0]) == nstep
    assert float(np.ravel(got["info"]["t"])[0]) == float(np.ravel(ref["info"]["t"])[0])
    assert np.abs(ref["prim"][4:7]).max() > 0.5          # a magnetised run
    assert np.array_equal(got["prim"], ref["prim"]), np.abs(got["prim"] - ref["prim"]).reshape(11, -1).max(axis=1)
