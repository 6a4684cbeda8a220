"""The MHD drop-in end to end: the reference program built with SOLVER=mhd, NVAR=8 and
PATCH=ramses_amd/patch_mhd (oracle/_ref/ramses3d_patch_mhd_mhd: godunov_fine of a fully refined periodic level through
ramses_amd_mhd_godunov_fine_f90, everything else the reference's) against the unmodified SOLVER=mhd program
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


@pytest.mark.parametrize("level,nstep,riemann,riemann2d,slope_type", [
    (5, 12, "llf", "llf", 1), (5, 12, "hlld", "hlld", 2), (6, 10, "hlld", "hlld", 1), (5, 8, "hll", "hll", 8), (5, 8, "hlld", "llf", 7),
])
def test_patched_mhd_program_equals_the_reference(gpu_lib, monkeypatch, level, nstep, riemann, riemann2d, slope_type):
    if not (os.path.exists(REF) and os.path.exists(PATCHED)):
        pytest.skip("oracle/_ref/ramses3d_mhd / ramses3d_patch_mhd_mhd not built")
    from mhd_common import mhd_namelist
    from oracle import ramses_snapshot as rs
    nml = mhd_namelist(level, nstep, riemann, riemann2d, slope_type)
    monkeypatch.setenv("RAMSES_AMD", "1")
    work, out = rs.run_reference(nml, binary=PATCHED)
    try:
        assert "MHD godunov_fine of fully refined levels on the MI355X" in out, out[-1500:]
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
