"""The Fortran drop-in of NDIM = 1 and NDIM = 2 builds (missing #7; BASELINE config C1 = namelist/sedov1d.nml on
one uniform level): oracle/_ref/ramses{1,2}d_patch -- the reference program with ramses_amd/patch, godunov_fine of a fully
refined level on the GPU as a brick embedded in three dimensions, the boundary octs as its ghost cells
(csrc/capi_host.hip ramses_amd_godunov_fine_lowdim_f90) -- against the UNMODIFIED reference oracle/_ref/ramses{1,2}d, live, on
the same namelist: every leaf cell of every snapshot bit for bit (verification arithmetic, tests/conftest.py)."""
import importlib.util
import os
import shutil

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _ref(name):
    return os.path.join(ROOT, "oracle", "_ref", name)


def _ab(nml, ndim, note):
    from oracle import ramses_snapshot as rs
    ref, pat = _ref("ramses%dd" % ndim), _ref("ramses%dd_patch" % ndim)
    if not (os.path.exists(ref) and os.path.exists(pat)):
        pytest.skip("oracle/_ref/ramses%dd[_patch] not built" % ndim)
    out = {}
    for tag, binary, env in (("ref", ref, "0"), ("gpu", pat, "1")):
        os.environ["RAMSES_AMD"] = env
        try:
            work, log = rs.run_reference(nml, ndim=ndim, binary=binary)
        finally:
            os.environ["RAMSES_AMD"] = "1"
        try:
            snaps = sorted(d for d in os.listdir(work) if d.startswith("output_"))
            out[tag] = ([rs.load_leaf_cells(os.path.join(work, d)) for d in snaps], log)
        finally:
            shutil.rmtree(work, ignore_errors=True)
    assert note in out["gpu"][1], out["gpu"][1][-2000:]
    # nothing silent: the library's exit line counts the sweeps per level, on the device and through the reference's host routine
    import re
    m = re.search(r"NDIM<3 godunov_fine: (\d+) sweeps on the device, (\d+) through the reference's host routine", out["gpu"][1])
    assert m, out["gpu"][1][-2000:]
    assert int(m.group(1)) > 0 and int(m.group(2)) == 0, m.group(0)
    assert "swept on the device" not in out["ref"][1]
    assert len(out["ref"][0]) == len(out["gpu"][0]) >= 2
    for a, b in zip(out["ref"][0], out["gpu"][0]):
        assert a["info"]["t"] == b["info"]["t"]
        ka = np.lexsort(tuple(a["x"][:, d] for d in range(a["x"].shape[1])))
        kb = np.lexsort(tuple(b["x"][:, d] for d in range(b["x"].shape[1])))
        assert np.array_equal(a["x"][ka], b["x"][kb]) and np.array_equal(a["level"][ka], b["level"][kb])
        assert np.array_equal(a["prim"][:, ka], b["prim"][:, kb]), np.abs(a["prim"][:, ka] - b["prim"][:, kb]).max()
    last = out["gpu"][0][-1]
    assert np.abs(last["prim"][1]).max() > 1e-3          # the blast moved


def _mk1d():
    spec = importlib.util.spec_from_file_location("mk1d", os.path.join(ROOT, "tests", "golden", "make_golden_sedov1d.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@pytest.mark.parametrize("bt,nstep,slope", [("1, 1", 40, 2), ("2, 2", 60, 2), ("1, 1", 20, 5)])
def test_c1_sedov1d_through_the_fortran_dropin(gpu_lib, bt, nstep, slope):
    """BASELINE config C1: sedov1d.nml, NDIM=1, levelmin = levelmax = 7 (128 cells), reflexive / outflow walls, a second blast
    at the right wall; slope type 5 (ultrabee) exists in the reference's NDIM=1 branch only"""
    nml = _mk1d().NML.format(level=7, nstep=nstep, foutput=10, bt=bt, slope=slope)
    _ab(nml, 1, "NDIM=1: level 7 (128 x 1 cells) is swept on the device as a brick embedded in 3-D; x between boundary octs")


@pytest.mark.parametrize("riemann,slope", [("hllc", 2), ("llf", 1)])
def test_periodic_sedov2d_through_the_fortran_dropin(gpu_lib, riemann, slope):
    from oracle import ramses_snapshot as rs
    nml = rs.sedov3d_namelist(level=6, nstepmax=6, foutput=3, riemann=riemann, slope_type=slope).replace("ngridtot=", "ngridtot=20000 !")
    _ab(nml, 2, "NDIM=2: level 6 (64 x 64 cells) is swept on the device as a brick embedded in 3-D; x periodic, y periodic")


def test_sedov2d_between_four_walls_through_the_fortran_dropin(gpu_lib):
    """namelist/sedov2d.nml's four reflexive walls (the x regions cover the corners) on one uniform level"""
    from oracle import ramses_snapshot as rs
    nml = rs.sedov3d_namelist(level=6, nstepmax=8, foutput=4, riemann="hllc", slope_type=1, boxlen=1.0).replace("ngridtot=", "ngridtot=20000 !")
    nml += """
&BOUNDARY_PARAMS
nboundary = 4
ibound_min= 0, 0,-1,+1
ibound_max= 0, 0,-1,+1
jbound_min=-1,+1,-1,-1
jbound_max=-1,+1,+1,+1
bound_type= 1, 1, 1, 1
/
"""
    _ab(nml, 2, "NDIM=2: level 6 (64 x 64 cells) is swept on the device as a brick embedded in 3-D; x between boundary octs, y between boundary octs")
