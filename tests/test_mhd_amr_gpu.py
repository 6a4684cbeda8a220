"""SOLVER=mhd on an AMR tree (second half): godfine1 of partly refined levels on the device
(csrc/mhd_amr.hip through ramses_amd/patch_mhd/godunov_fine.f90) -- the 6^3 stencil gathered through the tree, missing neighbour
octs interpolated divergence free from the coarser level (mhd/interpol_hydro.f90:612-793, 990-1473), mag_unsplit on the stencil,
fluxes and EMFs reset next to refined cells, constrained transport, and the flux / EMF corrections of the coarser level in the
reference's order (mhd/godunov_fine.f90:538-1459).

Live A/B: the SOLVER=mhd reference program (oracle/_ref/ramses3d_mhd) against the patched one (ramses3d_patch_mhd_mhd) on the same
namelist -- the magnetised blast of tests/mhd_common.py on levels 5-7 with gradient refinement, sub-cycling and a regrid every
coarse step: every leaf cell (level, position, density, velocity, the six face fields, pressure) and the time, bit for bit; the
divergence of B of every leaf cell at rounding; the patched program's exit line must say that no level went to the reference's
host routine."""
import os
import re
import shutil
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
REF = os.path.join(ROOT, "oracle", "_ref", "ramses3d_mhd")
PATCHED = os.path.join(ROOT, "oracle", "_ref", "ramses3d_patch_mhd_mhd")


def amr_namelist(lmin, lmax, nstep, riemann, riemann2d, slope_type, interpol_var, interpol_type, poisson=False, nsub="1,2,2,2,2"):
    from mhd_common import mhd_namelist
    nml = mhd_namelist(lmin, nstep, riemann, riemann2d, slope_type)
    if slope_type == 3:
        # the reference has no 27-point slope for the face fields ("Unknown slope_mag_type", mhd/umuscl.f90): name one
        nml = nml.replace("slope_type=3", "slope_type=3\nslope_mag_type=2")
    nml = nml.replace("levelmax=%d" % lmin, "levelmax=%d" % lmax)
    nml = re.sub(r"ngridtot=\d+", "ngridtot=300000", nml)
    nml = nml.replace("nsubcycle=10*1", "nsubcycle=%s" % nsub)
    nml += "&REFINE_PARAMS\ninterpol_var=%d\ninterpol_type=%d\nerr_grad_d=0.08\nerr_grad_p=0.15\nerr_grad_b2=0.2\n/\n" % (interpol_var, interpol_type)
    if poisson:
        nml = nml.replace("hydro=.true.", "hydro=.true.\npoisson=.true.")
        nml += "&POISSON_PARAMS\nepsilon=1d-5\ngravity_type=0\n/\n"
    return nml


def _sorted(snap):
    order = np.lexsort((snap["x"][:, 0], snap["x"][:, 1], snap["x"][:, 2], snap["level"]))
    return snap["level"][order], snap["x"][order], snap["prim"][:, order]


@pytest.mark.parametrize("lmin,lmax,nstep,riemann,riemann2d,slope_type,ivar,itype,poisson", [
    (5, 7, 6, "llf", "llf", 1, 0, 1, False),
    (5, 7, 6, "hlld", "hlld", 2, 1, 2, False),
    (4, 6, 8, "hll", "hlla", 1, 0, 3, False),
    (4, 6, 6, "hlld", "hlld", 3, 1, 0, False),
    (4, 6, 5, "hlld", "hlld", 1, 0, 1, True),
])
def test_patched_mhd_program_on_an_amr_tree_equals_the_reference(gpu_lib, monkeypatch, lmin, lmax, nstep, riemann, riemann2d, slope_type,
                                                                 ivar, itype, poisson):
    if not (os.path.exists(REF) and os.path.exists(PATCHED)):
        pytest.skip("oracle/_ref/ramses3d_mhd / ramses3d_patch_mhd_mhd not built")
    from oracle import ramses_snapshot as rs
    nml = amr_namelist(lmin, lmax, nstep, riemann, riemann2d, slope_type, ivar, itype, poisson)
    monkeypatch.setenv("RAMSES_AMD", "1")
    work, out = rs.run_reference(nml, binary=PATCHED)
    try:
        assert "MHD godunov_fine of AMR levels on the MI355X (staged)" in out, out[-2500:]
        m = re.search(r"MHD godunov_fine of AMR levels: (\d+) sweeps on the device \((\d+) octs\), (\d+) through the reference's host routine", out)
        assert m, out[-2500:]
        dev, octs, host = (int(x) for x in m.groups())
        assert dev > nstep and host == 0, (dev, octs, host)
        got = rs.load_leaf_cells(os.path.join(work, "output_00002"))
    finally:
        shutil.rmtree(work, ignore_errors=True)
    monkeypatch.setenv("RAMSES_AMD", "0")
    work, out = rs.run_reference(nml, binary=REF)
    try:
        ref = rs.load_leaf_cells(os.path.join(work, "output_00002"))
    finally:
        shutil.rmtree(work, ignore_errors=True)
    lg, xg, pg = _sorted(got)
    lr, xr, pr = _sorted(ref)
    counts = [int((lr == l).sum()) for l in range(lmin, lmax + 1)]
    print("MHD AMR %d-%d %s/%s slope %d interpol %d/%d%s: %d device sweeps (%d octs), leaves per level %s"
          % (lmin, lmax, riemann, riemann2d, slope_type, ivar, itype, " +gravity" if poisson else "", dev, octs, counts))
    assert min(counts) > 100, counts                      # every level is populated: coarse-fine boundaries on both sides
    assert pr.shape[0] == 11
    # (info's nstep counts the fine steps of the sub-cycling: nstep coarse steps are at least as many)
    assert int(np.ravel(got["info"]["nstep"])[0]) == int(np.ravel(ref["info"]["nstep"])[0]) >= nstep
    assert float(np.ravel(got["info"]["t"])[0]) == float(np.ravel(ref["info"]["t"])[0])
    assert np.array_equal(lg, lr) and np.array_equal(xg, xr), "the two runs refined different cells"
    assert np.abs(pr[4:7]).max() > 0.5                    # a magnetised run
    bad = np.nonzero((pg != pr).any(axis=0))[0]
    assert bad.size == 0, (bad.size, lg[bad[:5]], xg[bad[:5]], (pg - pr)[:, bad[:5]])
    # div B of every leaf cell from its own six face fields: (Bx_r - Bx_l + By_r - By_l + Bz_r - Bz_l) / dx at rounding
    dxl = 0.5 ** lg
    div = ((pg[7] - pg[4]) + (pg[8] - pg[5]) + (pg[9] - pg[6])) / dxl
    assert np.abs(div).max() < 1e-10 * np.abs(pg[4:10]).max() / dxl.min(), np.abs(div).max()


REF_MPI = os.path.join(ROOT, "oracle", "_ref", "ramses3d_mpi_mhd")
PATCHED_MPI = os.path.join(ROOT, "oracle", "_ref", "ramses3d_mpi_patch_mhd_mhd")


@pytest.mark.parametrize("nproc,lmin,lmax,nstep,riemann,riemann2d,slope_type,ivar,itype", [
    (2, 5, 7, 5, "hlld", "hlld", 1, 0, 1),
    (4, 4, 6, 6, "hll", "hll", 2, 1, 2),
])
def test_patched_mhd_program_on_an_amr_tree_under_mpi_equals_the_mpi_reference(gpu_lib, monkeypatch, nproc, lmin, lmax, nstep, riemann, riemann2d,
                                                                               slope_type, ivar, itype):
    """Several ranks (they share this box's one GPU): every rank sweeps its own active octs on the device; the virtual octs of its
    neighbours are ordinary octs of its tree and what the sweep owes to coarse cells of other ranks goes home through the reference's
    own make_virtual_reverse_dp on unew -- the patched MPI program of SOLVER=mhd against the MPI reference on the same ranks, leaf cell
    by leaf cell, bit for bit."""
    if not (os.path.exists(REF_MPI) and os.path.exists(PATCHED_MPI)):
        pytest.skip("oracle/_ref/ramses3d_mpi_mhd / ramses3d_mpi_patch_mhd_mhd not built")
    if nproc > (os.cpu_count() or 1):
        pytest.skip("fewer cores than ranks")
    from oracle import ramses_snapshot as rs
    nml = amr_namelist(lmin, lmax, nstep, riemann, riemann2d, slope_type, ivar, itype)
    monkeypatch.setenv("RAMSES_AMD", "1")
    work, out = rs.run_reference(nml, binary=PATCHED_MPI, nproc=nproc)
    try:
        assert "MHD godunov_fine of AMR levels on the MI355X (staged)" in out, out[-2500:]
        ms = re.findall(r"MHD godunov_fine of AMR levels: (\d+) sweeps on the device \((\d+) octs\), (\d+) through the reference's host routine", out)
        assert len(ms) == nproc, out[-2500:]                      # every rank reports
        assert all(int(m[0]) > nstep and int(m[2]) == 0 for m in ms), ms
        got = rs.load_leaf_cells(os.path.join(work, "output_00002"))
    finally:
        shutil.rmtree(work, ignore_errors=True)
    monkeypatch.setenv("RAMSES_AMD", "0")
    work, out = rs.run_reference(nml, binary=REF_MPI, nproc=nproc)
    try:
        ref = rs.load_leaf_cells(os.path.join(work, "output_00002"))
    finally:
        shutil.rmtree(work, ignore_errors=True)
    lg, xg, pg = _sorted(got)
    lr, xr, pr = _sorted(ref)
    counts = [int((lr == l).sum()) for l in range(lmin, lmax + 1)]
    print("MHD AMR %d-%d on %d ranks, %s/%s: device sweeps per rank %s, leaves per level %s" % (lmin, lmax, nproc, riemann, riemann2d, [int(m[0]) for m in ms], counts))
    assert min(counts) > 100, counts
    assert float(np.ravel(got["info"]["t"])[0]) == float(np.ravel(ref["info"]["t"])[0])
    assert np.array_equal(lg, lr) and np.array_equal(xg, xr), "the two runs refined different cells"
    bad = np.nonzero((pg != pr).any(axis=0))[0]
    assert bad.size == 0, (bad.size, lg[bad[:5]], xg[bad[:5]], (pg - pr)[:, bad[:5]])


WALLS = """&BOUNDARY_PARAMS
nboundary=6
ibound_min=-1,+1,-1,-1,-1,-1
ibound_max=-1,+1,+1,+1,+1,+1
jbound_min= 0, 0,-1,+1,-1,-1
jbound_max= 0, 0,-1,+1,+1,+1
kbound_min= 0, 0, 0, 0,-1,+1
kbound_max= 0, 0, 0, 0,-1,+1
bound_type= 1, 1, 1, 2, 2, 2
/
"""


def test_patched_mhd_program_on_an_amr_tree_between_walls_equals_the_reference(gpu_lib, monkeypatch):
    """&BOUNDARY_PARAMS (three reflexive, three outflow faces) on the AMR run: the boundary octs are ordinary octs of the tree,
    filled by the reference's make_boundary_hydro (mhd/hydro_boundary.f90) on the host before every sweep; the box's coarse grid is
    3 x 3 x 3 cells.  Patched program == reference, leaf cell by leaf cell."""
    if not (os.path.exists(REF) and os.path.exists(PATCHED)):
        pytest.skip("oracle/_ref/ramses3d_mhd / ramses3d_patch_mhd_mhd not built")
    from oracle import ramses_snapshot as rs
    lmin, lmax, nstep = 4, 6, 8
    nml = amr_namelist(lmin, lmax, nstep, "hlld", "hlld", 1, 0, 1) + WALLS
    monkeypatch.setenv("RAMSES_AMD", "1")
    work, out = rs.run_reference(nml, binary=PATCHED)
    try:
        m = re.search(r"MHD godunov_fine of AMR levels: (\d+) sweeps on the device \((\d+) octs\), (\d+) through the reference's host routine", out)
        assert m, out[-2500:]
        assert int(m.group(1)) > nstep and int(m.group(3)) == 0, m.group(0)
        got = rs.load_leaf_cells(os.path.join(work, "output_00002"))
    finally:
        shutil.rmtree(work, ignore_errors=True)
    monkeypatch.setenv("RAMSES_AMD", "0")
    work, out = rs.run_reference(nml, binary=REF)
    try:
        ref = rs.load_leaf_cells(os.path.join(work, "output_00002"))
    finally:
        shutil.rmtree(work, ignore_errors=True)
    lg, xg, pg = _sorted(got)
    lr, xr, pr = _sorted(ref)
    counts = [int((lr == l).sum()) for l in range(lmin, lmax + 1)]
    print("MHD AMR %d-%d between walls: %s device sweeps, leaves per level %s" % (lmin, lmax, m.group(1), counts))
    assert min(counts) > 50, counts
    assert float(np.ravel(got["info"]["t"])[0]) == float(np.ravel(ref["info"]["t"])[0])
    assert np.array_equal(lg, lr) and np.array_equal(xg, xr), "the two runs refined different cells"
    bad = np.nonzero((pg != pr).any(axis=0))[0]
    assert bad.size == 0, (bad.size, lg[bad[:5]], xg[bad[:5]], (pg - pr)[:, bad[:5]])
