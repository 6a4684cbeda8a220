"""GPU tests of AMR residency under MPI (e; BASELINE config C5's shape): the
patched MPI program (oracle/_ref/ramses3d_mpi_patch) on 2 and 4 ranks keeps uold/unew and the tree of every
rank on the GPU through an AMR run -- set_unew (incl. the zeroing of the virtual octs), the tree-walking
sweep, set_uold, upload_fine, courant_fine, hydro_flag -- and BOTH virtual-boundary exchanges of amr_step
run on the device on the reference's own cell vectors, addressed by the reference's own communicators:
    make_virtual_reverse_dp(unew(1,ivar),ilevel)   amr/virtual_boundaries.f90:693-983  (corrections owed to
                                                   cells of other ranks, accumulated peer by peer in icpu order)
    make_virtual_fine_dp(uold(1,ivar),ilevel)      amr/virtual_boundaries.f90:373-528
The run must reproduce the untouched MPI reference (oracle/_ref/ramses3d_mpi, same rank count) leaf cell by
leaf cell, bit for bit -- through regridding (refine_fine + build_comm rebuild the communicators), sub-cycling
and flux corrections across rank boundaries.  On a box with one GPU the ranks share the device and the messages
go through the program's own MPI on pinned host buffers (the run says which transport it uses)."""
import importlib.util
import os
import shutil

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_MPI = os.path.join(ROOT, "oracle", "_ref", "ramses3d_mpi")
PATCHED_MPI = os.path.join(ROOT, "oracle", "_ref", "ramses3d_mpi_patch")

# the blast off the corner: the refined patch does not sit symmetrically on the rank boundaries
OFF_CENTRE = """nregion=2
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


def _mka():
    spec = importlib.util.spec_from_file_location("mka", os.path.join(ROOT, "tests", "golden", "make_golden_amr.py"))
    mka = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mka)
    return mka


def _namelist(lmin, lmax, nsub, riemann, slope, nstep, init=None, pfix=False, nremap=0):
    from oracle import ramses_snapshot as rs
    mka = _mka()
    kw = {} if init is None else {"init": init}
    if pfix:      # pressure_fix: divu / enew travel through make_virtual_reverse_dp as well (amr/amr_step.f90:417-418)
        riemann = riemann + "'\npressure_fix=.true.\nbeta_fix=0.5\n!'"
    nml = rs.sedov3d_namelist(level=lmin, nstepmax=nstep, foutput=nstep, riemann=riemann, slope_type=slope,
                              extra=mka.REFINE.format(ivar=0, itype=2), mem_factor=1.0, **kw)
    nml = nml.replace("levelmax=%d" % lmin, "levelmax=%d" % lmax).replace("nsubcycle=10*1", "nsubcycle=" + nsub)
    assert "nremap=0" in nml
    nml = nml.replace("nremap=0", "nremap=%d" % nremap)
    return nml.replace("ngridtot=", "ngridtot=%d !" % (40000 if lmax >= 7 else 12000))


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


def _leaves(work, k):
    from oracle import ramses_snapshot as rs
    snap = rs.load_leaf_cells(os.path.join(work, "output_%05d" % k))
    order = np.lexsort((snap["x"][:, 0], snap["x"][:, 1], snap["x"][:, 2], snap["level"]))
    return snap["level"][order], snap["x"][order], snap["prim"][:, order], snap["info"]["t"]


@pytest.mark.parametrize("nproc,lmin,lmax,nsub,riemann,slope,nstep,init,pfix", [
    (2, 3, 5, "1,1,2,2", "llf", 1, 6, None, False),
    (4, 4, 6, "1,1,2,2", "hllc", 2, 5, OFF_CENTRE, False),
    (2, 5, 7, "10*2", "llf", 1, 4, OFF_CENTRE, False),
    (2, 3, 5, "1,1,2,2", "hllc", 1, 5, None, True),
], ids=["2ranks-3to5", "4ranks-4to6-offcentre", "2ranks-5to7-subcycled-offcentre", "2ranks-3to5-pressure_fix"])
def test_amr_resident_under_mpi_equals_mpi_reference(gpu_lib, nproc, lmin, lmax, nsub, riemann, slope, nstep, init, pfix):
    if not (os.path.exists(REF_MPI) and os.path.exists(PATCHED_MPI)):
        pytest.skip("oracle/_ref/ramses3d_mpi[_patch] not built")
    nml = _namelist(lmin, lmax, nsub, riemann, slope, nstep, init, pfix)
    workp, outp = _run(nml, PATCHED_MPI, nproc, {"RAMSES_AMD": "1"})
    try:
        assert "AMR levels stay resident on the GPU" in outp, outp[-3000:]
        assert ("halo exchange over RCCL" in outp) or ("staged through host MPI" in outp), outp[-3000:]
        got = _leaves(workp, 2)
    finally:
        shutil.rmtree(workp, ignore_errors=True)
    workr, outr = _run(nml, REF_MPI, nproc, {})
    try:
        ref = _leaves(workr, 2)
    finally:
        shutil.rmtree(workr, ignore_errors=True)
    assert got[3] == ref[3]
    assert len(set(int(l) for l in ref[0])) >= 2, "the run must have refined"
    assert np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1])
    assert np.array_equal(got[2], ref[2]), np.abs(got[2] - ref[2]).max()
    assert np.array_equal(got[2].view(np.int64), ref[2].view(np.int64))


def test_amr_resident_under_mpi_off_switch(gpu_lib):
    """RAMSES_AMD_RESIDENT_AMR_MPI=0: the staged path (arrays around every call, the reference's host halo)."""
    if not (os.path.exists(REF_MPI) and os.path.exists(PATCHED_MPI)):
        pytest.skip("oracle/_ref/ramses3d_mpi[_patch] not built")
    nml = _namelist(3, 5, "1,1,2,2", "llf", 1, 4)
    workp, outp = _run(nml, PATCHED_MPI, 2, {"RAMSES_AMD": "1", "RAMSES_AMD_RESIDENT_AMR_MPI": "0"})
    try:
        assert "AMR levels stay resident" not in outp
        got = _leaves(workp, 2)
    finally:
        shutil.rmtree(workp, ignore_errors=True)
    workr, outr = _run(nml, REF_MPI, 2, {})
    try:
        ref = _leaves(workr, 2)
    finally:
        shutil.rmtree(workr, ignore_errors=True)
    assert np.array_equal(got[0], ref[0]) and np.array_equal(got[2], ref[2])


@pytest.mark.parametrize("nproc,nremap", [(2, 1), (4, 2)])
def test_amr_resident_under_mpi_with_load_balancing(gpu_lib, nproc, nremap):
    """nremap > 0: every nremap coarse steps load_balance moves octs of every level between the ranks and rebuilds the
    communicators (amr/amr_step.f90:109, amr/load_balance.f90:5-280).  The shim load_balance.f90 hands the state back to
    the host first and drops the device image; the next device routine loads the re-balanced arrays.  Blast off the rank
    boundaries, so that the balance really changes; leaf cells equal the MPI reference bit for bit."""
    if not (os.path.exists(REF_MPI) and os.path.exists(PATCHED_MPI)):
        pytest.skip("oracle/_ref/ramses3d_mpi[_patch] not built")
    nml = _namelist(4, 6, "1,1,2,2", "llf", 1, 6, OFF_CENTRE, nremap=nremap)
    workp, outp = _run(nml, PATCHED_MPI, nproc, {"RAMSES_AMD": "1"})
    try:
        assert "AMR levels stay resident on the GPU" in outp, outp[-3000:]
        got = _leaves(workp, 2)
    finally:
        shutil.rmtree(workp, ignore_errors=True)
    workr, outr = _run(nml, REF_MPI, nproc, {})
    try:
        assert "Load balancing" in outr or "load balanc" in outr.lower(), outr[-2000:]
        ref = _leaves(workr, 2)
    finally:
        shutil.rmtree(workr, ignore_errors=True)
    assert got[3] == ref[3]
    assert np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1])
    assert np.array_equal(got[2], ref[2]), np.abs(got[2] - ref[2]).max()
