"""GPU tests of the device-resident level under MPI: the patched MPI program
(oracle/_ref/ramses3d_mpi_patch) on 2, 4 and 8 ranks keeps every rank's box of the level as a brick on the
GPU -- courant_fine, set_unew, godunov_fine (the DENSE sweep), set_uold, make_virtual_reverse_dp(unew) and
make_virtual_fine_dp(uold) through the shims of ramses_amd/patch (virtual_boundaries.f90 recognises the
array by its address) -- and must reproduce the untouched MPI reference (oracle/_ref/ramses3d_mpi, same
rank count) bit for bit.  On a box with one GPU the ranks share the device, RCCL refuses such a
communicator, and the exchange is staged through the program's own MPI on pinned host buffers (the run
says which transport it uses; the pack / unpack / plan code is the same)."""
import os
import shutil

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_MPI = os.path.join(ROOT, "oracle", "_ref", "ramses3d_mpi")
PATCHED_MPI = os.path.join(ROOT, "oracle", "_ref", "ramses3d_mpi_patch")


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


@pytest.mark.parametrize("nproc,level,riemann,slope,overlap", [(2, 5, "llf", 1, "1"), (4, 5, "hllc", 2, "1"), (8, 6, "llf", 1, "1"),
                                                               (2, 2, "llf", 1, "1"), (4, 6, "llf", 1, "0"), (8, 7, "llf", 1, "1")])
def test_mpi_resident_bricks_equal_mpi_reference(gpu_lib, nproc, level, riemann, slope, overlap):
    """overlap = 1 (default): shell sweep, then the pack / exchange of the new state on a second stream behind the
    interior sweep; 0: sweep, then exchange.  Same snapshots."""
    if not (os.path.exists(REF_MPI) and os.path.exists(PATCHED_MPI)):
        pytest.skip("oracle/_ref/ramses3d_mpi[_patch] not built")
    from oracle import ramses_snapshot as rs
    nstep = 6
    nml = rs.sedov3d_namelist(level=level, nstepmax=nstep, foutput=nstep, riemann=riemann, slope_type=slope, mem_factor=6.0)
    workp, outp = _run(nml, PATCHED_MPI, nproc, {"RAMSES_AMD": "1", "RAMSES_AMD_OVERLAP": overlap})
    try:
        assert "stays resident on the GPUs" in outp, outp[-2000:]
        assert ("halo exchange over RCCL" in outp) or ("staged through host MPI" in outp)
        got = rs.load_uniform_level(os.path.join(workp, "output_00002"), level)
    finally:
        shutil.rmtree(workp, ignore_errors=True)
    workr, outr = _run(nml, REF_MPI, nproc, {})
    try:
        ref = rs.load_uniform_level(os.path.join(workr, "output_00002"), level)
    finally:
        shutil.rmtree(workr, ignore_errors=True)
    assert got["info"]["t"] == ref["info"]["t"]
    assert np.array_equal(got["prim"], ref["prim"]), np.abs(got["prim"] - ref["prim"]).max()
    # bit patterns too (the + 0.0 of the reverse exchange turns -0.0 into +0.0 in the reference)
    assert np.array_equal(got["prim"].view(np.int64), ref["prim"].view(np.int64))


def test_mpi_resident_off_switch_takes_the_tree_walking_sweep(gpu_lib):
    """RAMSES_AMD_RESIDENT=0: the round-1 path (tree-walking sweep per rank + the reference's host MPI halo)."""
    if not (os.path.exists(REF_MPI) and os.path.exists(PATCHED_MPI)):
        pytest.skip("oracle/_ref/ramses3d_mpi[_patch] not built")
    from oracle import ramses_snapshot as rs
    nml = rs.sedov3d_namelist(level=4, nstepmax=3, foutput=3, mem_factor=6.0)
    workp, outp = _run(nml, PATCHED_MPI, 2, {"RAMSES_AMD": "1", "RAMSES_AMD_RESIDENT": "0"})
    try:
        assert "stays resident" not in outp
        got = rs.load_uniform_level(os.path.join(workp, "output_00002"), 4)
    finally:
        shutil.rmtree(workp, ignore_errors=True)
    workr, outr = _run(nml, REF_MPI, 2, {})
    try:
        ref = rs.load_uniform_level(os.path.join(workr, "output_00002"), 4)
    finally:
        shutil.rmtree(workr, ignore_errors=True)
    assert np.array_equal(got["prim"], ref["prim"])
