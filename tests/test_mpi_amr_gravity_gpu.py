"""GPU test of the AMR multigrid under MPI (round 2, item 3): the patched MPI program
(oracle/_ref/ramses3d_mpi_patch) on 2 and 4 ranks runs hydro + self-gravity on AMR levels 3-5.  Every level
(levelmin included) takes the reference's own multigrid driver (the MPI_ALLREDUCE of the norms stays with it) and
every compute routine -- Gauss-Seidel fine / coarse with the masked branch, residual, norm, restriction,
prolongation -- runs on the rank's GPU, over the rank's own octs followed by the reception octs of its neighbours
(csrc/capi_tree_poisson.hip: ramses_amd_mgamr_level_begin / _level_block / _fine_active).
  default (round 3)          the levels of the solve STAY on the device between the routines; make_virtual_fine_dp on phi and
                             the residual, make_virtual_mg_dp and make_reverse_mg_dp exchange the device arrays
                             (ramses_amd_mgamr_halo_*): no level array crosses PCIe after the upload -- asserted on the
                             transfer counters the library prints with RAMSES_AMD_MG_STATS=1
  (round 2's path -- the arrays across PCIe around every routine, the reference's host exchanges in between -- and its switch
  RAMSES_AMD_MG_MPI_SYNC were retired in round 6.)
phi, f, the hydro state of every leaf cell and the V-cycle counts must equal the untouched MPI reference
(oracle/_ref/ramses3d_mpi, same rank count) bit for bit."""
import importlib.util
import os
import re
import shutil

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_MPI = os.path.join(ROOT, "oracle", "_ref", "ramses3d_mpi")
PATCHED_MPI = os.path.join(ROOT, "oracle", "_ref", "ramses3d_mpi_patch")


def _mka():
    spec = importlib.util.spec_from_file_location("mka", os.path.join(ROOT, "tests", "golden", "make_golden_amr.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


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


def _sorted(snap):
    order = np.lexsort((snap["x"][:, 0], snap["x"][:, 1], snap["x"][:, 2], snap["level"]))
    return snap["level"][order], snap["x"][order], snap["prim"][:, order], snap["grav"][:, order]


@pytest.mark.parametrize("nproc,resident,mgsync", [(2, "1", "0"), (4, "1", "0"), (8, "1", "0"), (2, "0", "0")])
def test_amr_multigrid_under_mpi_equals_the_mpi_reference(gpu_lib, nproc, resident, mgsync):
    """resident=1 (default): the hydro state and the tree stay on every rank's GPU as well (virtual-boundary
    exchanges of uold / unew on the device, the acceleration mirrored incl. the virtual octs, density back for
    rho_fine); resident=0 (RAMSES_AMD_RESIDENT_AMR_MPI=0): hydro arrays staged around every call.
    The multigrid levels stay on the device during a solve (mgsync is always "0" since round 6)."""
    if not (os.path.exists(REF_MPI) and os.path.exists(PATCHED_MPI)):
        pytest.skip("oracle/_ref/ramses3d_mpi[_patch] not built")
    if nproc > (os.cpu_count() or 1):
        pytest.skip("fewer cores than ranks")
    from oracle import ramses_snapshot as rs
    nml = _mka().selfgrav_namelist().replace("ngridtot=6000 !", "ngridtot=60000 !")
    env = {"RAMSES_AMD": "1", "RAMSES_AMD_RESIDENT_AMR_MPI": resident, "RAMSES_AMD_MG_STATS": "1"}
    workp, outp = _run(nml, PATCHED_MPI, nproc, env)
    try:
        assert ("AMR levels stay resident on the GPU" in outp) == (resident == "1"), outp[-1500:]
        stats = [[int(v) for v in m] for m in re.findall(
            r"multigrid level\s+\d+: level arrays across PCIe after the upload:\s*(\d+) bytes in\s*(\d+) copies; halo:\s*(\d+) bytes in\s*(\d+) exchanges", outp)]
        assert len(stats) >= nproc          # every rank reports every solve
        if mgsync == "1":
            assert "multigrid under MPI: compute routines on the GPUs" in outp, outp[-1500:]      # (list-directed output wraps the line)
            assert all(s[0] > 0 and s[3] == 0 for s in stats)
        else:
            assert "multigrid under MPI: levels resident on the GPUs" in outp, outp[-1500:]
            assert all(s[0] == 0 and s[1] == 0 for s in stats), stats[:4]      # no level array crossed PCIe after the upload
            assert sum(s[3] for s in stats) > 100 and sum(s[2] for s in stats) > 0      # the exchanges ran on the device arrays
        sol_p = re.findall(r"==> Level=\s*(\d+) Step=\s*(\d+)", outp)
        got = _sorted(rs.load_leaf_cells(os.path.join(workp, "output_00002"), with_grav=True))
    finally:
        shutil.rmtree(workp, ignore_errors=True)
    workr, outr = _run(nml, REF_MPI, nproc, {})
    try:
        sol_r = re.findall(r"==> Level=\s*(\d+) Step=\s*(\d+)", outr)
        ref = _sorted(rs.load_leaf_cells(os.path.join(workr, "output_00002"), with_grav=True))
    finally:
        shutil.rmtree(workr, ignore_errors=True)
    assert sol_p == sol_r
    assert np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1])
    assert np.array_equal(got[3], ref[3]), np.abs(got[3] - ref[3]).max()     # phi, f
    assert np.array_equal(got[2], ref[2]), np.abs(got[2] - ref[2]).max()     # hydro state


def test_force_fine_under_mpi_leaves_the_acceleration_on_the_device(gpu_lib):
    """Round 6 (the f half): force_fine of the AMR levels of a resident run with several ranks files f
    of the rank's own cells into the resident acceleration on the device and exchanges the virtual octs there
    (ramses_amd_poisamr_force_mpi_resident + ramses_amd_amrres_halo_* direction 7); the host array sees f again only in
    backup_poisson.  Counters of the library's exit line: what still goes up is the first load and the levels a regrid
    rebuilt; RAMSES_AMD_F_RESIDENT=0 (f back to the host, the reference's three exchanges, the whole level up again) moves
    several times that.  Both equal the MPI reference bit for bit."""
    if not (os.path.exists(REF_MPI) and os.path.exists(PATCHED_MPI)):
        pytest.skip("oracle/_ref/ramses3d_mpi[_patch] not built")
    from oracle import ramses_snapshot as rs
    nproc = 2
    nml = _mka().selfgrav_namelist().replace("ngridtot=6000 !", "ngridtot=60000 !")
    pat = r"acceleration f over PCIe:\s*(\d+) bytes to the device,\s*(\d+) bytes back"
    res = {}
    for mode in ("1", "0"):
        work, out = _run(nml, PATCHED_MPI, nproc, {"RAMSES_AMD": "1", "RAMSES_AMD_STATS": "1", "RAMSES_AMD_F_RESIDENT": mode})
        try:
            assert "AMR levels stay resident on the GPU" in out, out[-1500:]
            traffic = [[int(a), int(b)] for a, b in re.findall(pat, out)]
            assert len(traffic) == nproc, out[-2000:]
            res[mode] = (traffic, _sorted(rs.load_leaf_cells(os.path.join(work, "output_00002"), with_grav=True)))
        finally:
            shutil.rmtree(work, ignore_errors=True)
    workr, _ = _run(nml, REF_MPI, nproc, {})
    try:
        ref = _sorted(rs.load_leaf_cells(os.path.join(workr, "output_00002"), with_grav=True))
    finally:
        shutil.rmtree(workr, ignore_errors=True)
    for mode in ("1", "0"):
        got = res[mode][1]
        assert np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1])
        assert np.array_equal(got[3], ref[3]), (mode, np.abs(got[3] - ref[3]).max())     # phi, f
        assert np.array_equal(got[2], ref[2]), (mode, np.abs(got[2] - ref[2]).max())     # hydro state
    up_new, up_old = sum(t[0] for t in res["1"][0]), sum(t[0] for t in res["0"][0])
    down_old = sum(t[1] for t in res["0"][0])
    assert down_old == 0          # (the old path's way back is poisamr_force_mpi's own copy, not counted here)
    assert 0 < up_new < 0.5 * up_old, (up_new, up_old)
    assert sum(t[1] for t in res["1"][0]) > 0         # backup_poisson fetched f for the snapshot


@pytest.mark.parametrize("nproc,ordered", [(2, "default"), (4, "default"), (8, "default"), (2, "0")])
def test_cg_levels_under_mpi_equal_the_mpi_reference(gpu_lib, nproc, ordered):
    """phi_fine_cg under MPI: cg_levelmin=4, so levels 4 and 5 of the self-gravitating AMR run
    are solved by the conjugate-gradient loop -- every loop body on the rank's GPU (ramses_amd_cgmpi_*), the two
    MPI_ALLREDUCEs per iteration (poisson/phi_fine_cg.f90:108,154) and the halo of p (:134) in the shim.  Ordered
    local sums: iteration counts, phi, f and the hydro state equal the MPI reference bit for bit; parallel sums:
    equal to rounding with the same iteration counts (+-1)."""
    if not (os.path.exists(REF_MPI) and os.path.exists(PATCHED_MPI)):
        pytest.skip("oracle/_ref/ramses3d_mpi[_patch] not built")
    from oracle import ramses_snapshot as rs
    spec = importlib.util.spec_from_file_location("mkcg", os.path.join(ROOT, "tests", "golden", "make_golden_cg.py"))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    nml = mk.cg_namelist().replace("ngridtot=6000 !", "ngridtot=60000 !")
    pat = r"==> Level=\s*(\d+) Step=\s*(\d+)"
    # default: dot products in the reference's order (parity scan), the halo of p exchanged from the device vector;
    # "0": parallel-tree sums (equal to rounding)
    if nproc > (os.cpu_count() or 1):
        pytest.skip("fewer cores than ranks")
    env = {"RAMSES_AMD": "1"}
    if ordered == "0":
        env["RAMSES_AMD_CG_ORDERED"] = "0"
    workp, outp = _run(nml, PATCHED_MPI, nproc, env)
    try:
        assert "Entering phi_fine_cg" not in outp or "MI355X" in outp
        sol_p = np.array([[int(a), int(b)] for a, b in re.findall(pat, outp)])
        got = _sorted(rs.load_leaf_cells(os.path.join(workp, "output_00002"), with_grav=True))
    finally:
        shutil.rmtree(workp, ignore_errors=True)
    workr, outr = _run(nml, REF_MPI, nproc, {})
    try:
        sol_r = np.array([[int(a), int(b)] for a, b in re.findall(pat, outr)])
        ref = _sorted(rs.load_leaf_cells(os.path.join(workr, "output_00002"), with_grav=True))
    finally:
        shutil.rmtree(workr, ignore_errors=True)
    assert len(sol_r) > 0 and {4, 5} <= set(int(l) for l in sol_r[:, 0])
    assert np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1])
    if ordered != "0":
        assert np.array_equal(sol_p, sol_r)
        assert np.array_equal(got[3], ref[3]), np.abs(got[3] - ref[3]).max()     # phi, f
        assert np.array_equal(got[2], ref[2]), np.abs(got[2] - ref[2]).max()     # hydro state
    else:
        assert sol_p.shape == sol_r.shape and np.abs(sol_p - sol_r).max() <= 1
        assert np.abs(got[3] - ref[3]).max() <= 1e-9 * np.abs(ref[3]).max()
        assert np.abs(got[2] - ref[2]).max() <= 1e-11 * np.abs(ref[2]).max()
