"""GPU test of multigrid_fine on a uniform level with several MPI ranks (SURVEY.md 8 rows a21,
a28, a30, e): BASELINE config C4's shape -- hydro + self-gravity on a fully refined periodic level -- run by the
patched MPI program (oracle/_ref/ramses3d_mpi_patch) on 2, 4 and 8 ranks.  The rank domains of the Hilbert
decomposition are half boxes, quarter columns, octants; multigrid_fine(levelmin) takes the distributed DENSE V-cycles
behind the C ABI (csrc/mg_dist.hip through ramses_amd/patch/multigrid_fine_commons.f90: one brick per rank inside 5 ghost
layers, one deep-halo exchange per smoother launch instead of one per colour pass, coarse levels replicated), the
messages through the program's own MPI on the library's pinned buffers (the ranks share the test box's one GPU).
V-cycle counts, phi, f and the hydro state must equal the untouched MPI reference on the same number of ranks; with
RAMSES_AMD_MG_DIST=0 the same run takes the multigrid of AMR levels (round 3's path) and must give the same."""
import importlib.util
import os
import shutil

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_MPI = os.path.join(ROOT, "oracle", "_ref", "ramses3d_mpi")
PATCHED_MPI = os.path.join(ROOT, "oracle", "_ref", "ramses3d_mpi_patch")


def _mkb():
    spec = importlib.util.spec_from_file_location("mkb", os.path.join(ROOT, "tests", "golden", "make_golden_baseline.py"))
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


_REF = {}


def _reference(nml, nproc, level):
    """the MPI reference on nproc ranks (once per rank count)"""
    from oracle import ramses_snapshot as rs
    if (level, nproc) not in _REF:
        work, out = _run(nml, REF_MPI, nproc, {"RAMSES_AMD": "0"})
        try:
            _REF[(level, nproc)] = (rs.load_uniform_level(os.path.join(work, "output_00002"), level, with_grav=True), _mkb().solves(out))
        finally:
            shutil.rmtree(work, ignore_errors=True)
    return _REF[(level, nproc)]


@pytest.mark.parametrize("level,nproc,dist", [(7, 2, "1"), (7, 4, "1"), (7, 8, "1"), (7, 2, "0"), (8, 8, "1")])
def test_uniform_self_gravity_under_mpi_equals_the_mpi_reference(gpu_lib, level, nproc, dist):
    """level 7: 128^3 (one distributed level above the replicated ones on 8 ranks); level 8 on 8 ranks: BASELINE config C4
    as stated (256^3) under MPI, 128^3 bricks, two distributed levels."""
    if not (os.path.exists(REF_MPI) and os.path.exists(PATCHED_MPI)):
        pytest.skip("oracle/_ref/ramses3d_mpi[_patch] not built")
    if nproc > (os.cpu_count() or 1):
        pytest.skip("fewer cores than ranks")
    from oracle import ramses_snapshot as rs
    mkb = _mkb()
    nml = mkb.c4_namelist(level=level).replace("ngridtot=", "ngridtot=%d !" % (3 * sum(8 ** l for l in range(level)) + 1000))
    ref, ref_solves = _reference(nml, nproc, level)
    work, out = _run(nml, PATCHED_MPI, nproc, {"RAMSES_AMD": "1", "RAMSES_AMD_MG_DIST": dist, "RAMSES_AMD_STATS": "1"})
    try:
        got = rs.load_uniform_level(os.path.join(work, "output_00002"), level, with_grav=True)
        got_solves = mkb.solves(out)
    finally:
        shutil.rmtree(work, ignore_errors=True)
    # round 6: the acceleration of a uniform self-gravitating run never crosses PCIe after the
    # first load -- force_fine files it from the brick (or from the force kernel's buffer) into the resident array, the
    # virtual octs are exchanged there; what goes down is backup_poisson's fetch for the snapshot, once per oct and output
    import re
    traffic = [[int(a), int(b)] for a, b in re.findall(r"acceleration f over PCIe:\s*(\d+) bytes to the device,\s*(\d+) bytes back", out)]
    assert len(traffic) == nproc, out[-2000:]
    octs = 8 ** (level - 1)
    up, down = sum(t[0] for t in traffic), sum(t[1] for t in traffic)
    # (own + virtual octs: the virtual ones are a shell around each rank's box, < 60 % on top at these sizes)
    assert up <= 2 * 1.6 * octs * 8 * 3 * 8, (up, octs)      # the load of the start-up (the image is built twice there), nothing per step
    assert 0 < down <= 1.6 * octs * 8 * 3 * 8 * 2, (down, octs)      # at most the two snapshots
    def poisson_traffic(text):
        """(rho to the device, phi back, rho back) summed over the ranks: the exit lines of RAMSES_AMD_STATS=1"""
        mg = [[int(a), int(b)] for a, b in re.findall(r"distributed multigrid over PCIe: rho\s*(\d+) bytes to the device, phi\s*(\d+) bytes back", text)]
        rd = [int(a) for a in re.findall(r"density deposit rho over PCIe:\s*(\d+) bytes back", text)]
        return sum(t[0] for t in mg), sum(t[1] for t in mg), sum(rd)

    if dist == "1":
        # rho and phi of this one-level run: the solves of the start-up take the host vectors (initialisation, the step that
        # builds the device image, the step after it: a brick of rho up and one of phi back each), every later one reads the
        # deposit on the device and leaves phi on the brick; backup_poisson fetches both once
        rho_up, phi_down, rho_down = poisson_traffic(out)
        cells = 8 ** level
        assert 0 < rho_up <= 3 * 8 * cells and 0 < phi_down <= 4 * 8 * cells, (rho_up, phi_down, cells)
        assert rho_down <= 3 * 1.6 * 8 * cells, (rho_down, cells)
    if (level, nproc, dist) == (7, 2, "1"):
        # ... whatever the number of steps: twice as many steps, the same bytes
        work2, out2 = _run(nml.replace("nstepmax=3", "nstepmax=6").replace("foutput=3", "foutput=6"), PATCHED_MPI, nproc,
                           {"RAMSES_AMD": "1", "RAMSES_AMD_MG_DIST": dist, "RAMSES_AMD_STATS": "1"})
        shutil.rmtree(work2, ignore_errors=True)
        assert "nstepmax=3" in nml and "Main step=      6" in out2, out2[-1500:]
        traffic2 = [[int(a), int(b)] for a, b in re.findall(r"acceleration f over PCIe:\s*(\d+) bytes to the device,\s*(\d+) bytes back", out2)]
        assert sum(t[0] for t in traffic2) == up, (traffic2, traffic)
        assert poisson_traffic(out2) == poisson_traffic(out), (poisson_traffic(out2), poisson_traffic(out))
        # and the switch gives the path before round 6 back: a brick of rho up and one of phi back per solve
        work3, out3 = _run(nml, PATCHED_MPI, nproc, {"RAMSES_AMD": "1", "RAMSES_AMD_MG_DIST": dist, "RAMSES_AMD_STATS": "1", "RAMSES_AMD_PHI_RESIDENT": "0"})
        try:
            got3 = rs.load_uniform_level(os.path.join(work3, "output_00002"), level, with_grav=True)
        finally:
            shutil.rmtree(work3, ignore_errors=True)
        assert poisson_traffic(out3)[0] >= 3 * 8 * cells
        assert np.array_equal(got3["grav"], ref["grav"]) and np.array_equal(got3["prim"], ref["prim"])
    said = "distributed over" in out
    assert said == (dist == "1"), out[-2000:]
    # round 4: the hydro state of a uniform self-gravitating level stays on the ranks' GPUs too (cell vectors + tree resident,
    # virtual boundaries, rho_fine's deposit and the gravity terms on the device) instead of being staged around every sweep
    assert "AMR levels stay resident on the GPU" in out, out[-2000:]
    if dist == "1":
        assert "one per rank (dense V-cycles" in out
    assert len(got_solves) >= 3 and got_solves == ref_solves, (got_solves, ref_solves)
    assert got["info"]["t"] == ref["info"]["t"]
    assert np.array_equal(got["grav"], ref["grav"]), np.abs(got["grav"] - ref["grav"]).max()     # phi, f
    assert np.array_equal(got["prim"], ref["prim"]), np.abs(got["prim"] - ref["prim"]).max()     # hydro state


def test_amr_run_with_a_fully_refined_levelmin_under_mpi(gpu_lib):
    """AMR + self-gravity on 2 ranks with levelmin = 7 (128^3, fully refined, two half boxes) and a refined patch at level 8:
    multigrid_fine(levelmin) takes the distributed dense V-cycles, level 8 the multigrid of AMR levels (the reference's
    driver, device operators, first guess and boundary values interpolated from the level-7 phi the dense solve left on
    the host), the hydro state stays resident on the GPUs -- leaf cells, phi, f and the V-cycle counts of every level must
    equal the MPI reference."""
    if not (os.path.exists(REF_MPI) and os.path.exists(PATCHED_MPI)):
        pytest.skip("oracle/_ref/ramses3d_mpi[_patch] not built")
    if (os.cpu_count() or 1) < 2:
        pytest.skip("fewer cores than ranks")
    import re
    from oracle import ramses_snapshot as rs
    nml = _mkb().amr_grav_namelist(lmin=7, lmax=8, nstep=2)

    def leaves(work):
        snap = rs.load_leaf_cells(os.path.join(work, "output_00002"), with_grav=True)
        order = np.lexsort((snap["x"][:, 0], snap["x"][:, 1], snap["x"][:, 2], snap["level"]))
        return snap["level"][order], snap["x"][order], snap["prim"][:, order], snap["grav"][:, order]

    work, out = _run(nml, PATCHED_MPI, 2, {"RAMSES_AMD": "1"})
    try:
        assert "distributed over" in out, out[-2000:]
        got = leaves(work)
        sol_p = re.findall(r"==> Level=\s*(\d+) Step=\s*(\d+)", out)
    finally:
        shutil.rmtree(work, ignore_errors=True)
    work, out = _run(nml, REF_MPI, 2, {"RAMSES_AMD": "0"})
    try:
        ref = leaves(work)
        sol_r = re.findall(r"==> Level=\s*(\d+) Step=\s*(\d+)", out)
    finally:
        shutil.rmtree(work, ignore_errors=True)
    assert (got[0] == 8).any() and sol_p == sol_r, (sol_p, sol_r)
    assert np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1])
    assert np.array_equal(got[3], ref[3]), np.abs(got[3] - ref[3]).max()     # phi, f
    assert np.array_equal(got[2], ref[2]), np.abs(got[2] - ref[2]).max()     # hydro state
