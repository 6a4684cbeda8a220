"""GPU tests: a run with ONE uniform level that no brick path carries -- pressure_fix, difmag, several ranks whose domains
are not boxes, nremap > 0 under MPI -- stays device-resident through the AMR path (cell vectors + tree on the GPU, the
tree-walking sweep with its PFIX / DIFMAG branches; ramses_amd_iface: ramses_amd_amr_config) instead of being staged around
every call.  Live against the untouched reference program, leaf cells bit for bit."""
import importlib.util
import os
import shutil

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref", "ramses3d")
PATCHED = os.path.join(ROOT, "oracle", "_ref", "ramses3d_patch")
REF_MPI = os.path.join(ROOT, "oracle", "_ref", "ramses3d_mpi")
PATCHED_MPI = os.path.join(ROOT, "oracle", "_ref", "ramses3d_mpi_patch")


def _mka():
    spec = importlib.util.spec_from_file_location("mka", os.path.join(ROOT, "tests", "golden", "make_golden_amr.py"))
    mka = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mka)
    return mka


def _namelist(level, nstep, riemann, slope, opts, poisson=False, nremap=0):
    from oracle import ramses_snapshot as rs
    mka = _mka()
    r = riemann + "'\n" + opts + "\n!'" if opts else riemann
    kw = {"init": mka.SELFGRAV_INIT} if poisson else {}
    extra = "&POISSON_PARAMS\nepsilon=1e-5\n/\n" if poisson else ""
    nml = rs.sedov3d_namelist(level=level, nstepmax=nstep, foutput=nstep, riemann=r, slope_type=slope, extra=extra,
                              poisson=poisson, mem_factor=3.0, **kw)
    return nml.replace("nremap=0", "nremap=%d" % nremap)


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


def _leaves(work, k=2):
    from oracle import ramses_snapshot as rs
    snap = rs.load_leaf_cells(os.path.join(work, "output_%05d" % k))
    order = np.lexsort((snap["x"][:, 0], snap["x"][:, 1], snap["x"][:, 2], snap["level"]))
    return snap["level"][order], snap["x"][order], snap["prim"][:, order], snap["info"]["t"]


def _ab(nml, nproc, env, resident_message=True):
    ref_bin, pat_bin = (REF, PATCHED) if nproc == 1 else (REF_MPI, PATCHED_MPI)
    if not (os.path.exists(ref_bin) and os.path.exists(pat_bin)):
        pytest.skip("oracle/_ref programs not built")
    e = {"RAMSES_AMD": "1"}
    e.update(env)
    workp, outp = _run(nml, pat_bin, nproc, e)
    try:
        assert ("AMR levels stay resident on the GPU" in outp) == resident_message, outp[-3000:]
        got = _leaves(workp)
    finally:
        shutil.rmtree(workp, ignore_errors=True)
    workr, outr = _run(nml, ref_bin, nproc, {})
    try:
        ref = _leaves(workr)
    finally:
        shutil.rmtree(workr, ignore_errors=True)
    assert got[3] == ref[3]
    assert np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1])
    assert np.array_equal(got[2].view(np.int64), ref[2].view(np.int64)), np.abs(got[2] - ref[2]).max()


@pytest.mark.parametrize("opts,riemann,slope", [
    ("pressure_fix=.true.\nbeta_fix=0.5", "hllc", 1),
    ("difmag=0.1", "llf", 2),
    ("pressure_fix=.true.\nbeta_fix=0.5\ndifmag=0.05", "hll", 1),
], ids=["pressure_fix", "difmag", "both"])
def test_uniform_level_with_options_of_the_tree_walking_sweep_is_resident(gpu_lib, opts, riemann, slope):
    _ab(_namelist(5, 10, riemann, slope, opts), 1, {})


def test_resident_switch_off_keeps_the_staging_path(gpu_lib):
    _ab(_namelist(5, 6, "hllc", 1, "pressure_fix=.true.\nbeta_fix=0.5"), 1, {"RAMSES_AMD_RESIDENT": "0"}, resident_message=False)


def test_uniform_level_with_pressure_fix_and_self_gravity_on_one_rank(gpu_lib):
    _ab(_namelist(5, 4, "llf", 1, "pressure_fix=.true.\nbeta_fix=0.5", poisson=True), 1, {})


@pytest.mark.parametrize("nproc,opts,nremap", [(2, "pressure_fix=.true.\nbeta_fix=0.5", 0), (3, "", 0), (2, "", 2)],
                         ids=["2ranks-pressure_fix", "3ranks-domains-are-not-boxes", "2ranks-nremap"])
def test_uniform_level_under_mpi_without_a_brick_path_is_resident(gpu_lib, nproc, opts, nremap):
    _ab(_namelist(5, 8, "hllc", 1, opts, nremap=nremap), nproc, {})
