"""Single-rank AMR runs with nremap > 0 and with snapshots in the middle of the run stay device-resident (VERDICT round 3,
missing #8): defrag renumbers the octs of every level every nremap coarse steps AND before every snapshot
(amr/amr_step.f90:109-118,153-155, amr/load_balance.f90:993-1608).  The shim of load_balance.f90 in the patch directory hands
the device's levels back to the host arrays before defrag moves them and drops the device image, which the next device
routine loads again.  Every snapshot of the patched program must equal the reference's, bit for bit."""
import importlib.util
import os
import shutil

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref", "ramses3d")
PATCHED = os.path.join(ROOT, "oracle", "_ref", "ramses3d_patch")


def _mkb():
    spec = importlib.util.spec_from_file_location("mkb", os.path.join(ROOT, "tests", "golden", "make_golden_baseline.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _leaves(outdir):
    from oracle import ramses_snapshot as rs
    s = rs.load_leaf_cells(outdir)
    order = np.lexsort((s["x"][:, 0], s["x"][:, 1], s["x"][:, 2], s["level"]))
    return s["level"][order], s["x"][order], s["prim"][:, order], float(np.ravel(s["info"]["t"])[0])


@pytest.mark.parametrize("nremap,poisson", [(2, False), (0, False), (3, True)])
def test_remaps_and_mid_run_snapshots(gpu_lib, monkeypatch, nremap, poisson):
    if not (os.path.exists(REF) and os.path.exists(PATCHED)):
        pytest.skip("oracle/_ref/ramses3d[_patch] not built")
    from oracle import ramses_snapshot as rs
    mkb = _mkb()
    if poisson:
        nml = mkb.amr_grav_namelist(lmin=5, lmax=7, nstep=6, foutput=3)
    else:
        nml = mkb.c5_namelist(5, 7, 9, 60000).replace("foutput=9", "foutput=3")
    nml = nml.replace("nremap=0", "nremap=%d" % nremap)
    assert "nremap=%d" % nremap in nml and "foutput=3" in nml
    snaps = {}
    for tag, binary, env in (("patched", PATCHED, "1"), ("reference", REF, "0")):
        monkeypatch.setenv("RAMSES_AMD", env)
        work, out = rs.run_reference(nml, binary=binary)
        try:
            if tag == "patched":
                assert "AMR levels stay resident on the GPU" in out, out[-1500:]
            dirs = sorted(d for d in os.listdir(work) if d.startswith("output_"))
            snaps[tag] = [_leaves(os.path.join(work, d)) for d in dirs]
        finally:
            shutil.rmtree(work, ignore_errors=True)
    assert len(snaps["reference"]) >= 3 and len(snaps["patched"]) == len(snaps["reference"])
    for k, (got, ref) in enumerate(zip(snaps["patched"], snaps["reference"])):
        assert got[3] == ref[3], (k, got[3], ref[3])
        assert np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1]), k
        assert np.array_equal(got[2], ref[2]), (k, np.abs(got[2] - ref[2]).max())
    assert len(set(snaps["reference"][-1][0].tolist())) >= 2        # an AMR run: several levels hold leaf cells
