"""The reference's OWN golden vectors (tests/hydro/*/...-ref.dat) against the
reference program built by oracle/build_ref.sh: pins oracle/_ref -- the anchor of
the whole parity chain -- to what the reference's test-suite expects.

The checker restates tests/visu/visu_ramses.py::check_solution (:495-667):
per-variable fsum(log10|x|) for density/pressure, fsum(|x|) otherwise, compared
with relative tolerance 3e-13.  Needs the reference tree (build container only)."""
import math
import os
import shutil

import numpy as np
import pytest

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _read_ref_dat(path):
    out = {}
    for line in open(path):
        k, v = line.split(":")
        out[k.strip()] = float(v)
    return out


def _checksums(data, threshold=2.0e-14, min_variance=1.0e-14):
    sol = {}
    for key, arr in data.items():
        av = np.average(arr)
        kd = arr if av == 0.0 else np.where(np.abs(arr - av) / abs(av) < min_variance, av, arr)
        if key in ("density", "pressure"):
            s = np.log10(np.abs(kd))
        else:
            s = np.where(np.abs(kd) < threshold * 1.0, 0.0, np.abs(kd))
        sol[key] = math.fsum(s)
    return sol


# (case, NDIM, tolerance) -- the cases of tests/hydro that touch the hot path
# without patches: sod-tube (godfine1 + trace1d + HLLC + moncen, AMR 3-10,
# interpol_hydro), implosion (2-D, AMR 5-8), barotrop (1-D self-gravity:
# multigrid_fine on AMR levels + force_fine; the reference loosens it to 2e-12)
CASES = [("sod-tube", 1, 3.0e-13), ("implosion", 2, 3.0e-13), ("barotrop", 1, 2.0e-12)]


@pytest.mark.parametrize("variant", ["", "_patch"])
@pytest.mark.parametrize("name,ndim,tol", CASES)
def test_reference_golden(name, ndim, tol, variant):
    """variant "_patch": the same reference built with ramses_amd/patch as its PATCH= directory and run
    with RAMSES_AMD=0 (the shims compile for NDIM=1 and 2 and hand every call to the reference's routine)."""
    if variant and name == "implosion" and os.environ.get("RAMSES_AMD_LONG_TESTS") != "1":
        pytest.skip("50 s; the 2-D shims are exercised by the build, the 1-D ones by sod-tube and barotrop (RAMSES_AMD_LONG_TESTS=1 runs it)")
    binary = os.path.join(ROOT, "oracle", "_ref", "ramses%dd%s" % (ndim, variant))
    case = os.path.join(REF, "tests", "hydro", name)
    if not (os.path.exists(binary) and os.path.isdir(case)):
        pytest.skip("needs the reference tree and oracle/_ref/ramses%dd%s (oracle/build_ref.sh ramses %d serial [PATCHDIR])"
                    % (ndim, variant, ndim))
    from oracle import ramses_snapshot as rs
    old = os.environ.get("RAMSES_AMD")
    os.environ["RAMSES_AMD"] = "0"
    try:
        work, out = rs.run_reference(open(os.path.join(case, name + ".nml")).read(), ndim=ndim, binary=binary)
    finally:
        if old is None:
            os.environ.pop("RAMSES_AMD", None)
        else:
            os.environ["RAMSES_AMD"] = old
    try:
        outs = sorted(d for d in os.listdir(work) if d.startswith("output_"))
        leaf = rs.load_leaf_cells(os.path.join(work, outs[-1]))
    finally:
        shutil.rmtree(work, ignore_errors=True)
    gold = _read_ref_dat(os.path.join(case, name + "-ref.dat"))
    data = {"density": leaf["prim"][0], "pressure": leaf["prim"][ndim + 1], "level": leaf["level"].astype(float)}
    for d, ax in enumerate("xyz"[:ndim]):
        data["velocity_" + ax] = leaf["prim"][1 + d]
        data[ax] = leaf["x"][:, d]
    sol = _checksums(data)
    assert len(leaf["level"]) == int(gold["ncells"])
    assert sol["level"] == gold["level"]
    assert abs(leaf["info"]["t"] - gold["time"]) <= tol * gold["time"]
    for key in sorted(k for k in data if k != "level"):
        assert abs(sol[key] - gold[key]) <= tol * abs(gold[key]), (key, sol[key], gold[key])
