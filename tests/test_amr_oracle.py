"""CPU: the C restatement of the AMR godfine1 (oracle/amr_godfine_oracle.c) against dumps of
the UNMODIFIED reference (tests/golden/amr_godunov_ref.npz, made by oracle/dump_patch inside
AMR Sedov runs): unew -- and divu/enew with pressure_fix -- after godunov_fine(ilevel) on
partially refined levels, bit for bit.  This pins the oracle; the GPU tests then compare the
HIP sweep with the same dumps."""
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden", "amr_godunov_ref.npz")


def _cases():
    z = np.load(GOLD)
    return [(tag, int(c)) for tag in ("a", "b", "g", "p") for c in z[tag + "_calls"]]


@pytest.mark.parametrize("tag,call", _cases())
def test_amr_godfine_oracle_equals_reference_dump(oracle, tag, call):
    z = np.load(GOLD)
    k = "%s%d_" % (tag, call)
    ilevel, ngrid, ngridmax, ncoarse, nvar, nvec, nlevelmax, ivar, itype = [int(x) for x in z[k + "meta"]]
    dx, dt, gamma, smallr, smallc = [float(x) for x in z[k + "real"]]
    p = oracle.make_params(gamma=gamma, smallr=smallr, smallc=smallc, riemann=str(z[tag + "_riemann"]),
                           slope_type=int(z[tag + "_slope"]))
    unew = np.ascontiguousarray(z[k + "unew"]).copy()
    f = np.ascontiguousarray(z[k + "f"]) if z[k + "f"].size else None
    pfix = z[k + "divu"].size > 0
    divu = np.ascontiguousarray(z[k + "divu"]).copy() if pfix else None
    enew = np.ascontiguousarray(z[k + "enew"]).copy() if pfix else None
    oracle.godunov_fine_amr(p, z[k + "igrid"], z[k + "son"], z[k + "nbor"], z[k + "father"], ngridmax, ncoarse,
                            z[k + "uold"], unew, dx, dt, nvec, ivar, itype, f=f, divu=divu, enew=enew)
    assert np.array_equal(unew, z[k + "unew_out"]), np.abs(unew - z[k + "unew_out"]).max()
    if pfix:
        assert np.array_equal(divu, z[k + "divu_out"])
        assert np.array_equal(enew, z[k + "enew_out"])
