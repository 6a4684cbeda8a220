"""The C restatement of the conjugate-gradient solver's iteration loop (oracle/cg_oracle.c)
against dumps of the UNMODIFIED reference (oracle/dump_patch/phi_fine_cg.f90 ->
tests/golden/cg_ref.npz): same iteration count, phi and r/p/Ap bit for bit.  CPU only."""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(__file__), "golden", "cg_ref.npz")


def load(z, s):
    k = "s%d_" % s
    ilevel, ngrid, ngridmax, ncoarse = [int(x) for x in z[k + "meta"]]
    epsilon, rho_tot, boxlen = [float(x) for x in z[k + "real"]]
    ncell = ncoarse + 8 * ngridmax
    f = np.zeros((3, ncell))
    f[:2] = z[k + "f"]
    return dict(ilevel=ilevel, ngrid=ngrid, ngridmax=ngridmax, ncoarse=ncoarse, epsilon=epsilon, igrid=z[k + "igrid"],
                son=z[k + "son"], nbor=z[k + "nbor"], phi=z[k + "phi"].copy(), f=f, phi_out=z[k + "phi_out"], f_out=z[k + "f_out"])


@pytest.mark.parametrize("solve", [1, 2, 6])
def test_cg_oracle_equals_reference_dump(oracle, solve):
    z = np.load(GOLD)
    d = load(z, solve)
    iters_ref = int(z["solves"][solve - 1][1])
    assert int(z["solves"][solve - 1][0]) == d["ilevel"]
    it, err, err_ini = oracle.cg_solve(d["igrid"], d["son"], d["nbor"], d["ngridmax"], d["ncoarse"], d["phi"], d["f"], d["epsilon"])
    assert it == iters_ref
    assert np.array_equal(d["phi"], d["phi_out"])
    lev = np.zeros(d["phi"].size, bool)
    for ind in range(8):
        lev[d["ncoarse"] + ind * d["ngridmax"] + d["igrid"] - 1] = True
    assert np.array_equal(d["f"][:, lev], d["f_out"][:, lev])
    assert np.array_equal(d["phi"][~lev], z["s%d_phi" % solve][~lev])         # nothing off the level is touched
    # the log line's relative error (error/error_ini, printed with 3 digits)
    assert abs(err / err_ini - z["errors"][solve - 1][1]) <= 6e-4 * z["errors"][solve - 1][1]
