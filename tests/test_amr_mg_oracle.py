"""The C restatement of the fine-level AMR multigrid routines (oracle/mg_amr_oracle.c) against
dumps of the UNMODIFIED reference (oracle/dump_patch/multigrid_fine_fine.f90 ->
tests/golden/amr_mg_ref.npz): gauss_seidel_mg_fine (red, black) and cmp_residual_mg_fine on a
partially refined level with cells on the masked branch, bit for bit.  CPU only."""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(__file__), "golden", "amr_mg_ref.npz")


@pytest.mark.parametrize("call", [81, 82, 1021])
def test_amr_mg_oracle_equals_reference_dump(oracle, call):
    z = np.load(GOLD)
    k = "c%d_" % call
    kind, ilevel, ngrid, ngridmax, ncoarse, iflag = [int(x) for x in z[k + "meta"]]
    phi = np.ascontiguousarray(z[k + "phi"]).copy()
    f = np.ascontiguousarray(z[k + "f"]).copy()
    args = (z[k + "igrid"], z[k + "son"], z[k + "nbor"], z[k + "flag2"], ngridmax, ncoarse, phi, f)
    if kind == 1:
        oracle.gauss_seidel_mg_fine(ilevel, bool(iflag & 1), bool(iflag & 2), *args)
        assert (phi != z[k + "phi"]).any()
    else:
        oracle.cmp_residual_mg_fine(ilevel, *args)
        assert (f[0] != z[k + "f"][0]).any()
    assert np.array_equal(phi, z[k + "phi_out"])
    assert np.array_equal(f[0], z[k + "f1_out"])
    assert np.array_equal(f[1:], z[k + "f"][1:])
