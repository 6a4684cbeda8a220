"""The C restatement of rho_fine's hydro deposit (oracle/rho_fine_oracle.c: multipole_fine +
cic_from_multipole / cic_cell) against dumps of the UNMODIFIED reference
(oracle/dump_patch/rho_fine.f90 -> tests/golden/rho_fine_ref.npz): rho of every cell of the level,
the multipole sums and rho_tot, bit for bit, on states where the gas moves (the deposit then
differs from the cell density in the last bits).  CPU only."""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(__file__), "golden", "rho_fine_ref.npz")


@pytest.mark.parametrize("which", [0, 1])
def test_rho_fine_oracle_equals_reference_dump(oracle, which):
    z = np.load(GOLD)
    c = int(z["calls"][which])
    k = "c%d_" % c
    ilevel, icount, ngrid, ngridmax, ncoarse, levelmin, nvector = [int(x) for x in z[k + "meta"]]
    boxlen, smallr = [float(x) for x in z[k + "real"]]
    rho, mp, rho_tot = oracle.rho_fine_hydro(ilevel, levelmin, nvector, z[k + "igrid"], z[k + "xg"], z[k + "son"], z[k + "nbor"],
                                             z[k + "father"], ngridmax, ncoarse, boxlen, smallr, z[k + "dens"])
    lev = np.zeros(rho.size, bool)
    for ind in range(8):
        lev[ncoarse + ind * ngridmax + z[k + "igrid"] - 1] = True
    assert (z[k + "rho"][lev] != z[k + "dens"][lev]).any()          # the golden is not the trivial case
    assert np.array_equal(rho[lev], z[k + "rho"][lev]), np.abs(rho[lev] - z[k + "rho"][lev]).max()
    assert np.array_equal(mp, z[k + "multipole"])
    assert rho_tot == float(z[k + "rho_tot"][0])


@pytest.mark.parametrize("which", [0, 1])
def test_gather_formulation_of_the_deposit_equals_the_sequential_one(oracle, which):
    """The deposit as a gather with order tags (one target cell at a time, contributions produced in
    reverse order, sorted by their position in the reference's loop nest, then added): the shape a
    device kernel needs.  Same rho as the reference's sequential accumulation, bit for bit."""
    z = np.load(GOLD)
    k = "c%d_" % int(z["calls"][which])
    ilevel, icount, ngrid, ngridmax, ncoarse, levelmin, nvector = [int(x) for x in z[k + "meta"]]
    boxlen, smallr = [float(x) for x in z[k + "real"]]
    rho = oracle.rho_deposit_gather(ilevel, levelmin, nvector, z[k + "igrid"], z[k + "xg"], z[k + "son"], z[k + "nbor"],
                                    z[k + "father"], ngridmax, ncoarse, boxlen, smallr, z[k + "dens"])
    lev = np.zeros(rho.size, bool)
    for ind in range(8):
        lev[ncoarse + ind * ngridmax + z[k + "igrid"] - 1] = True
    assert np.array_equal(rho[lev], z[k + "rho"][lev]), np.abs(rho[lev] - z[k + "rho"][lev]).max()
