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


GOLD_AMR = os.path.join(os.path.dirname(__file__), "golden", "rho_fine_amr_ref.npz")


@pytest.mark.parametrize("which", [0, 1])
def test_rho_fine_oracle_on_amr_levels_equals_reference_dump(oracle, which):
    """rho_fine on AMR levels (ora_rho_fine_amr): multipoles of split cells as the sums of their children, the deposit of a
    partially refined level (CIC corners without an oct are dropped), every level the call visits -- against dumps of the
    unmodified reference in the self-gravitating AMR run (levels 3-5; tests/golden/make_golden_rho_amr.py): rho of every cell
    of every visited level, multipole(1:4) and rho_tot, bit for bit."""
    z = np.load(GOLD_AMR)
    k = "c%d_" % int(z["calls"][which])
    ilevel, icount, ngrid, ngridmax, ncoarse, levelmin, nvector = [int(x) for x in z[k + "meta"]]
    boxlen, smallr = [float(x) for x in z[k + "real"]]
    nlevelmax = int(z[k + "nlevelmax"][0])
    first, igrid_all = z[k + "first"], z[k + "igrid_all"]
    assert np.array_equal(igrid_all[first[0]:first[1]], z[k + "igrid"])
    rho, mp, rho_tot, _ = oracle.rho_fine_amr(ilevel, nlevelmax, levelmin, nvector, first, igrid_all, z[k + "xg"], z[k + "son"],
                                              z[k + "nbor"], z[k + "father"], ngridmax, ncoarse, boxlen, smallr, z[k + "dens"])
    son = z[k + "son"]
    nvisited, npartial = 0, 0
    for li in range(len(first) - 1):
        ig = igrid_all[first[li]:first[li + 1]]
        if len(ig) == 0:
            continue
        lev = np.zeros(rho.size, bool)
        for ind in range(8):
            lev[ncoarse + ind * ngridmax + ig - 1] = True
        assert np.array_equal(rho[lev], z[k + "rho"][lev]), (ilevel + li, np.abs(rho[lev] - z[k + "rho"][lev]).max())
        nvisited += 1
        npartial += int(len(ig) * 8 != 8 ** (ilevel + li))            # a partially refined level
        if (son[lev] > 0).any():
            # a level with split cells: their deposited mass comes from the children's multipoles, not from their own density
            assert (z[k + "rho"][lev] != z[k + "dens"][lev]).any()
    assert nvisited >= 2 and npartial >= 1
    if ilevel == levelmin:
        assert np.array_equal(mp, z[k + "multipole"])
        assert rho_tot == float(z[k + "rho_tot"][0])


@pytest.mark.parametrize("which", [0, 1])
def test_gather_formulation_on_amr_levels(oracle, which):
    """The order-tagged gather -- the shape a device kernel needs -- on every level an AMR call visits, from the multipoles
    of ora_rho_fine_amr: sources whose target oct does not exist drop out, a target collects from the octs that exist.
    Same rho as the reference's sequential accumulation on partially refined levels, bit for bit."""
    z = np.load(GOLD_AMR)
    k = "c%d_" % int(z["calls"][which])
    ilevel, icount, ngrid, ngridmax, ncoarse, levelmin, nvector = [int(x) for x in z[k + "meta"]]
    boxlen, smallr = [float(x) for x in z[k + "real"]]
    nlevelmax = int(z[k + "nlevelmax"][0])
    first, igrid_all = z[k + "first"], z[k + "igrid_all"]
    _, _, _, unew = oracle.rho_fine_amr(ilevel, nlevelmax, levelmin, nvector, first, igrid_all, z[k + "xg"], z[k + "son"],
                                        z[k + "nbor"], z[k + "father"], ngridmax, ncoarse, boxlen, smallr, z[k + "dens"])
    checked = 0
    for li in range(len(first) - 1):
        ig = igrid_all[first[li]:first[li + 1]]
        if len(ig) == 0:
            continue
        rho = oracle.rho_deposit_gather_level(ilevel + li, nvector, ig, z[k + "xg"], z[k + "son"], z[k + "nbor"], z[k + "father"],
                                              ngridmax, ncoarse, boxlen, unew)
        lev = np.zeros(rho.size, bool)
        for ind in range(8):
            lev[ncoarse + ind * ngridmax + ig - 1] = True
        assert np.array_equal(rho[lev], z[k + "rho"][lev]), (ilevel + li, np.abs(rho[lev] - z[k + "rho"][lev]).max())
        checked += 1
    assert checked >= 2
