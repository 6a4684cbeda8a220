"""rho_fine's hydro deposit on the device (ramses_amd/csrc/rho_fine.hip through
ramses_amd_resident_rho_fine_f90) against dumps of the UNMODIFIED reference
(tests/golden/rho_fine_ref.npz, made by tests/golden/make_golden_rho.py from a self-gravity run in
which the gas moves, so the deposit is not the cell density bit for bit): rho of every cell of the
level and the four multipole sums (hence rho_tot), bit for bit."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "rho_fine_ref.npz")


@pytest.mark.parametrize("which", [0, 1])
@pytest.mark.parametrize("nvector", [32, 7])
def test_device_deposit_equals_reference_dump(gpu_lib, oracle, which, nvector):
    import ramses_amd
    z = np.load(GOLD)
    k = "c%d_" % int(z["calls"][which])
    ilevel, icount, ngrid, ngridmax, ncoarse, levelmin, nvec_ref = [int(x) for x in z[k + "meta"]]
    boxlen, smallr = [float(x) for x in z[k + "real"]]
    igrid = np.ascontiguousarray(z[k + "igrid"])
    xg = np.ascontiguousarray(z[k + "xg"])
    ncell = ncoarse + 8 * ngridmax
    rng = np.random.default_rng(1)
    uold = rng.uniform(0.5, 1.5, (5, ncell))
    uold[0] = z[k + "dens"]
    if nvector == nvec_ref:
        want_rho, want_mp = z[k + "rho"], z[k + "multipole"]
    else:
        # another NVECTOR changes the order of the sums: the oracle (pinned on the dumps at the reference's own
        # NVECTOR) gives the expected values
        want_rho, want_mp, _ = oracle.rho_fine_hydro(ilevel, levelmin, nvector, igrid, xg, z[k + "son"], z[k + "nbor"],
                                                     z[k + "father"], ngridmax, ncoarse, boxlen, smallr, z[k + "dens"])
    p = ramses_amd.make_params(smallr=smallr)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    assert gpu_lib.ramses_amd_resident_invalidate() == 0
    mp = np.zeros(4)
    rc = gpu_lib.ramses_amd_resident_rho_fine_f90(C.byref(p), ilevel, ngrid, vp(igrid), vp(xg), ngridmax, ncoarse, 1,
                                                  vp(uold), boxlen, nvector, vp(mp))
    assert rc == 0, gpu_lib.ramses_amd_last_error()
    assert np.array_equal(mp, want_mp), (mp, want_mp)
    phi = np.zeros(ncell)
    f = np.zeros((3, ncell))
    rho = np.full(ncell, -7.0)
    assert gpu_lib.ramses_amd_resident_sync_poisson_f90(vp(phi), vp(f), vp(rho)) == 0
    lev = np.zeros(ncell, bool)
    for ind in range(8):
        lev[ncoarse + ind * ngridmax + igrid - 1] = True
    assert np.array_equal(rho[lev], want_rho[lev]), np.abs(rho[lev] - want_rho[lev]).max()
    assert (rho[~lev] == -7.0).all()            # cells of other levels are untouched
    assert (want_rho[lev] != z[k + "dens"][lev]).any()
    assert gpu_lib.ramses_amd_resident_invalidate() == 0


@pytest.mark.parametrize("level,nvector", [(6, 32), (6, 5), (7, 32)])
def test_multipole_scan_equals_the_sequential_sum(gpu_lib, level, nvector):
    """multipole(1:4) are strictly sequential floating-point sums over all cells (cic_from_multipole,
    pm/rho_fine.f90:858-866); the device reproduces them with a scan of parity functions.  Compared with the
    sequential sums (numpy cumsum = left-to-right accumulation) on densities chosen to stress it: many decades,
    exact powers of two and small dyadic multiples (ties in the rounding), zeros below smallr."""
    import ramses_amd
    n, no = 2 ** level, 2 ** (level - 1)
    ngrid = no ** 3
    ngridmax, ncoarse = ngrid + 3, 1
    ncell = ncoarse + 8 * ngridmax
    rng = np.random.default_rng(level * 100 + nvector)
    perm = rng.permutation(ngrid)                       # list order != position order
    oz, oy, ox = np.unravel_index(perm, (no, no, no))
    igrid = np.arange(1, ngrid + 1, dtype=np.int32)
    xg = np.zeros((3, ngridmax))
    for d, o in enumerate((ox, oy, oz)):
        xg[d, :ngrid] = (o + 0.5) / no
    dens = np.zeros(ncell)
    kind = rng.integers(0, 4, ncell)
    dens[kind == 0] = 10.0 ** rng.uniform(-8, 6, (kind == 0).sum())
    dens[kind == 1] = 2.0 ** rng.integers(-30, 20, (kind == 1).sum())
    dens[kind == 2] = rng.integers(1, 64, (kind == 2).sum()) * 2.0 ** rng.integers(-40, 0, (kind == 2).sum())
    dens[kind == 3] = 0.0
    uold = np.zeros((5, ncell))
    uold[0] = dens
    boxlen, smallr = 1.0, 1e-10
    p = ramses_amd.make_params(smallr=smallr)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    assert gpu_lib.ramses_amd_resident_invalidate() == 0
    mp = np.zeros(4)
    rc = gpu_lib.ramses_amd_resident_rho_fine_f90(C.byref(p), level, ngrid, vp(igrid), vp(xg), ngridmax, ncoarse, 1,
                                                  vp(uold), boxlen, nvector, vp(mp))
    assert rc == 0, gpu_lib.ramses_amd_last_error()
    assert gpu_lib.ramses_amd_resident_invalidate() == 0
    # the reference's order: batches of nvector octs, ind_son, oct in the batch
    dx = 0.5 ** level
    vol_loc = (dx * boxlen) ** 3
    want = np.zeros(4)
    chunks = [[], [], [], []]
    for b0 in range(0, ngrid, nvector):
        g = igrid[b0:b0 + nvector]
        for ind in range(8):
            cells = ncoarse + ind * ngridmax + g - 1
            mm = np.maximum(dens[cells], smallr) * vol_loc
            chunks[0].append(mm)
            for d in range(3):
                xc = (((ind >> d) & 1) - 0.5) * dx
                xx = (xg[d, g - 1] + xc - 0.0) * boxlen
                chunks[d + 1].append(mm * xx)
    for c in range(4):
        want[c] = np.cumsum(np.concatenate(chunks[c]))[-1]
    assert np.array_equal(mp, want), (mp, want, mp - want)


GOLD_AMR = os.path.join(os.path.dirname(__file__), "golden", "rho_fine_amr_ref.npz")


@pytest.mark.parametrize("which", [0, 1])
@pytest.mark.parametrize("nvector", [32, 5])
def test_amr_device_deposit_equals_reference_dump(gpu_lib, oracle, which, nvector):
    """rho_fine on AMR levels (missing #1): ramses_amd_amrres_rho_fine -- multipoles of leaf and split
    cells, the order-tagged CIC gather through the tree, the sequential multipole sums at levelmin -- on the reference's own
    cell vectors and tree against dumps of the UNMODIFIED reference in a self-gravitating AMR run (every level the call
    visits, partially refined ones included; tests/golden/rho_fine_amr_ref.npz), bit for bit; with another NVECTOR against
    the oracle (ora_rho_fine_amr, pinned on the same dumps)."""
    import ramses_amd
    from ramses_amd._capi import check
    z = np.load(GOLD_AMR)
    k = "c%d_" % int(z["calls"][which])
    ilevel, icount, ngrid, ngridmax, ncoarse, levelmin, nvec_ref = [int(x) for x in z[k + "meta"]]
    boxlen, smallr = [float(x) for x in z[k + "real"]]
    nlevelmax = int(z[k + "nlevelmax"][0])
    first = np.ascontiguousarray(z[k + "first"], np.int32)
    igrid_all = np.ascontiguousarray(z[k + "igrid_all"], np.int32)
    son, nbor, father = (np.ascontiguousarray(z[k + n], np.int32) for n in ("son", "nbor", "father"))
    xg = np.ascontiguousarray(z[k + "xg"])
    ncell = ncoarse + 8 * ngridmax
    if nvector == nvec_ref:
        want_rho, want_mp = z[k + "rho"], z[k + "multipole"]
    else:
        want_rho, want_mp, _, _ = oracle.rho_fine_amr(ilevel, nlevelmax, levelmin, nvector, first, igrid_all, xg, son, nbor, father,
                                                      ngridmax, ncoarse, boxlen, smallr, z[k + "dens"])
    rng = np.random.default_rng(5)
    uold = rng.uniform(0.5, 1.5, (5, ncell))
    uold[0] = z[k + "dens"]
    p = ramses_amd.make_params(smallr=smallr)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    L = gpu_lib
    check(L.ramses_amd_amrres_load(5, ngridmax, ncoarse, vp(uold), vp(son), vp(nbor), vp(father)))
    try:
        rho = np.full(ncell, -7.0)
        mp = np.full(4, -1.0)
        # without the oct centres the call refuses
        assert L.ramses_amd_amrres_rho_fine(C.byref(p), ilevel, nlevelmax, levelmin, nvector, vp(first), vp(igrid_all), boxlen,
                                            vp(rho), vp(mp)) != 0
        check(L.ramses_amd_amrres_xg(vp(xg)))
        check(L.ramses_amd_amrres_rho_fine(C.byref(p), ilevel, nlevelmax, levelmin, nvector, vp(first), vp(igrid_all), boxlen,
                                           vp(rho), vp(mp)))
        visited = np.zeros(ncell, bool)
        nlev = 0
        for li in range(len(first) - 1):
            ig = igrid_all[first[li]:first[li + 1]]
            if len(ig) == 0:
                continue
            lev = np.zeros(ncell, bool)
            for ind in range(8):
                lev[ncoarse + ind * ngridmax + ig - 1] = True
            assert np.array_equal(rho[lev], want_rho[lev]), (ilevel + li, np.abs(rho[lev] - want_rho[lev]).max())
            visited |= lev
            nlev += 1
        assert nlev >= 2
        assert (rho[~visited] == -7.0).all()
        if ilevel == levelmin:
            assert np.array_equal(mp, want_mp), (mp, want_mp)
        else:
            assert (mp == -1.0).all()                  # levelmin is not visited: the sums are not touched
        # a second call gives the same bits (the oct -> list position table is clean again)
        rho2 = np.full(ncell, -7.0)
        check(L.ramses_amd_amrres_rho_fine(C.byref(p), ilevel, nlevelmax, levelmin, nvector, vp(first), vp(igrid_all), boxlen,
                                           vp(rho2), vp(mp)))
        assert np.array_equal(rho2.view(np.int64), rho.view(np.int64))
    finally:
        check(L.ramses_amd_amrres_invalidate())


def test_multipole_scan_many_workgroups_256(gpu_lib):
    """The multi-workgroup scan at the size of config C4 (256^3: 2048 segments of 8192 cells per component): equal to the
    left-to-right sums, on a density with a wide dynamic range so that the running sum crosses many binades."""
    import ramses_amd
    level, nvector = 8, 32
    n, no = 2 ** level, 2 ** (level - 1)
    ngrid = no ** 3
    ngridmax, ncoarse = ngrid + 1, 1
    ncell = ncoarse + 8 * ngridmax
    rng = np.random.default_rng(77)
    perm = rng.permutation(ngrid)
    oz, oy, ox = np.unravel_index(perm, (no, no, no))
    igrid = np.arange(1, ngrid + 1, dtype=np.int32)
    xg = np.zeros((3, ngridmax))
    for d, o in enumerate((ox, oy, oz)):
        xg[d, :ngrid] = (o + 0.5) / no
    dens = np.zeros(ncell)
    dens[:] = 10.0 ** rng.uniform(-6, 3, ncell)
    dens[rng.integers(0, ncell, 1000)] = 2.0 ** rng.integers(-20, 24, 1000).astype(float)     # binade jumps, exact ties
    uold = np.zeros((5, ncell))
    uold[0] = dens
    boxlen, smallr = 1.0, 1e-10
    p = ramses_amd.make_params(smallr=smallr)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    assert gpu_lib.ramses_amd_resident_invalidate() == 0
    mp = np.zeros(4)
    rc = gpu_lib.ramses_amd_resident_rho_fine_f90(C.byref(p), level, ngrid, vp(igrid), vp(xg), ngridmax, ncoarse, 1,
                                                  vp(uold), boxlen, nvector, vp(mp))
    assert rc == 0, gpu_lib.ramses_amd_last_error()
    assert gpu_lib.ramses_amd_resident_invalidate() == 0
    dx = 0.5 ** level
    vol_loc = (dx * boxlen) ** 3
    g = igrid.reshape(-1, nvector)                       # ngrid is a multiple of nvector here
    want = np.zeros(4)
    cols = [[] for _ in range(4)]
    for ind in range(8):
        cells = ncoarse + ind * ngridmax + g - 1         # [batch, j]
        mm = np.maximum(dens[cells], smallr) * vol_loc
        cols[0].append(mm)
        for d in range(3):
            xc = (((ind >> d) & 1) - 0.5) * dx
            cols[d + 1].append(mm * ((xg[d, g - 1] + xc - 0.0) * boxlen))
    for c in range(4):
        seq = np.stack(cols[c], axis=1).reshape(-1)       # (batch, ind_son, j) order
        want[c] = np.cumsum(seq)[-1]
    assert np.array_equal(mp, want), (mp, want, mp - want)
