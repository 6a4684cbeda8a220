"""Conjugate-gradient Poisson solver (phi_fine_cg) on the MI355X:
kernel level against dumps of the UNMODIFIED reference (tests/golden/cg_ref.npz, made by
tests/golden/make_golden_cg.py with oracle/dump_patch/phi_fine_cg.f90), against the C oracle on
a synthetic partially refined tree, and end to end through the patched reference program."""
import ctypes as C
import importlib.util
import os
import re
import shutil
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden", "cg_ref.npz")
vp = lambda a: a.ctypes.data_as(C.c_void_p) if a is not None else None  # noqa: E731


def _solve(gpu_lib, d, ordered, rho=None, rho_tot=0.0, fact=1.0, itermax=10000):
    """through the host mirror ramses_amd.amr.phi_fine_cg (the C ABI's ramses_amd_cg_solve_host);
    ordered: 1 parity scan, 2 one-lane chain, 0 parallel tree, -1 what the environment says"""
    from ramses_amd import amr
    tree = amr.AmrTree(d["son"], d["nbor"], np.zeros(d["ngridmax"], np.int32), d["ngridmax"], d["ncoarse"])
    it, e, e_ini, rhs = amr.phi_fine_cg(tree, d["ilevel"], d["igrid"], d["phi"], d["f"], d["epsilon"], ordered=int(ordered),
                                        rho=rho, rho_tot=rho_tot, fact=fact, itermax=itermax)
    return it, [e, e_ini, rhs]


def _load(z, s):
    from test_cg_oracle import load
    d = load(z, s)
    for k in ("igrid", "son", "nbor"):
        d[k] = np.ascontiguousarray(d[k], np.int32)
    return d


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [-1, 1, 2])
@pytest.mark.parametrize("solve", [1, 2, 6])
def test_cg_ordered_equals_reference_dump(gpu_lib, solve, mode, monkeypatch):
    """Dot products summed in the reference's order -- by the parallel parity scan (1; -1: what the environment says, which
    without RAMSES_AMD_CG_ORDERED is the same: the DEFAULT) or by a one-lane chain (2): iteration count, phi and r/p/Ap bit
    for bit."""
    monkeypatch.delenv("RAMSES_AMD_CG_ORDERED", raising=False)
    z = np.load(GOLD)
    d = _load(z, solve)
    it, err = _solve(gpu_lib, d, ordered=mode)
    assert it == int(z["solves"][solve - 1][1])
    assert np.array_equal(d["phi"], d["phi_out"]), np.abs(d["phi"] - d["phi_out"]).max()
    lev = np.zeros(d["phi"].size, bool)
    for ind in range(8):
        lev[d["ncoarse"] + ind * d["ngridmax"] + d["igrid"] - 1] = True
    assert np.array_equal(d["f"][:, lev], d["f_out"][:, lev])
    assert np.array_equal(d["f"][:2, ~lev], z["s%d_f" % solve][:, ~lev])       # cells off the level untouched
    assert abs(err[0] / err[1] - z["errors"][solve - 1][1]) <= 6e-4 * z["errors"][solve - 1][1]


@pytest.mark.gpu
@pytest.mark.parametrize("solve", [1, 2, 6])
def test_cg_parallel_sums_agree_with_reference_dump(gpu_lib, solve):
    """RAMSES_AMD_CG_ORDERED=0 (fixed parallel reduction tree, not the default any more): same iteration count on these
    solves, phi equal to 1e-12 of its range, and the result is reproducible run to run."""
    z = np.load(GOLD)
    d = _load(z, solve)
    it, _ = _solve(gpu_lib, d, ordered=0)
    assert it == int(z["solves"][solve - 1][1])
    scale = np.abs(d["phi_out"]).max()
    assert np.abs(d["phi"] - d["phi_out"]).max() <= 1e-12 * scale
    d2 = _load(z, solve)
    it2, _ = _solve(gpu_lib, d2, ordered=0)
    assert it2 == it and np.array_equal(d2["phi"], d["phi"])


@pytest.mark.gpu
@pytest.mark.parametrize("ordered", [1, 2, 0])
def test_cg_on_a_synthetic_tree_equals_the_oracle(gpu_lib, oracle, ordered):
    """Random right-hand side on a two-level synthetic tree (level 5 octs in a box crossing the
    periodic boundary, zero outside the level): HIP against oracle/cg_oracle.c, incl. rhs_norm."""
    from helpers import uniform_tree
    T = uniform_tree(4, refine_box=((13, 20), (2, 9), (6, 12)))
    rng = np.random.default_rng(5)
    ncell = T["ncell"]
    igrid = np.ascontiguousarray(T["igrid_fine"], np.int32)
    lev = T["fine_cells"]() - 1
    phi0 = np.zeros(ncell)
    phi0[lev] = rng.normal(size=lev.size)
    f0 = np.zeros((3, ncell))
    f0[0, lev] = rng.normal(size=lev.size)
    f0[1] = f0[0]
    f0[1, ~np.isin(np.arange(ncell), lev)] = 0.0
    rho = rng.random(ncell)
    d = dict(ilevel=5, ngrid=len(igrid), ngridmax=T["ngridmax"], ncoarse=T["ncoarse"], igrid=igrid, son=T["son"], nbor=T["nbor"],
             epsilon=1e-8, phi=phi0.copy(), f=f0.copy())
    it, err = _solve(gpu_lib, d, ordered, rho=rho, rho_tot=0.5, fact=0.37)
    phi, f = phi0.copy(), f0.copy()
    it_o, e_o, e_ini_o = oracle.cg_solve(igrid, T["son"], T["nbor"], T["ngridmax"], T["ncoarse"], phi, f, 1e-8)
    assert it_o > 20
    rhs_o = np.float64(0.0)
    for ind in range(8):          # the reference's summation order (:63-70)
        for g in igrid:
            dd = rho[T["ncoarse"] + ind * T["ngridmax"] + g - 1] - 0.5
            rhs_o = rhs_o + (0.37 * 0.37) * dd * dd
    rhs_o = np.sqrt(rhs_o / (8.0 * len(igrid)))
    if ordered:
        assert it == it_o and np.array_equal(d["phi"], phi) and np.array_equal(d["f"], f)
        assert err[0] == e_o and err[1] == e_ini_o and err[2] == rhs_o
    else:
        assert abs(it - it_o) <= 1
        assert np.abs(d["phi"] - phi).max() <= 1e-7 * np.abs(phi).max()     # both converged to epsilon = 1e-8
        assert abs(err[2] - rhs_o) <= 1e-14 * rhs_o


@pytest.mark.gpu
@pytest.mark.parametrize("ordered", ["default", "chain", "0"])
def test_patched_program_with_cg_levels_equals_reference(gpu_lib, ordered):
    """The self-gravitating AMR run with cg_levelmin=4 (levels 4 and 5 solved by phi_fine_cg, level 3
    by multigrid) through the patched reference program: ordered sums (the DEFAULT: no environment variable; or the
    one-lane chain) -> the reference's snapshot bit for bit and the same iteration counts; parallel sums
    (RAMSES_AMD_CG_ORDERED=0) -> equal to rounding."""
    patched = os.path.join(ROOT, "oracle", "_ref", "ramses3d_patch")
    if not os.path.exists(patched):
        pytest.skip("oracle/_ref/ramses3d_patch not built")
    from oracle import ramses_snapshot as rs
    spec = importlib.util.spec_from_file_location("mkcg", os.path.join(ROOT, "tests", "golden", "make_golden_cg.py"))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    z = np.load(GOLD)
    os.environ.pop("RAMSES_AMD_CG_ORDERED", None)
    if ordered != "default":
        os.environ["RAMSES_AMD_CG_ORDERED"] = ordered
    try:
        work, out = rs.run_reference(mk.cg_namelist(), binary=patched)
    finally:
        os.environ.pop("RAMSES_AMD_CG_ORDERED", None)
    try:
        assert "phi_fine_cg" not in out or "MI355X" in out
        solves = re.findall(r"==> Level=\s*(\d+) Step=\s*(\d+) Error=\s*(\S+)[ \t]+(\S+)[ \t]*\n", out)
        solves = np.array([[int(a), int(b)] for a, b, _, _ in solves])
        snap = rs.load_leaf_cells(os.path.join(work, "output_00002"), with_grav=True)
        order = np.lexsort((snap["x"][:, 0], snap["x"][:, 1], snap["x"][:, 2], snap["level"]))
        assert np.array_equal(snap["level"][order], z["level"])
        if ordered != "0":
            assert np.array_equal(solves, z["solves"])
            assert np.array_equal(snap["grav"][:, order], z["grav"]), np.abs(snap["grav"][:, order] - z["grav"]).max()
            assert np.array_equal(snap["prim"][:, order], z["prim"])
        else:
            assert solves.shape == z["solves"].shape and np.abs(solves - z["solves"]).max() <= 1
            assert np.abs(snap["grav"][:, order] - z["grav"]).max() <= 1e-9 * np.abs(z["grav"]).max()
            assert np.abs(snap["prim"][:, order] - z["prim"]).max() <= 1e-11 * np.abs(z["prim"]).max()
    finally:
        shutil.rmtree(work, ignore_errors=True)


@pytest.mark.gpu
def test_patched_program_with_cg_levels_inside_walls_equals_reference(gpu_lib):
    """The same run inside six reflexive walls (ordered sums): boundary octs are neighbours whose p
    the solver reads but never updates, and level 3 = levelmin takes the AMR multigrid path."""
    patched = os.path.join(ROOT, "oracle", "_ref", "ramses3d_patch")
    if not os.path.exists(patched):
        pytest.skip("oracle/_ref/ramses3d_patch not built")
    from oracle import ramses_snapshot as rs
    spec = importlib.util.spec_from_file_location("mkcg", os.path.join(ROOT, "tests", "golden", "make_golden_cg.py"))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    z = np.load(GOLD)
    os.environ["RAMSES_AMD_CG_ORDERED"] = "1"
    try:
        work, out = rs.run_reference(mk.cg_walls_namelist(), binary=patched)
    finally:
        os.environ.pop("RAMSES_AMD_CG_ORDERED", None)
    try:
        solves = re.findall(r"==> Level=\s*(\d+) Step=\s*(\d+) Error=\s*(\S+)[ \t]+(\S+)[ \t]*\n", out)
        assert np.array_equal(np.array([[int(a), int(b)] for a, b, _, _ in solves]), z["w_solves"])
        snap = rs.load_leaf_cells(os.path.join(work, "output_00002"), with_grav=True)
        order = np.lexsort((snap["x"][:, 0], snap["x"][:, 1], snap["x"][:, 2], snap["level"]))
        assert np.array_equal(snap["level"][order], z["w_level"])
        assert np.array_equal(snap["grav"][:, order], z["w_grav"]), np.abs(snap["grav"][:, order] - z["w_grav"]).max()
        assert np.array_equal(snap["prim"][:, order], z["w_prim"])
    finally:
        shutil.rmtree(work, ignore_errors=True)


@pytest.mark.gpu
@pytest.mark.parametrize("ordered", [1, 0])
def test_cg_homogeneity_on_a_full_level(gpu_lib, ordered):
    """A fully refined periodic level of 2.1 M cells (no CPU oracle run at this size): every
    operation of the iteration is linear and a factor 2 is exact, so doubling r and p doubles phi
    and r/p/Ap bit for bit with the same iteration count (alpha and beta are ratios) -- with the ordered sums of
    the parity scan (1) as with the parallel tree (0)."""
    from helpers import uniform_tree
    L = 7
    T = uniform_tree(L, order="morton")
    ncell = T["ncell"]
    igrid = np.ascontiguousarray(T["igrid"], np.int32)
    rng = np.random.default_rng(9)
    lev = np.concatenate([T["ncoarse"] + ind * T["ngridmax"] + igrid - 1 for ind in range(8)])
    r0 = np.zeros(ncell)
    r0[lev] = rng.normal(size=lev.size)
    r0[lev] -= r0[lev].mean()
    outs = []
    for scale in (1.0, 2.0):
        f = np.zeros((3, ncell))
        f[0] = scale * r0
        f[1] = scale * r0
        d = dict(ilevel=L, ngrid=len(igrid), ngridmax=T["ngridmax"], ncoarse=T["ncoarse"], igrid=igrid,
                 son=np.ascontiguousarray(T["son"], np.int32), nbor=np.ascontiguousarray(T["nbor"], np.int32),
                 epsilon=1e-6, phi=np.zeros(ncell), f=f)
        it, err = _solve(gpu_lib, d, ordered=ordered)
        outs.append((it, err, d["phi"], d["f"]))
    (it1, e1, phi1, f1), (it2, e2, phi2, f2) = outs
    assert it1 == it2 and it1 > 30
    assert e2[0] == 2.0 * e1[0] and e2[1] == 2.0 * e1[1]
    assert np.array_equal(phi2, 2.0 * phi1) and np.array_equal(f2, 2.0 * f1)
    assert np.abs(phi1[lev]).max() > 0 and e1[0] <= 1e-6 * e1[1]


@pytest.mark.gpu
def test_cg_scan_equals_the_one_lane_chain_on_a_full_level(gpu_lib):
    """2.1 M cells, 12 iterations: the ordered sums of the parallel parity scan (the default) and of the one-lane chain
    (N dependent adds in the reference's order) give the same bits in phi, r, p, Ap and the residual norms -- the signed
    products p*Ap included."""
    from helpers import uniform_tree
    L = 7
    T = uniform_tree(L, order="morton")
    ncell = T["ncell"]
    igrid = np.ascontiguousarray(T["igrid"], np.int32)
    rng = np.random.default_rng(11)
    lev = np.concatenate([T["ncoarse"] + ind * T["ngridmax"] + igrid - 1 for ind in range(8)])
    r0 = np.zeros(ncell)
    r0[lev] = rng.normal(size=lev.size)
    r0[lev] -= r0[lev].mean()
    outs = []
    for mode in (1, 2):
        f = np.zeros((3, ncell))
        f[0] = r0
        f[1] = r0
        d = dict(ilevel=L, ngrid=len(igrid), ngridmax=T["ngridmax"], ncoarse=T["ncoarse"], igrid=igrid,
                 son=np.ascontiguousarray(T["son"], np.int32), nbor=np.ascontiguousarray(T["nbor"], np.int32),
                 epsilon=1e-30, phi=np.zeros(ncell), f=f)
        it, (e, e_ini, _) = _solve(gpu_lib, d, mode, itermax=12)
        outs.append((it, e, e_ini, d["phi"], d["f"]))
    a, b = outs
    assert a[0] == b[0] == 12 and a[1] == b[1] and a[2] == b[2]
    assert np.array_equal(a[3], b[3]) and np.array_equal(a[4], b[4])
    assert (a[4][1][lev] * a[4][2][lev] < 0).any()          # p*Ap has negative terms: the signed scan was exercised
