"""godunov_fine of a FULLY COVERED level of an AMR run through the dense sweep (since round 5: on the device's tiles,
csrc/amr_layout.hpp -- levels of 64^3 cells and more; tests/test_amr_tiles_gpu.py holds the same path against the ORACLE)
(hydro/godunov_fine.f90:661-666 `ok`, :720-747 fluxes reset at refined faces, :752-790 the update
of unew, which already holds what the finer level owes to this one) against the tree-walking sweep of the same level, which
the reference dumps, goldens and live A/B runs pin bit for bit (tests/test_amr_godunov_gpu.py, test_baseline_sizes_gpu.py).
One amr_step's worth of calls on the resident state: set_unew on both levels, godunov_fine of the partially refined finer
level (its corrections land in unew of the covered level), godunov_fine of the covered level, set_uold -- once with the
dense path, once with RAMSES_AMD_COVERED_DENSE=0.  uold of both levels must be equal bit for bit, the counter must say which
path ran.  (Every single-rank AMR test of the suite exercises the same path through the patched program.)"""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _dense_sweep_on_small_levels_too(monkeypatch):
    """(levels below RAMSES_AMD_TILE_MIN_OCTS octs take the tree-walking sweep in production: the tests force the tiles)"""
    monkeypatch.setenv("RAMSES_AMD_TILE_MIN_OCTS", "0")


def _state(T, L, seed):
    """a smooth flow with a blast on the covered level L, its restriction-consistent copy on the finer octs"""
    rng = np.random.default_rng(seed)
    n = 2 ** L
    x = (np.arange(n) + 0.5) / n
    Z, Y, X = np.meshgrid(x, x, x, indexing="ij")
    rho = 1.0 + 0.3 * np.sin(2 * np.pi * X) * np.cos(2 * np.pi * Y) + 0.05 * rng.random((n, n, n))
    vel = [0.3 * np.sin(2 * np.pi * Y), -0.2 * np.cos(2 * np.pi * Z), 0.25 * np.sin(2 * np.pi * (X + Z))]
    r2 = (X - 0.5) ** 2 + (Y - 0.5) ** 2 + (Z - 0.5) ** 2
    p = 0.4 + 6.0 * np.exp(-r2 / (2 * 0.07 ** 2))
    u = np.zeros((5, n, n, n))
    u[0] = rho
    for d in range(3):
        u[1 + d] = rho * vel[d]
    u[4] = p / 0.4 + 0.5 * rho * sum(v * v for v in vel)
    vec = np.zeros((5, T["ncell"]))
    vec[0] = 1.0
    vec[4] = 1.0
    T["to_cells"](u, vec)
    # the finer octs: the father cell's state plus a small perturbation (any positive state will do)
    son, ngm, nc = T["son"], T["ngridmax"], T["ncoarse"]
    for g in T["igrid_fine"]:
        fc = int(T["father"][g - 1]) - 1
        for ind in range(8):
            c = nc + ind * ngm + g - 1
            vec[:, c] = vec[:, fc] * (1.0 + 0.01 * (ind - 3.5) / 8.0)
    return vec


@pytest.mark.parametrize("L,riemann,slope_type,grav", [(6, 0, 1, False), (6, 1, 2, False), (6, 2, 8, True), (6, 3, 7, False), (6, 0, 1, True)])
def test_dense_masked_sweep_equals_the_tree_walking_sweep(gpu_lib, monkeypatch, L, riemann, slope_type, grav):
    import ramses_amd
    from ramses_amd import ic
    from ramses_amd._capi import check
    Lb = gpu_lib
    n = 2 ** L
    z, y, x = np.meshgrid(np.arange(n), np.arange(n), np.arange(n), indexing="ij")
    r = np.sqrt((x - n / 2 + 0.5) ** 2 + (y - n / 2 + 0.5) ** 2 + (z - n / 2 + 0.5) ** 2)
    mask = (r >= 0.2 * n) & (r <= 0.33 * n)
    mask[0, 0, :3] = True                                   # refined cells on the periodic seam too
    T = ic.uniform_tree(L, order="morton", refine_mask=mask)
    vec0 = _state(T, L, 7 + L)
    p = ramses_amd.make_params(riemann=riemann, slope_type=slope_type, fast_math=False)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)      # noqa: E731
    ig, igf = np.ascontiguousarray(T["igrid"]), np.ascontiguousarray(T["igrid_fine"])
    dx = 1.0 / n
    dt = 0.1 * dx
    rng = np.random.default_rng(3)
    f = np.ascontiguousarray(0.5 * rng.standard_normal((3, T["ncell"]))) if grav else None

    def run(dense):
        monkeypatch.setenv("RAMSES_AMD_COVERED_DENSE", "1" if dense else "0")
        u = vec0.copy()
        check(Lb.ramses_amd_amrres_invalidate())
        check(Lb.ramses_amd_amrres_load(5, T["ngridmax"], T["ncoarse"], vp(u), vp(T["son"]), vp(T["nbor"]), vp(T["father"])))
        if grav:
            check(Lb.ramses_amd_amrres_load_f(len(ig), vp(ig), vp(f)))
            check(Lb.ramses_amd_amrres_load_f(len(igf), vp(igf), vp(f)))
        before = Lb.ramses_amd_amrres_covered_sweeps()
        check(Lb.ramses_amd_amrres_set_unew(len(ig), vp(ig)))
        check(Lb.ramses_amd_amrres_set_unew(len(igf), vp(igf)))
        check(Lb.ramses_amd_amrres_godunov(C.byref(p), L + 1, len(igf), vp(igf), dx / 2, dt / 2, 32, 0, 1))
        check(Lb.ramses_amd_amrres_godunov(C.byref(p), L, len(ig), vp(ig), dx, dt, 32, 0, 1))
        took = Lb.ramses_amd_amrres_covered_sweeps() - before
        check(Lb.ramses_amd_amrres_set_uold(C.byref(p), len(igf), vp(igf)))
        check(Lb.ramses_amd_amrres_set_uold(C.byref(p), len(ig), vp(ig)))
        check(Lb.ramses_amd_amrres_sync_level(len(ig), vp(ig), vp(u)))
        check(Lb.ramses_amd_amrres_sync_level(len(igf), vp(igf), vp(u)))
        return u, took

    tree, took0 = run(False)
    dense, took1 = run(True)
    assert took0 == 0 and took1 == 1            # the finer level is not covered: it walks the tree in both runs
    assert np.abs(tree - vec0).max() > 1e-3
    # the corrections of the finer level reached the covered level (unew != uold before its own sweep): the refined cells
    # themselves keep their state, their unrefined neighbours moved
    assert np.array_equal(dense, tree), np.abs(dense - tree).max()
    check(Lb.ramses_amd_amrres_invalidate())
