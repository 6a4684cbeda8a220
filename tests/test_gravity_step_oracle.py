"""The oracle chain through amr_step's gravity branch, on the CPU: starting from the initial
condition, the C restatements (sweep with the gravity predictor, Courant condition with gravity,
multigrid_fine, gradient_phi, synchro_hydro_fine, add_gravity_source_terms, newdt_fine's free-fall
limit) are stepped in the reference's order (amr/amr_step.f90:219-301,380-430) and reproduce
END-TO-END runs of the reference program with self-gravity bit for bit
(tests/golden/poisson_ref_runs.npz, made by tests/golden/make_golden_poisson.py): phi, f and the
hydro state after one coarse step (three configurations) and after three (one configuration)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden", "poisson_ref_runs.npz")
GAMMA, COURANT, SMALLR, SMALLC = 1.4, 0.8, 1e-10, 1e-10
TWOPI = 6.2831853            # amr/constants.f90:5-6: the reference's truncated value, pi = twopi/2
PI = TWOPI / 2.0


def step(oracle, p, u, f, dt_old, rho_tot, boxlen, eps, safe, rho=None):
    """one coarse step of amr_step(levelmin) for hydro + self-gravity on a uniform periodic level.
    rho: the multigrid source rho_fine produced for this step.  Its hydro deposit (multipole_fine +
    cic_from_multipole, pm/rho_fine.f90:666-891: the mass of a cell CIC-deposited at its centre of
    mass (m*x)/m) equals the cell density only where the gas has not moved yet; it stays the
    reference's host code in the product and is therefore taken from the reference run here."""
    n = u.shape[1]
    dx = boxlen / n
    if rho is None:
        rho = u[0].copy()                                             # gas at rest: the deposit is the density
    oracle.synchro_hydro(u, f, -0.5 * dt_old, SMALLR)                 # :246 old force out
    r = oracle.mg_solve_uniform(rho, rho_tot, boxlen=boxlen, epsilon=eps, safe_mode=safe)
    phi = r["phi"]
    f = oracle.gradient_phi_uniform(phi)                              # force_fine
    rho_max = float(np.abs(rho).max())
    oracle.synchro_hydro(u, f, +0.5 * dt_old, SMALLR)                 # :279 new force in
    # newdt_fine (pm/newdt_fine.f90:36-52,177): box crossing, free fall, Courant
    dt = boxlen / SMALLC
    threepi2 = 3.0 * PI ** 2
    fourpi = 4.0 * PI
    tff = np.sqrt(threepi2 / 8 / fourpi / (rho_max + SMALLR))
    dt = min(dt, COURANT * tff)
    dt = min(dt, oracle.courant_uniform(p, u, dx, COURANT, grav=f))
    unew = oracle.godunov_uniform(p, u, dx, dt, grav=f)               # set_unew + godunov_fine
    oracle.add_gravity_source(unew, u, f, dt, SMALLR)                 # set_uold: gravity source, half a step
    u = unew
    oracle.synchro_hydro(u, f, +0.5 * dt, SMALLR)                     # :428
    return u, f, phi, dt, r["iters"], r["safe_mode"]


def initial_state(rho):
    u = np.zeros((5,) + rho.shape)
    u[0] = rho
    u[4] = 1.0 / (GAMMA - 1.0)                                        # p_region = 1 everywhere, gas at rest
    return u


@pytest.mark.parametrize("key", ["l4_b1_e4", "l4_b2_e6", "l5_b1_e6"])
def test_one_coarse_step_with_self_gravity_equals_the_reference_run(oracle, key):
    sys.path.insert(0, ROOT)
    from oracle import ramses_snapshot as rs
    z = np.load(GOLD)
    rho_tot, boxlen, eps, iters, _ = [float(x) for x in z[key + "_meta"]]
    p = oracle.make_params(gamma=GAMMA, smallr=SMALLR, smallc=SMALLC)
    u = initial_state(np.ascontiguousarray(z[key + "_rho"]))
    u, f, phi, dt, it, _ = step(oracle, p, u, np.zeros((3,) + u[0].shape), 0.0, rho_tot, boxlen, eps, False)
    assert it == int(iters)
    assert np.array_equal(phi, z[key + "_phi"])
    assert np.array_equal(f, z[key + "_f"])
    assert np.array_equal(rs.cons_to_prim(u, GAMMA, SMALLR), z[key + "_prim2"])


def test_three_coarse_steps_with_self_gravity_equal_the_reference_run(oracle):
    sys.path.insert(0, ROOT)
    from oracle import ramses_snapshot as rs
    z = np.load(GOLD)
    key = "l5_b1_e6"
    rho_tot, boxlen, eps, _, _ = [float(x) for x in z[key + "_meta"]]
    p = oracle.make_params(gamma=GAMMA, smallr=SMALLR, smallc=SMALLC)
    u = initial_state(np.ascontiguousarray(z[key + "_rho"]))
    f = np.zeros((3,) + u[0].shape)
    dt, safe, its = 0.0, False, []
    for k in range(3):
        rho = np.ascontiguousarray(z[key + "_s3_rho"][k])
        if k == 0:
            assert np.array_equal(rho, u[0])
        else:
            assert 0 < np.abs(rho - u[0]).max() < 1e-11              # the CIC deposit, not the density itself
        # (rho_tot too is rho_fine's: recomputed every step from the summed multipole, last bits move)
        u, f, phi, dt, it, safe = step(oracle, p, u, f, dt, float(z[key + "_s3_rho_tot"][k]), boxlen, eps, safe, rho=rho)
        its.append(it)
    assert its == [int(x) for x in z[key + "_s3_iters"][:3]]
    g = z[key + "_s3_grav"]
    g = g[1:] if g.shape[0] == 5 else g
    assert np.array_equal(phi, g[0])
    assert np.array_equal(f, g[1:4])
    assert np.array_equal(rs.cons_to_prim(u, GAMMA, SMALLR), z[key + "_s3_prim"])


def test_three_coarse_steps_with_the_deposit_oracle_equal_the_reference_run(oracle):
    """The same three steps with NOTHING taken from the reference run but the initial state and the
    tree (oct list, xg, son/nbor/father of tests/golden/rho_fine_ref.npz, the same 32^3 run): the
    multigrid source and rho_tot come from oracle/rho_fine_oracle.c (multipole_fine +
    cic_from_multipole restated, pinned on dumps of the reference in tests/test_rho_fine_oracle.py)."""
    sys.path.insert(0, ROOT)
    from oracle import ramses_snapshot as rs
    z = np.load(GOLD)
    t = np.load(os.path.join(ROOT, "tests", "golden", "rho_fine_ref.npz"))
    key = "l5_b1_e6"
    _, boxlen, eps, _, _ = [float(x) for x in z[key + "_meta"]]
    k = "c%d_" % int(t["calls"][0])
    ilevel, _, ngrid, ngridmax, ncoarse, levelmin, nvector = [int(x) for x in t[k + "meta"]]
    n = 1 << ilevel
    igrid, xg = t[k + "igrid"], t[k + "xg"]
    # cell vector <-> brick: centre of cell (ind, oct) = xg + (bit - 1/2) dx, in units of the box
    cells, bi = [], []
    for ind in range(8):
        cells.append(ncoarse + ind * ngridmax + igrid - 1)
        bi.append([np.floor((xg[d, igrid - 1] + (((ind >> d) & 1) - 0.5) / n) * n).astype(int) for d in range(3)])
    cells = np.concatenate(cells)
    ix, iy, iz = (np.concatenate([b[d] for b in bi]) for d in range(3))
    assert cells.size == n ** 3 and len(set(zip(ix, iy, iz))) == n ** 3
    p = oracle.make_params(gamma=GAMMA, smallr=SMALLR, smallc=SMALLC)
    u = initial_state(np.ascontiguousarray(z[key + "_rho"]))
    f = np.zeros((3,) + u[0].shape)
    dt, safe = 0.0, False
    for s in range(3):
        dens = np.zeros(ncoarse + 8 * ngridmax)
        dens[cells] = u[0][iz, iy, ix]
        rho_v, _, rho_tot = oracle.rho_fine_hydro(ilevel, levelmin, nvector, igrid, xg, t[k + "son"], t[k + "nbor"], t[k + "father"],
                                                  ngridmax, ncoarse, boxlen, SMALLR, dens)
        rho = np.zeros_like(u[0])
        rho[iz, iy, ix] = rho_v[cells]
        assert np.array_equal(rho, z[key + "_s3_rho"][s]) and rho_tot == float(z[key + "_s3_rho_tot"][s])
        u, f, phi, dt, it, safe = step(oracle, p, u, f, dt, rho_tot, boxlen, eps, safe, rho=rho)
    g = z[key + "_s3_grav"]
    g = g[1:] if g.shape[0] == 5 else g
    assert np.array_equal(phi, g[0]) and np.array_equal(f, g[1:4])
    assert np.array_equal(rs.cons_to_prim(u, GAMMA, SMALLR), z[key + "_s3_prim"])
