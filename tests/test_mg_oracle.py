"""CPU tests that PIN the multigrid oracle (oracle/mg_oracle.c) bit-for-bit
against END-TO-END runs of the reference program: tests/golden/
poisson_ref_runs.npz holds rho, phi and f of the reference's first
multigrid_fine + force_fine solve (tests/golden/make_golden_poisson.py)."""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "poisson_ref_runs.npz")
KEYS = ["l4_b1_e4", "l4_b2_e6", "l5_b1_e6"]


@pytest.mark.parametrize("key", KEYS)
def test_mg_solve_matches_reference_run(oracle, key):
    z = np.load(GOLD)
    rho_tot, boxlen, eps, iters, err = z[key + "_meta"]
    r = oracle.mg_solve_uniform(z[key + "_rho"], rho_tot, boxlen=boxlen, epsilon=eps)
    assert r["iters"] == int(iters)
    assert abs(r["err"] - err) <= 6e-4 * err          # the log prints 4 digits
    assert np.array_equal(r["phi"], z[key + "_phi"])
    f = oracle.gradient_phi_uniform(r["phi"])
    assert np.array_equal(f, z[key + "_f"])


def test_mg_operators_consistency(oracle):
    """Residual of the exact discrete solution is zero; restriction preserves the mean;
    prolongation of a constant is that constant."""
    import ctypes as C
    L = oracle.lib()
    n = 8
    rng = np.random.default_rng(0)
    phi = rng.normal(size=(n, n, n))
    dx = 0.5 ** 3
    # rhs := L phi  -> residual(phi, rhs) == 0 up to round-off
    zero = np.zeros_like(phi)
    lap = np.zeros_like(phi)
    L.ora_mg_residual(phi, zero, lap, n, dx)       # lap = -(L phi)
    res = np.zeros_like(phi)
    L.ora_mg_residual(phi, -lap, res, n, dx)
    assert np.abs(res).max() <= 1e-9 * np.abs(lap).max()
    coarse = np.zeros((n // 2,) * 3)
    L.ora_mg_restrict(phi, coarse, n)
    assert abs(coarse.mean() - phi.mean()) < 1e-14
    fine = np.zeros_like(phi)
    L.ora_mg_interp_correct(fine, np.full((n // 2,) * 3, 2.5), n)
    assert np.allclose(fine, 2.5, rtol=0, atol=1e-15)
