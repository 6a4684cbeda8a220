"""GPU parity tests of the multigrid Poisson path (through the C ABI) against
the oracle and against the reference-run goldens (tests/golden/poisson_ref_runs.npz)."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "poisson_ref_runs.npz")


def _dev(a):
    import torch
    return torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float64).cuda()


def _p(t):
    return C.c_void_p(t.data_ptr())


@pytest.mark.parametrize("n", [2, 4, 8, 32, 64])
def test_operators_bit_exact(gpu_lib, oracle, n):
    import torch
    L = gpu_lib
    OL = oracle.lib()
    rng = np.random.default_rng(n)
    phi = rng.normal(size=(n, n, n))
    rhs = rng.normal(size=(n, n, n))
    dx = 1.0 / n
    # Gauss-Seidel, both colours
    for red in (1, 0):
        ref = phi.copy()
        OL.ora_mg_gauss_seidel(ref, rhs, n, dx * dx, red)
        d = _dev(phi)
        assert L.ramses_amd_mg_gauss_seidel(_p(d), _p(_dev(rhs)), n, dx * dx, red, None) == 0
        torch.cuda.synchronize()
        assert np.array_equal(d.cpu().numpy(), ref)
        phi = ref
    # residual + norm
    ref = np.zeros_like(phi)
    OL.ora_mg_residual(phi, rhs, ref, n, dx)
    dres = _dev(np.zeros_like(phi))
    work = _dev(np.zeros(4096 + 8))
    norm = _dev(np.zeros(1))
    assert L.ramses_amd_mg_residual(_p(_dev(phi)), _p(_dev(rhs)), _p(dres), n, dx, _p(work), _p(norm), None) == 0
    torch.cuda.synchronize()
    assert np.array_equal(dres.cpu().numpy(), ref)
    nref = OL.ora_mg_norm2(ref, n, dx)
    assert abs(norm.item() - nref) <= 1e-13 * nref
    if n >= 4:
        # restriction
        cref = np.zeros((n // 2,) * 3)
        OL.ora_mg_restrict(ref, cref, n)
        dc = _dev(np.ones_like(cref))
        du = _dev(np.ones_like(cref))
        assert L.ramses_amd_mg_restrict(_p(dres), _p(dc), _p(du), n, None) == 0
        torch.cuda.synchronize()
        assert np.array_equal(dc.cpu().numpy(), cref)
        assert (du.cpu().numpy() == 0).all()
        # prolongation
        corr = rng.normal(size=(n // 2,) * 3)
        pref = phi.copy()
        OL.ora_mg_interp_correct(pref, corr, n)
        dp = _dev(phi)
        assert L.ramses_amd_mg_interp_correct(_p(dp), _p(_dev(corr)), n, None) == 0
        torch.cuda.synchronize()
        assert np.array_equal(dp.cpu().numpy(), pref)


@pytest.mark.parametrize("key", ["l4_b1_e4", "l4_b2_e6", "l5_b1_e6"])
def test_solve_matches_reference_run(gpu_lib, key):
    import torch
    from ramses_amd.poisson import PoissonLevel
    z = np.load(GOLD)
    rho_tot, boxlen, eps, iters, err = z[key + "_meta"]
    level = int(np.log2(z[key + "_rho"].shape[0]))
    lev = PoissonLevel(level, boxlen=boxlen, epsilon=eps)
    lev.rho.copy_(_dev(z[key + "_rho"]))
    it, e = lev.multigrid_fine(rho_tot)
    lev.force_fine()
    torch.cuda.synchronize()
    assert it == int(iters)
    assert abs(e - err) <= 6e-4 * err
    assert np.array_equal(lev.phi.cpu().numpy(), z[key + "_phi"])
    assert np.array_equal(lev.f.cpu().numpy(), z[key + "_f"])


def test_solve_64_matches_oracle(gpu_lib, oracle):
    import torch
    from ramses_amd.poisson import PoissonLevel
    n = 64
    rng = np.random.default_rng(1)
    rho = 1.0 + 0.5 * rng.random((n, n, n))
    rho[20:30, 10:40, 5:9] += 20.0
    rho_tot = float(rho.mean())
    r = oracle.mg_solve_uniform(rho, rho_tot, boxlen=1.0, epsilon=1e-6)
    lev = PoissonLevel(6, boxlen=1.0, epsilon=1e-6)
    lev.rho.copy_(_dev(rho))
    it, e = lev.multigrid_fine(rho_tot)
    lev.force_fine()
    torch.cuda.synchronize()
    assert it == r["iters"]
    assert np.array_equal(lev.phi.cpu().numpy(), r["phi"])
    assert np.array_equal(lev.f.cpu().numpy(), oracle.gradient_phi_uniform(r["phi"]))


@pytest.mark.parametrize("n,npass", [(64, 4), (64, 2), (128, 4), (70, 4), (100, 2)])
def test_fused_smoother_bit_exact(gpu_lib, oracle, n, npass):
    """The time-skewed fused smoother (+ residual + norm) equals npass colour
    passes of the plain red-black sweep, bit for bit (also for n that is not a
    multiple of the tile and smaller than it)."""
    import torch
    L, OL = gpu_lib, oracle.lib()
    rng = np.random.default_rng(n + npass)
    phi = rng.normal(size=(n, n, n))
    rhs = rng.normal(size=(n, n, n))
    dx = 1.0 / 64
    ref = phi.copy()
    for p in range(npass):
        OL.ora_mg_gauss_seidel(ref, rhs, n, dx * dx, 1 if p % 2 == 0 else 0)
    rres = np.zeros_like(ref)
    OL.ora_mg_residual(ref, rhs, rres, n, dx)
    din, dout, dres = _dev(phi), _dev(np.zeros_like(phi)), _dev(np.zeros_like(phi))
    work, norm = _dev(np.zeros(4096 + 8)), _dev(np.zeros(1))
    rc = L.ramses_amd_mg_smooth_fused(_p(din), _p(dout), _p(_dev(rhs)), _p(dres), _p(work), _p(norm), n, dx, npass, None)
    assert rc == 0, L.ramses_amd_last_error()
    torch.cuda.synchronize()
    assert np.array_equal(din.cpu().numpy(), phi)          # input untouched
    assert np.array_equal(dout.cpu().numpy(), ref)
    assert np.array_equal(dres.cpu().numpy(), rres)
    nref = OL.ora_mg_norm2(rres, n, dx)
    assert abs(norm.item() - nref) <= 1e-13 * nref
    # norm only (the residual stays on chip)
    dout3, norm3 = _dev(np.zeros_like(phi)), _dev(np.zeros(1))
    assert L.ramses_amd_mg_smooth_fused(_p(din), _p(dout3), _p(_dev(rhs)), None, _p(work), _p(norm3), n, dx, npass, None) == 0
    torch.cuda.synchronize()
    assert np.array_equal(dout3.cpu().numpy(), ref) and norm3.item() == norm.item()
    # smoother only
    dout2 = _dev(np.zeros_like(phi))
    assert L.ramses_amd_mg_smooth_fused(_p(din), _p(dout2), _p(_dev(rhs)), None, None, None, n, dx, npass, None) == 0
    torch.cuda.synchronize()
    assert np.array_equal(dout2.cpu().numpy(), ref)


def test_solve_fused_equals_unfused(gpu_lib):
    import torch
    from ramses_amd.poisson import PoissonLevel
    n = 128
    rng = np.random.default_rng(5)
    rho = 1.0 + rng.random((n, n, n))
    rho[30:60, 40:90, 10:30] += 9.0
    outs = []
    for fused in (1, 4, 16, 24, 0):     # default (2+2 passes, 32-row tiles), one 4-pass launch, other tiles, per colour
        gpu_lib.ramses_amd_mg_tune(fused)
        lev = PoissonLevel(7, boxlen=1.0, epsilon=1e-7)
        lev.rho.copy_(_dev(rho))
        it, e = lev.multigrid_fine(float(rho.mean()))
        torch.cuda.synchronize()
        outs.append((it, lev.phi.cpu().numpy()))
    gpu_lib.ramses_amd_mg_tune(1)
    for it, phi in outs[1:]:
        assert it == outs[0][0]
        assert np.array_equal(phi, outs[0][1])


@pytest.mark.parametrize("level", [8, 9])
def test_full_size_properties(gpu_lib, level):
    """BASELINE sizes (256^3 = config C4, 512^3 = the bench's V-cycle), where no CPU oracle finishes
    in seconds: size-independent properties of multigrid_fine + force_fine.
      * homogeneity: every operation of the solve is linear and a factor 2 is exact in binary
        floating point, so doubling rho and rho_tot doubles phi and f bit for bit and leaves the
        iteration count and the relative error unchanged;
      * the reported error is the one an independent evaluation gives with the per-operator
        kernels (not the fused smoother): norm of the final residual over the norm of the residual
        after the first pre-smoothing (multigrid_fine_commons.f90:197-215,267);
      * the fused smoother and the one-kernel-per-colour path give the same phi bit for bit."""
    import torch
    from ramses_amd.poisson import PoissonLevel
    n = 1 << level
    g = torch.Generator(device="cuda").manual_seed(level)
    rho = 1.0 + 0.5 * torch.rand((n, n, n), dtype=torch.float64, device="cuda", generator=g)
    a, b = n // 8, n // 3
    rho[a:a + n // 6, b:b + n // 4, n // 2:n // 2 + n // 5] += 20.0
    rho[-(n // 10):, :n // 7, n // 3:n // 2] += 7.0          # straddles the periodic boundary
    rho_tot = float(rho.mean())
    eps = 1e-7
    lev = PoissonLevel(level, boxlen=1.0, epsilon=eps)
    lev.rho.copy_(rho)
    it, err = lev.multigrid_fine(rho_tot)
    lev.force_fine()
    torch.cuda.synchronize()
    assert 2 <= it <= 10 and err < eps
    phi = lev.phi.clone()
    f = lev.f.clone()
    # independent residual: r = rhs - A phi with the per-operator kernel, rhs = fourpi*(rho-rho_tot)
    L = gpu_lib
    fourpi = 2.0 * 6.2831853 * 1.0                # 2*twopi*scale, the reference's truncated twopi
    rhs = fourpi * (rho - rho_tot)
    res = torch.empty_like(rho)
    work = torch.zeros(8192, dtype=torch.float64, device="cuda")
    nrm = torch.zeros(2, dtype=torch.float64, device="cuda")
    dx = 1.0 / n
    zero = torch.zeros_like(rho)
    for _ in range(2):                            # ngs_fine = 2 sweeps, red then black
        assert L.ramses_amd_mg_gauss_seidel(_p(zero), _p(rhs), n, dx * dx, 1, None) == 0
        assert L.ramses_amd_mg_gauss_seidel(_p(zero), _p(rhs), n, dx * dx, 0, None) == 0
    assert L.ramses_amd_mg_residual(_p(zero), _p(rhs), _p(res), n, dx, _p(work), _p(nrm), None) == 0
    n0 = float(nrm[0].item())
    assert L.ramses_amd_mg_residual(_p(phi), _p(rhs), _p(res), n, dx, _p(work), _p(nrm), None) == 0
    n1 = float(nrm[0].item())
    torch.cuda.synchronize()
    assert abs(np.sqrt(n1 / n0) - err) <= 1e-6 * err
    del res, zero, rhs
    # homogeneity
    lev.rho.copy_(2.0 * rho)
    it2, err2 = lev.multigrid_fine(2.0 * rho_tot)
    lev.force_fine()
    torch.cuda.synchronize()
    assert it2 == it and err2 == err
    assert torch.equal(lev.phi, 2.0 * phi)
    assert torch.equal(lev.f, 2.0 * f)
    del f
    # fused smoother == one kernel per colour pass
    L.ramses_amd_mg_tune(0)
    try:
        lev.rho.copy_(rho)
        it3, _ = lev.multigrid_fine(rho_tot)
        torch.cuda.synchronize()
    finally:
        L.ramses_amd_mg_tune(1)
    assert it3 == it and torch.equal(lev.phi, phi)
