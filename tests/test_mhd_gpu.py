"""GPU parity of the MHD sweep (SOLVER=mhd): ramses_amd_mhd_godunov_brick through the C ABI against
the COMPILED REFERENCE used the way godfine1 uses it (mhd/godunov_fine.f90:538-1022): for every oct of a periodic level
the 6^3 stencil of the eleven fields is gathered, the unmodified mag_unsplit (oracle/_ref/libref_kernels3d_mhd.so behind
oracle/ref_shim_mhd.f90) returns fluxes and EMFs, and the conservative + constrained-transport update of the oct's
eight cells is applied as the reference writes it.  Several steps in a row, bit for bit, for the 1-D / 2-D solver pairs
and slope types the device supports; div B stays at rounding.  The CPU leg of the same headers: tests/test_mhd_core_host.py."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref", "libref_kernels3d_mhd.so")


def mhd_ic(n, gamma, seed=1):
    """a blast in a magnetised, sheared medium on an n^3 periodic box; face fields from the discrete curl of an edge
    potential (div B = 0 to rounding), right faces = the neighbours' left faces.  Returns u[11, n, n, n] ([k][j][i])."""
    rng = np.random.default_rng(seed)
    x = (np.arange(n) + 0.5) / n
    xf = np.arange(n) / n                      # low faces / edges
    Z, Y, X = np.meshgrid(x, x, x, indexing="ij")
    two_pi = 2 * np.pi
    # edge-centred vector potential (Ax on x-edges: x centred, y and z on faces; ...)
    Zf, Yf, Xc = np.meshgrid(xf, xf, x, indexing="ij")
    Ax = 0.08 * np.sin(two_pi * Yf) * np.cos(two_pi * Zf) + 0.02 * np.cos(two_pi * Xc)
    Zf, Yc, Xf = np.meshgrid(xf, x, xf, indexing="ij")
    Ay = 0.06 * np.cos(two_pi * Xf) * np.sin(two_pi * Zf)
    Zc, Yf, Xf = np.meshgrid(x, xf, xf, indexing="ij")
    Az = 0.1 * np.sin(two_pi * Xf) * np.sin(two_pi * Yf) + 0.05 * np.cos(4 * np.pi * Yf)
    d = 1.0 / n
    rollp = lambda a, ax: np.roll(a, -1, axis=ax)      # noqa: E731  value at +1 along ax (axes: 0 z, 1 y, 2 x)
    bx = 0.3 + (rollp(Az, 1) - Az) / d - (rollp(Ay, 0) - Ay) / d        # on low x faces
    by = -0.2 + (rollp(Ax, 0) - Ax) / d - (rollp(Az, 2) - Az) / d       # on low y faces
    bz = 0.1 + (rollp(Ay, 2) - Ay) / d - (rollp(Ax, 1) - Ax) / d        # on low z faces
    u = np.zeros((11, n, n, n))
    u[5], u[6], u[7] = bx, by, bz
    u[8], u[9], u[10] = rollp(bx, 2), rollp(by, 1), rollp(bz, 0)
    rho = 1.0 + 0.3 * np.sin(two_pi * X) * np.cos(two_pi * Y) + 0.05 * rng.random((n, n, n))
    vel = [0.3 * np.sin(two_pi * Y), -0.25 * np.cos(two_pi * Z), 0.2 * np.sin(two_pi * (X + Y))]
    r2 = (X - 0.5) ** 2 + (Y - 0.5) ** 2 + (Z - 0.5) ** 2
    p = 0.5 + 8.0 * np.exp(-r2 / (2 * 0.08 ** 2))
    bc = [0.5 * (u[5 + c] + u[8 + c]) for c in range(3)]
    u[0] = rho
    for c in range(3):
        u[1 + c] = rho * vel[c]
    u[4] = p / (gamma - 1.0) + 0.5 * rho * sum(v * v for v in vel) + 0.5 * sum(b * b for b in bc)
    return u


class StencilReference:
    """godfine1 of a fully refined periodic level through the compiled reference's mag_unsplit"""

    def __init__(self, gamma, smallr, smallc, slope_type, theta, riemann, riemann2d, slope_mag_type=None):
        self.ref = C.CDLL(REF)
        nd, nvar, nvec = C.c_int(), C.c_int(), C.c_int()
        self.ref.ref_mhd_get_dims(C.byref(nd), C.byref(nvar), C.byref(nvec))
        assert (nd.value, nvar.value) == (3, 8)
        self.nvec = nvec.value
        self.ref.ref_mhd_set_params(C.c_double(gamma), C.c_double(smallr), C.c_double(smallc), slope_type,
                                    slope_type if slope_mag_type is None else slope_mag_type, C.c_double(theta),
                                    riemann, riemann2d)

    def step(self, u, dx, dt):
        n = u.shape[1]
        no = n // 2
        octs = [(ok, oj, oi) for ok in range(no) for oj in range(no) for oi in range(no)]
        unew = u.copy()
        nv = self.nvec
        vp = lambda a: a.ctypes.data_as(C.c_void_p)      # noqa: E731
        off = np.arange(-2, 4)
        for b0 in range(0, len(octs), nv):
            batch = octs[b0:b0 + nv]
            ng = len(batch)
            uin = np.zeros((11, 6, 6, 6, nv))
            for l, (ok, oj, oi) in enumerate(batch):
                kk, jj, ii = (2 * ok + off) % n, (2 * oj + off) % n, (2 * oi + off) % n
                uin[..., l] = u[:, kk[:, None, None], jj[None, :, None], ii[None, None, :]]
            grav = np.zeros((3, 6, 6, 6, nv))
            flux = np.zeros((3, 8, 3, 3, 3, nv))
            tmp = np.zeros((3, 2, 3, 3, 3, nv))
            emf = [np.zeros((3, 3, 3, nv)) for _ in range(3)]
            uin = np.ascontiguousarray(uin)
            self.ref.ref_mag_unsplit(vp(uin), vp(grav), vp(flux), vp(emf[0]), vp(emf[1]), vp(emf[2]), vp(tmp), C.c_double(dx), C.c_double(dx),
                                     C.c_double(dx), C.c_double(dt), ng)
            ex, ey, ez = emf                                  # [k3][j3][i3][l], indices 1..3 -> 0..2
            for l, (ok, oj, oi) in enumerate(batch):
                for k2 in range(2):
                    for j2 in range(2):
                        for i2 in range(2):
                            c = (2 * ok + k2, 2 * oj + j2, 2 * oi + i2)
                            # Euler system (mhd/godunov_fine.f90:909-942), the face fields with their fluxes reset to zero (:801-903)
                            for var in range(5):
                                v = unew[(var,) + c]
                                v = v + (flux[0, var, k2, j2, i2, l] - flux[0, var, k2, j2, i2 + 1, l])
                                v = v + (flux[1, var, k2, j2, i2, l] - flux[1, var, k2, j2 + 1, i2, l])
                                v = v + (flux[2, var, k2, j2, i2, l] - flux[2, var, k2 + 1, j2, i2, l])
                                unew[(var,) + c] = v
                            for var in range(5, 11):
                                v = unew[(var,) + c]
                                for _ in range(3):
                                    v = v + (0.0 - 0.0)
                                unew[(var,) + c] = v
                            # induction system (:966-1022)
                            i3, j3, k3 = i2, j2, k2
                            unew[(5,) + c] += (ey[k3, j3, i3, l] - ey[k3 + 1, j3, i3, l]) - (ez[k3, j3, i3, l] - ez[k3, j3 + 1, i3, l])
                            unew[(8,) + c] += (ey[k3, j3, i3 + 1, l] - ey[k3 + 1, j3, i3 + 1, l]) - (ez[k3, j3, i3 + 1, l] - ez[k3, j3 + 1, i3 + 1, l])
                            unew[(6,) + c] += (ez[k3, j3, i3, l] - ez[k3, j3, i3 + 1, l]) - (ex[k3, j3, i3, l] - ex[k3 + 1, j3, i3, l])
                            unew[(9,) + c] += (ez[k3, j3 + 1, i3, l] - ez[k3, j3 + 1, i3 + 1, l]) - (ex[k3, j3 + 1, i3, l] - ex[k3 + 1, j3 + 1, i3, l])
                            unew[(7,) + c] += (ex[k3, j3, i3, l] - ex[k3, j3 + 1, i3, l]) - (ey[k3, j3, i3, l] - ey[k3, j3, i3 + 1, l])
                            unew[(10,) + c] += (ex[k3 + 1, j3, i3, l] - ex[k3 + 1, j3 + 1, i3, l]) - (ey[k3 + 1, j3, i3, l] - ey[k3 + 1, j3, i3 + 1, l])
        return unew


def divb(u, dx):
    return ((u[8] - u[5]) + (u[9] - u[6]) + (u[10] - u[7])) / dx


@pytest.mark.parametrize("riemann,riemann2d,slope_type,n,nsteps", [
    ("llf", "llf", 1, 16, 4), ("hlld", "hlld", 2, 16, 4), ("hll", "hll", 1, 16, 3), ("hlld", "llf", 8, 16, 3),
    ("llf", "hlld", 7, 16, 3), ("upwind", "llf", 0, 16, 2), ("hlld", "hlld", 1, 24, 3), ("hlld", "hlla", 2, 16, 3),
    ("llf", "upwind", 1, 16, 3), ("hlld", "hlld", (3, 1), 16, 3), ("hydro", "llf", 1, 16, 3), ("roe", "roe", 1, 16, 3),
    ("hlld", "roe", 2, 16, 2), ("roe", "hlld", 8, 16, 2),
])
def test_mhd_sweep_equals_the_compiled_reference(gpu_lib, riemann, riemann2d, slope_type, n, nsteps):
    if not os.path.exists(REF):
        pytest.skip("oracle/_ref/libref_kernels3d_mhd.so not built")
    import torch
    from ramses_amd.mhd import RIEMANN, RIEMANN2D, MhdLevel, make_mhd_params
    gamma, smallr, smallc, theta = 5.0 / 3.0, 1e-10, 1e-10, 1.5
    u = mhd_ic(n, gamma)
    dx = 1.0 / n
    # (slope_type, slope_mag_type): slope_type = 3 goes with an explicit slope type of the face fields
    slope_type, slope_mag = slope_type if isinstance(slope_type, tuple) else (slope_type, -1)
    lev = MhdLevel(n, n, n, dx, params=make_mhd_params(gamma=gamma, smallr=smallr, smallc=smallc, slope_type=slope_type,
                                                         slope_mag_type=slope_mag, slope_theta=theta, riemann=riemann,
                                                         riemann2d=riemann2d))
    lev.upload(u)
    ref = StencilReference(gamma, smallr, smallc, slope_type, theta, RIEMANN[riemann], RIEMANN2D[riemann2d],
                           slope_mag_type=None if slope_mag == -1 else slope_mag)
    uo = u
    dt = 0.2 * dx / 4.0        # fast speed ~ 3-4 in the blast
    assert np.abs(divb(uo, dx)).max() < 1e-10
    for step in range(nsteps):
        lev.step(dt)
        uo = ref.step(uo, dx, dt)
        torch.cuda.synchronize()
        got = lev.download()
        assert np.isfinite(uo).all()
        if not np.array_equal(got, uo):
            bad = [int(v) for v in range(11) if not np.array_equal(got[v], uo[v])]
            raise AssertionError("step %d: fields %s differ, max |diff| %g" % (step + 1, bad, np.abs(got - uo).max()))
    assert np.abs(uo - u).max() > 1e-3                              # the state moved
    assert np.abs(divb(got, dx)).max() < 1e-9                       # constrained transport: div B stays at rounding
    assert np.array_equal(got[8], np.roll(got[5], -1, axis=2))      # right faces == the neighbours' left faces, bit for bit
    assert np.array_equal(got[9], np.roll(got[6], -1, axis=1)) and np.array_equal(got[10], np.roll(got[7], -1, axis=0))


def test_mhd_sweep_refuses_what_it_does_not_implement(gpu_lib):
    import torch
    from ramses_amd import RamsesAmdError
    from ramses_amd.mhd import MhdLevel, make_mhd_params
    for kw in (dict(slope_type=3), dict(slope_type=1, slope_mag_type=3), dict(slope_type=4), dict(riemann=7)):
        lev = MhdLevel(8, 8, 8, 0.125, params=make_mhd_params(**kw))
        lev.uold[0].fill_(1.0)
        lev.uold[4].fill_(1.0)
        with pytest.raises(RamsesAmdError):
            lev.godunov_fine(1e-3)
    # inconsistent faces (right face != the neighbour's left face) are refused, not silently averaged
    lev = MhdLevel(8, 8, 8, 0.125)
    lev.uold[0].fill_(1.0)
    lev.uold[4].fill_(1.0)
    lev.uold[8, 2, 3, 4] = 0.5
    with pytest.raises(RamsesAmdError, match="right-face"):
        lev.godunov_fine(1e-3)
    torch.cuda.synchronize()
