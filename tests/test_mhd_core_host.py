"""The MHD arithmetic of the product (ramses_amd/csrc/mhd_core.hpp + mhd_assemble.hpp, the headers csrc/mhd_sweep.hip is
built from) against the COMPILED REFERENCE on the CPU: the headers are compiled for the host (tests/native/
mhd_host_check.cpp) and driven over batches of the reference's own 6^3 stencils; mag_unsplit of the unmodified
reference (mhd/umuscl.f90:31-238 behind oracle/ref_shim_mhd.f90, oracle/_ref/libref_kernels3d_mhd.so) sees the same
stencils.  Fluxes of the five Euler variables and the three edge EMFs must be equal bit for bit, for every supported
combination of 1-D solver (llf, roe, hll, hlld, upwind, hydro), 2-D solver (llf, roe, upwind, hll, hlla, hlld) and slope type (0, 1, 2, 7, 8, and 3 with slope_mag_type 1 / 8).
SURVEY.md 8 row f4; the GPU leg is tests/test_mhd_gpu.py."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref", "libref_kernels3d_mhd.so")
SRC = os.path.join(ROOT, "tests", "native", "mhd_host_check.cpp")


@pytest.fixture(scope="module")
def libs(tmp_path_factory):
    if not os.path.exists(REF):
        pytest.skip("oracle/_ref/libref_kernels3d_mhd.so not built (oracle/build_ref.sh kernels_mhd)")
    out = str(tmp_path_factory.mktemp("mhd") / "libmhd_host_check.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-shared", "-fPIC", "-o", out, SRC])
    ref = C.CDLL(REF)
    nd, nvar, nvec = C.c_int(), C.c_int(), C.c_int()
    ref.ref_mhd_get_dims(C.byref(nd), C.byref(nvar), C.byref(nvec))
    assert (nd.value, nvar.value) == (3, 8)
    return C.CDLL(out), ref, nvec.value


def stencils(nvec, seed, kind):
    """uin(nvector,-1:4,-1:4,-1:4,11) in Fortran order: a smooth magnetised flow with a jump across a tilted plane; the
    face fields are consistent (the right face of a cell is the left face of its neighbour), as on a real level"""
    rng = np.random.default_rng(seed)
    u = np.zeros((11, 6, 6, 6, nvec))           # C order [var][k][j][i][l]  == Fortran (l,i,j,k,var)
    k, j, i = np.meshgrid(np.arange(6), np.arange(6), np.arange(6), indexing="ij")
    for l in range(nvec):
        ph = rng.uniform(0, 2 * np.pi, 8)
        nrm = rng.standard_normal(3)
        side = ((i - 2.5) * nrm[0] + (j - 2.5) * nrm[1] + (k - 2.5) * nrm[2] > rng.uniform(-1, 1)).astype(float)
        amp = 0.3 if kind == "smooth" else 1.0
        rho = 1.0 + 0.2 * np.sin(0.7 * i + ph[0]) * np.cos(0.5 * j + ph[1]) + amp * 2.0 * side
        vel = [0.4 * np.sin(0.6 * k + ph[2]) + amp * 0.8 * side, 0.3 * np.cos(0.8 * i + ph[3]) - amp * 0.5 * side, 0.2 * np.sin(0.9 * j + ph[4])]
        p = 0.6 + 0.1 * np.cos(0.4 * (i + j + k) + ph[5]) + amp * 3.0 * side
        # face fields on a 7^3 lattice of faces (low face of cell i = face i)
        f = np.arange(7)
        kk, jj, ii = np.meshgrid(f, f, f, indexing="ij")
        bface = [0.5 + 0.3 * np.sin(0.7 * jj + ph[6]) + 0.2 * np.cos(0.5 * kk), -0.4 + 0.3 * np.cos(0.6 * ii + ph[7]) + 0.1 * kk / 6.0,
                 0.3 + 0.25 * np.sin(0.8 * ii + 0.3 * jj)]
        bl = [bface[0][:6, :6, :6], bface[1][:6, :6, :6], bface[2][:6, :6, :6]]
        br = [bface[0][:6, :6, 1:], bface[1][:6, 1:, :6], bface[2][1:, :6, :6]]
        bc = [0.5 * (a + b) for a, b in zip(bl, br)]
        u[0, ..., l] = rho
        for d in range(3):
            u[1 + d, ..., l] = rho * vel[d]
            u[5 + d, ..., l] = bl[d]
            u[8 + d, ..., l] = br[d]
        u[4, ..., l] = p / (5.0 / 3.0 - 1.0) + 0.5 * rho * sum(v * v for v in vel) + 0.5 * sum(b * b for b in bc)
    return np.ascontiguousarray(u)


@pytest.mark.parametrize("slope_type", [1, 2, 0, 7, 8, (3, 1), (3, 8)])
@pytest.mark.parametrize("riemann,riemann2d", [(0, 0), (3, 5), (2, 3), (3, 0), (0, 5), (4, 0), (3, 4), (0, 2), (2, 2), (4, 4), (5, 0), (5, 5), (1, 0), (1, 1), (3, 1), (1, 5)])
@pytest.mark.parametrize("kind", ["smooth", "jump"])
def test_headers_equal_the_compiled_reference(libs, slope_type, riemann, riemann2d, kind):
    host, ref, nvec = libs
    # (slope_type, slope_mag_type): uslope has no branch 3 for the face fields, slope_type = 3 goes with an explicit slope_mag_type
    slope_type, slope_mag_type = slope_type if isinstance(slope_type, tuple) else (slope_type, slope_type)
    gamma, smallr, smallc, theta = 5.0 / 3.0, 1e-10, 1e-10, 1.5
    uin = stencils(nvec, 100 * slope_type + 10 * riemann + riemann2d, kind)
    dx, dt = 1.0 / 64, 0.2 / 64
    grav = np.zeros((3, 6, 6, 6, nvec))
    vp = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    dbl = C.c_double
    # the reference
    flux_r = np.full((3, 8, 3, 3, 3, nvec), np.nan)
    tmp_r = np.full((3, 2, 3, 3, 3, nvec), np.nan)
    emf_r = [np.full((3, 3, 3, nvec), np.nan) for _ in range(3)]
    ref.ref_mhd_set_params(dbl(gamma), dbl(smallr), dbl(smallc), slope_type, slope_mag_type, dbl(theta), riemann, riemann2d)
    ref.ref_mag_unsplit(vp(uin), vp(grav), vp(flux_r), vp(emf_r[0]), vp(emf_r[1]), vp(emf_r[2]), vp(tmp_r), dbl(dx), dbl(dx), dbl(dx), dbl(dt), nvec)
    # the product's headers on the host
    flux_h = np.full_like(flux_r, np.nan)
    emf_h = [np.full_like(e, np.nan) for e in emf_r]
    rc = host.mhd_host_unsplit(vp(uin), nvec, nvec, dbl(dx), dbl(dt), dbl(gamma), dbl(smallr), dbl(smallc), slope_type, slope_mag_type, dbl(theta),
                               riemann, riemann2d, vp(flux_h), vp(emf_h[0]), vp(emf_h[1]), vp(emf_h[2]))
    assert rc == 0
    # where mag_unsplit defines its outputs (:100-236): fluxes through the faces of the central 2^3 cells, EMFs on their edges
    for d in range(3):
        sl = [slice(0, 3 if d == 2 else 2), slice(0, 3 if d == 1 else 2), slice(0, 3 if d == 0 else 2)]      # [k][j][i]
        a, b = flux_h[d][:5][:, sl[0], sl[1], sl[2]], flux_r[d][:5][:, sl[0], sl[1], sl[2]]
        assert np.isfinite(b).all()
        assert np.array_equal(a, b), (d, np.abs(a - b).max())
    for e in range(3):
        sl = [slice(0, 2 if e == 2 else 3), slice(0, 2 if e == 1 else 3), slice(0, 2 if e == 0 else 3)]
        a, b = emf_h[e][sl[0], sl[1], sl[2]], emf_r[e][sl[0], sl[1], sl[2]]
        assert np.isfinite(b).all()
        assert np.array_equal(a, b), (e, np.abs(a - b).max())


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_cmpdt_cell_equals_the_compiled_reference(libs, seed):
    """cmpdt of the MHD solver (mhd/godunov_utils.f90:5-115; fast magnetosonic speed, ischeme = 0): the product's cmpdt_cell
    (csrc/mhd_core.hpp, what the resident courant_fine runs per cell) against the compiled routine, cell by cell and as a
    minimum over a vector of cells"""
    host, ref, nvec = libs
    rng = np.random.default_rng(seed)
    gamma, smallr, smallc, cfl = 5.0 / 3.0, 1e-10, 1e-10, 0.8
    ref.ref_mhd_set_params(C.c_double(gamma), C.c_double(smallr), C.c_double(smallc), 1, 1, C.c_double(1.5), 0, 0)
    ref.ref_mhd_cmpdt.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_int, C.c_void_p]
    host.mhd_host_cmpdt.restype = C.c_double
    host.mhd_host_cmpdt.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double]
    u = np.zeros((11, nvec))
    u[0] = rng.uniform(1e-3, 3.0, nvec)
    u[0, :2] = [1e-12, 5e-11]                       # below the density floor
    vel = rng.normal(0.0, 1.5, (3, nvec))
    u[1:4] = u[0] * vel
    u[5:8] = rng.normal(0.0, 1.0, (3, nvec))
    u[8:11] = u[5:8] + rng.normal(0.0, 0.05, (3, nvec))
    bc = 0.5 * (u[5:8] + u[8:11])
    pth = rng.uniform(1e-6, 2.0, nvec)
    pth[2] = -1.0                                   # a cell whose pressure falls on the floor
    u[4] = pth / (gamma - 1.0) + 0.5 * u[0] * (vel ** 2).sum(0) + 0.5 * (bc ** 2).sum(0)
    for dx in (1.0 / 64, 0.37):
        for ncell in (1, 7, nvec):
            work = np.ascontiguousarray(u).copy()
            dt = C.c_double()
            ref.ref_mhd_cmpdt(work.ctypes.data_as(C.c_void_p), dx, cfl, ncell, C.byref(dt))
            got = host.mhd_host_cmpdt(np.ascontiguousarray(u).ctypes.data_as(C.c_void_p), ncell, nvec, dx, cfl, gamma, smallr, smallc)
            assert np.float64(got).view(np.int64) == np.float64(dt.value).view(np.int64), (dx, ncell, got, dt.value)
        # cell by cell
        for l in range(0, nvec, 5):
            one = np.zeros((11, nvec))
            one[:, 0] = u[:, l]
            work = one.copy()
            dt = C.c_double()
            ref.ref_mhd_cmpdt(work.ctypes.data_as(C.c_void_p), dx, cfl, 1, C.byref(dt))
            got = host.mhd_host_cmpdt(one.ctypes.data_as(C.c_void_p), 1, nvec, dx, cfl, gamma, smallr, smallc)
            assert np.float64(got).view(np.int64) == np.float64(dt.value).view(np.int64)


@pytest.mark.parametrize("riemann,riemann2d,slope_type", [(3, 5, 2), (0, 0, 1), (1, 1, 8)])
def test_gravity_predictor_of_ctoprim_equals_the_compiled_reference(libs, riemann, riemann2d, slope_type):
    """ctoprim's half-step gravity kick (mhd/umuscl.f90 ctoprim: u, v, w += gravin * dt / 2) through the whole of mag_unsplit:
    the branch of ctoprim_cell the device does not use yet (self-gravitating MHD runs stay the reference's), held ready"""
    host, ref, nvec = libs
    gamma, smallr, smallc, theta = 5.0 / 3.0, 1e-10, 1e-10, 1.5
    uin = stencils(nvec, 4242 + riemann, "jump")
    rng = np.random.default_rng(7)
    grav = np.ascontiguousarray(rng.normal(0.0, 2.0, (3, 6, 6, 6, nvec)))
    dx, dt = 1.0 / 64, 0.2 / 64
    vp = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    dbl = C.c_double
    flux_r = np.full((3, 8, 3, 3, 3, nvec), np.nan)
    tmp_r = np.full((3, 2, 3, 3, 3, nvec), np.nan)
    emf_r = [np.full((3, 3, 3, nvec), np.nan) for _ in range(3)]
    ref.ref_mhd_set_params(dbl(gamma), dbl(smallr), dbl(smallc), slope_type, slope_type, dbl(theta), riemann, riemann2d)
    ref.ref_mag_unsplit(vp(uin), vp(grav), vp(flux_r), vp(emf_r[0]), vp(emf_r[1]), vp(emf_r[2]), vp(tmp_r), dbl(dx), dbl(dx), dbl(dx), dbl(dt), nvec)
    flux_h = np.full_like(flux_r, np.nan)
    emf_h = [np.full_like(e, np.nan) for e in emf_r]
    host.mhd_host_set_gravin(vp(grav))
    try:
        rc = host.mhd_host_unsplit(vp(uin), nvec, nvec, dbl(dx), dbl(dt), dbl(gamma), dbl(smallr), dbl(smallc), slope_type, slope_type, dbl(theta),
                                   riemann, riemann2d, vp(flux_h), vp(emf_h[0]), vp(emf_h[1]), vp(emf_h[2]))
    finally:
        host.mhd_host_set_gravin(None)
    assert rc == 0
    for d in range(3):
        sl = [slice(0, 3 if d == 2 else 2), slice(0, 3 if d == 1 else 2), slice(0, 3 if d == 0 else 2)]
        a, b = flux_h[d][:5][:, sl[0], sl[1], sl[2]], flux_r[d][:5][:, sl[0], sl[1], sl[2]]
        assert np.array_equal(a, b), (d, np.abs(a - b).max())
    for e in range(3):
        sl = [slice(0, 2 if e == 2 else 3), slice(0, 2 if e == 1 else 3), slice(0, 2 if e == 0 else 3)]
        assert np.array_equal(emf_h[e][sl[0], sl[1], sl[2]], emf_r[e][sl[0], sl[1], sl[2]])
    # and gravity does change the answer
    g0 = np.zeros_like(grav)
    flux_0 = np.full_like(flux_r, np.nan)
    ref.ref_mag_unsplit(vp(uin), vp(g0), vp(flux_0), vp(emf_r[0]), vp(emf_r[1]), vp(emf_r[2]), vp(tmp_r), dbl(dx), dbl(dx), dbl(dx), dbl(dt), nvec)
    assert not np.array_equal(flux_0[0][:5, :2, :2, :3], flux_r[0][:5, :2, :2, :3])
