"""pyoracle.py -- TEST INFRASTRUCTURE ONLY.

ctypes bindings for oracle/liboracle.so (the C restatement) and, when present,
oracle/_ref/libref_kernels{N}d.so (the reference's own routines compiled by
oracle/build_ref.sh).  Imported only by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg -- never by the product package ramses_amd/.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))

RIEMANN = {"llf": 0, "hllc": 1, "hll": 2, "acoustic": 3, "exact": 4}
SCHEME = {"muscl": 0, "plmde": 1}


class HydroParams(C.Structure):
    _fields_ = [
        ("ndim", C.c_int), ("nvar", C.c_int),
        ("gamma", C.c_double), ("smallr", C.c_double), ("smallc", C.c_double),
        ("slope_type", C.c_int), ("slope_theta", C.c_double),
        ("riemann", C.c_int), ("scheme", C.c_int), ("niter_riemann", C.c_int),
        ("difmag", C.c_double),
    ]


def make_params(ndim=3, nvar=None, gamma=1.4, smallr=1e-10, smallc=1e-10,
                slope_type=1, slope_theta=1.5, riemann="llf", scheme="muscl",
                niter_riemann=10, difmag=0.0):
    """Defaults are hydro/hydro_parameters.f90:75-89."""
    return HydroParams(ndim, nvar if nvar else ndim + 2, gamma, smallr, smallc,
                       slope_type, slope_theta, RIEMANN[riemann], SCHEME[scheme],
                       niter_riemann, difmag)


def build_oracle():
    subprocess.check_call(["make", "-s", "-C", HERE, "liboracle.so"])


_dp = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_lib = None


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(HERE, "liboracle.so")
        if not os.path.exists(path):
            build_oracle()
        L = C.CDLL(path)
        L.ora_unsplit.argtypes = [C.POINTER(HydroParams), _dp, _dp, _dp, _dp,
                                  C.c_double, C.c_double, C.c_double, C.c_double,
                                  C.c_int, C.c_int]
        L.ora_unsplit.restype = None
        L.ora_riemann.argtypes = [C.POINTER(HydroParams), _dp, _dp, _dp, C.c_int, C.c_int]
        L.ora_riemann.restype = None
        L.ora_godunov_uniform.argtypes = [C.POINTER(HydroParams), _dp, C.c_void_p, _dp,
                                          C.c_int, C.c_int, C.c_int, C.c_double, C.c_double]
        L.ora_godunov_uniform.restype = None
        L.ora_courant_uniform.argtypes = [C.POINTER(HydroParams), _dp, C.c_void_p,
                                          C.c_int, C.c_int, C.c_int, C.c_double, C.c_double]
        L.ora_courant_uniform.restype = C.c_double
        L.ora_mg_gauss_seidel.argtypes = [_dp, _dp, C.c_int, C.c_double, C.c_int]
        L.ora_mg_residual.argtypes = [_dp, _dp, _dp, C.c_int, C.c_double]
        L.ora_mg_norm2.argtypes = [_dp, C.c_int, C.c_double]
        L.ora_mg_norm2.restype = C.c_double
        L.ora_mg_restrict.argtypes = [_dp, _dp, C.c_int]
        L.ora_mg_interp_correct.argtypes = [_dp, _dp, C.c_int]
        L.ora_mg_solve_uniform.argtypes = [_dp, C.c_double, C.c_int, C.c_double, C.c_double,
                                           C.POINTER(C.c_int), _dp, _dp, _dp, C.POINTER(C.c_double)]
        L.ora_mg_solve_uniform.restype = C.c_int
        L.ora_gradient_phi_uniform.argtypes = [_dp, C.c_int, _dp]
        L.ora_interpol_hydro.argtypes = [_dp, _dp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double]
        L.ora_interpol_hydro.restype = None
        L.ora_upl.argtypes = [_dp, _dp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double]
        L.ora_upl.restype = None
        _ip = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
        L.ora_godunov_fine_amr.argtypes = [C.POINTER(HydroParams), C.c_int, _ip, _ip, _ip, _ip, C.c_long, C.c_long,
                                           _dp, _dp, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_double,
                                           C.c_int, C.c_int, C.c_int]
        L.ora_godunov_fine_amr.restype = None
        for fn in (L.ora_mg_gauss_seidel, L.ora_mg_residual, L.ora_mg_restrict, L.ora_mg_interp_correct,
                   L.ora_gradient_phi_uniform):
            fn.restype = None
        _lib = L
    return _lib


def patch_shapes(ndim, nvar, nvector):
    """Fortran-order shapes of the reference's patch arrays, returned as the
    equivalent C-order numpy shapes (reversed)."""
    nc = [6 if d < ndim else 1 for d in range(3)]
    nf = [3 if d < ndim else 1 for d in range(3)]
    uin = (nvar, nc[2], nc[1], nc[0], nvector)
    grav = (ndim, nc[2], nc[1], nc[0], nvector)
    flux = (ndim, nvar, nf[2], nf[1], nf[0], nvector)
    tmp = (ndim, 2, nf[2], nf[1], nf[0], nvector)
    return uin, grav, flux, tmp


def unsplit(p, uin, gravin, dx, dt, ngrid=None):
    """ora_unsplit on arrays shaped as patch_shapes(); returns (flux, tmp)."""
    nvector = uin.shape[-1]
    ngrid = nvector if ngrid is None else ngrid
    _, _, fs, ts = patch_shapes(p.ndim, p.nvar, nvector)
    flux = np.zeros(fs)
    tmp = np.zeros(ts)
    lib().ora_unsplit(C.byref(p), np.ascontiguousarray(uin), np.ascontiguousarray(gravin),
                      flux, tmp, dx, dx, dx, dt, ngrid, nvector)
    return flux, tmp


def riemann(p, qleft, qright):
    """qleft/qright: (nvar, nvector) -> fgdnv (nvar+1, nvector)."""
    nvector = qleft.shape[-1]
    fg = np.zeros((p.nvar + 1, nvector))
    lib().ora_riemann(C.byref(p), np.ascontiguousarray(qleft), np.ascontiguousarray(qright),
                      fg, nvector, nvector)
    return fg


def godunov_uniform(p, uold, dx, dt, grav=None):
    """One godunov_fine sweep on a periodic uniform brick uold[nvar,nz,ny,nx];
    returns unew (set_unew + godfine1 updates)."""
    nvar, nz, ny, nx = uold.shape
    uold = np.ascontiguousarray(uold)
    unew = uold.copy()
    gptr = None
    if grav is not None:
        grav = np.ascontiguousarray(grav)
        gptr = grav.ctypes.data_as(C.c_void_p)
    lib().ora_godunov_uniform(C.byref(p), uold, gptr, unew, nx, ny, nz, dx, dt)
    return unew


def courant_uniform(p, uold, dx, courant_factor, grav=None):
    nvar, nz, ny, nx = uold.shape
    gptr = None
    if grav is not None:
        grav = np.ascontiguousarray(grav)
        gptr = grav.ctypes.data_as(C.c_void_p)
    return lib().ora_courant_uniform(C.byref(p), np.ascontiguousarray(uold), gptr,
                                     nx, ny, nz, dx, courant_factor)


def godunov_fine_amr(p, igrid, son, nbor, father, ngridmax, ncoarse, uold, unew, dx, dt, nvector,
                     interpol_var, interpol_type, f=None, divu=None, enew=None):
    """godunov_fine(ilevel) on an AMR level, on the reference's tree arrays (1-based indices);
    unew (and divu, enew when given) are updated in place."""
    vp = lambda a: a.ctypes.data_as(C.c_void_p) if a is not None else None  # noqa: E731
    lib().ora_godunov_fine_amr(C.byref(p), len(igrid), np.ascontiguousarray(igrid, np.int32),
                               np.ascontiguousarray(son, np.int32), np.ascontiguousarray(nbor, np.int32),
                               np.ascontiguousarray(father, np.int32), ngridmax, ncoarse,
                               np.ascontiguousarray(uold), unew, vp(f), vp(divu), vp(enew), dx, dt, nvector,
                               interpol_var, interpol_type)


def gauss_seidel_mg_fine(ilevel, redstep, safe, igrid, son, nbor, flag2, ngridmax, ncoarse, phi, f):
    """One colour of gauss_seidel_mg_fine on an AMR level; phi[ncell] updated in place, f = (3, ncell)."""
    L = lib()
    L.ora_gauss_seidel_mg_fine.restype = None
    L.ora_gauss_seidel_mg_fine.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]
    ig, so, nb, fl = (np.ascontiguousarray(a, np.int32) for a in (igrid, son, nbor, flag2))
    assert phi.flags.c_contiguous and f.flags.c_contiguous and f.shape[0] == 3
    L.ora_gauss_seidel_mg_fine(ilevel, 1 if redstep else 0, 1 if safe else 0, len(ig), ig.ctypes.data, so.ctypes.data,
                               nb.ctypes.data, fl.ctypes.data, ngridmax, ncoarse, phi.ctypes.data, f.ctypes.data)


def cmp_residual_mg_fine(ilevel, igrid, son, nbor, flag2, ngridmax, ncoarse, phi, f):
    """cmp_residual_mg_fine on an AMR level; f[0] (= f(:,1)) updated in place."""
    L = lib()
    L.ora_cmp_residual_mg_fine.restype = None
    L.ora_cmp_residual_mg_fine.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                           C.c_int64, C.c_void_p, C.c_void_p]
    ig, so, nb, fl = (np.ascontiguousarray(a, np.int32) for a in (igrid, son, nbor, flag2))
    assert phi.flags.c_contiguous and f.flags.c_contiguous and f.shape[0] == 3
    L.ora_cmp_residual_mg_fine(ilevel, len(ig), ig.ctypes.data, so.ctypes.data, nb.ctypes.data, fl.ctypes.data, ngridmax,
                               ncoarse, phi.ctypes.data, f.ctypes.data)


def synchro_hydro(u, f, dteff, smallr=1e-10):
    """synchro_hydro_fine(ilevel,dteff,1) on a dense brick u[nvar,...] with f[3,...]: in place."""
    assert u.flags.c_contiguous and f.flags.c_contiguous and u.dtype == np.float64 and f.dtype == np.float64
    L = lib()
    L.ora_synchro_hydro.restype = None
    L.ora_synchro_hydro.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_double, C.c_double]
    L.ora_synchro_hydro(u.ctypes.data, f.ctypes.data, u[0].size, dteff, smallr)


def add_gravity_source(unew, uold, f, dt, smallr=1e-10):
    """add_gravity_source_terms on dense bricks: unew updated in place."""
    for a in (unew, uold, f):
        assert a.flags.c_contiguous and a.dtype == np.float64
    L = lib()
    L.ora_add_gravity_source.restype = None
    L.ora_add_gravity_source.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_double, C.c_double]
    L.ora_add_gravity_source(unew.ctypes.data, uold.ctypes.data, f.ctypes.data, unew[0].size, dt, smallr)


def rho_fine_hydro(ilevel, levelmin, nvector, igrid, xg, son, nbor, father, ngridmax, ncoarse, boxlen, smallr, dens):
    """rho_fine's hydro deposit on one fully refined level: returns (rho[ncell], multipole[4], rho_tot)."""
    ncell = ncoarse + 8 * ngridmax
    ig, so, nb, fa = (np.ascontiguousarray(a, np.int32) for a in (igrid, son, nbor, father))
    xg = np.ascontiguousarray(xg, np.float64)
    dens = np.ascontiguousarray(dens, np.float64)
    unew = np.zeros((4, ncell))
    rho = np.zeros(ncell)
    mp = np.zeros(4)
    rt = C.c_double()
    L = lib()
    L.ora_rho_fine_hydro.restype = None
    L.ora_rho_fine_hydro.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_int64, C.c_int64, C.c_double, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_void_p]
    L.ora_rho_fine_hydro(ilevel, levelmin, nvector, len(ig), ig.ctypes.data, xg.ctypes.data, so.ctypes.data, nb.ctypes.data,
                         fa.ctypes.data, ngridmax, ncoarse, boxlen, smallr, dens.ctypes.data, unew.ctypes.data,
                         rho.ctypes.data, mp.ctypes.data, C.addressof(rt))
    return rho, mp, rt.value


def rho_fine_amr(ilevel, nlevelmax, levelmin, nvector, first, igrid_all, xg, son, nbor, father, ngridmax, ncoarse, boxlen, smallr, dens):
    """rho_fine's hydro deposit on AMR levels (ora_rho_fine_amr): what a call rho_fine(ilevel,icount) with ilevel == levelmin
    or icount > 1 leaves.  first / igrid_all: the oct lists of levels ilevel..nlevelmax, concatenated.
    Returns (rho[ncell], multipole[4], rho_tot, unew[4, ncell])."""
    ncell = ncoarse + 8 * ngridmax
    fi, ig, so, nb, fa = (np.ascontiguousarray(a, np.int32) for a in (first, igrid_all, son, nbor, father))
    assert len(fi) == nlevelmax - ilevel + 2 and fi[-1] == len(ig)
    xg = np.ascontiguousarray(xg, np.float64)
    dens = np.ascontiguousarray(dens, np.float64)
    unew = np.zeros((4, ncell))
    rho = np.zeros(ncell)
    mp = np.zeros(4)
    rt = C.c_double()
    L = lib()
    L.ora_rho_fine_amr.restype = None
    L.ora_rho_fine_amr.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_int64, C.c_int64, C.c_double, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_void_p]
    L.ora_rho_fine_amr(ilevel, nlevelmax, levelmin, nvector, fi.ctypes.data, ig.ctypes.data, xg.ctypes.data, so.ctypes.data,
                       nb.ctypes.data, fa.ctypes.data, ngridmax, ncoarse, boxlen, smallr, dens.ctypes.data, unew.ctypes.data,
                       rho.ctypes.data, mp.ctypes.data, C.addressof(rt))
    return rho, mp, rt.value, unew


def rho_deposit_gather(ilevel, levelmin, nvector, igrid, xg, son, nbor, father, ngridmax, ncoarse, boxlen, smallr, dens):
    """The same deposit formulated as a gather with order tags (ora_rho_deposit_gather): returns rho[ncell]."""
    ncell = ncoarse + 8 * ngridmax
    ig, so, nb, fa = (np.ascontiguousarray(a, np.int32) for a in (igrid, son, nbor, father))
    xg = np.ascontiguousarray(xg, np.float64)
    dens = np.ascontiguousarray(dens, np.float64)
    unew = np.zeros((4, ncell))
    rho = np.zeros(ncell)
    mp = np.zeros(4)
    rt = C.c_double()
    L = lib()
    rho_fine_hydro(ilevel, levelmin, nvector, igrid, xg, son, nbor, father, ngridmax, ncoarse, boxlen, smallr, dens)  # argtypes
    L.ora_rho_fine_hydro(ilevel, levelmin, nvector, len(ig), ig.ctypes.data, xg.ctypes.data, so.ctypes.data, nb.ctypes.data,
                         fa.ctypes.data, ngridmax, ncoarse, boxlen, smallr, dens.ctypes.data, unew.ctypes.data,
                         rho.ctypes.data, mp.ctypes.data, C.addressof(rt))                                       # multipoles
    out = np.zeros(ncell)
    L.ora_rho_deposit_gather.restype = None
    L.ora_rho_deposit_gather.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_int64, C.c_int64, C.c_double, C.c_void_p, C.c_void_p]
    L.ora_rho_deposit_gather(ilevel, nvector, len(ig), ig.ctypes.data, xg.ctypes.data, so.ctypes.data, nb.ctypes.data,
                             fa.ctypes.data, ngridmax, ncoarse, boxlen, unew.ctypes.data, out.ctypes.data)
    return out


def rho_deposit_gather_level(level, nvector, igrid, xg, son, nbor, father, ngridmax, ncoarse, boxlen, unew):
    """The order-tagged gather (ora_rho_deposit_gather) on one level from given multipoles unew[4, ncell]: rho[ncell]
    (only the cells of the level's octs are written)."""
    ncell = ncoarse + 8 * ngridmax
    ig, so, nb, fa = (np.ascontiguousarray(a, np.int32) for a in (igrid, son, nbor, father))
    xg = np.ascontiguousarray(xg, np.float64)
    unew = np.ascontiguousarray(unew, np.float64)
    out = np.zeros(ncell)
    L = lib()
    L.ora_rho_deposit_gather.restype = None
    L.ora_rho_deposit_gather.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_int64, C.c_int64, C.c_double, C.c_void_p, C.c_void_p]
    L.ora_rho_deposit_gather(level, nvector, len(ig), ig.ctypes.data, xg.ctypes.data, so.ctypes.data, nb.ctypes.data,
                             fa.ctypes.data, ngridmax, ncoarse, boxlen, unew.ctypes.data, out.ctypes.data)
    return out


def cg_solve(igrid, son, nbor, ngridmax, ncoarse, phi, f, epsilon, itermax=10000):
    """Iteration loop of phi_fine_cg on one level (serial): phi[ncell] and f[3, ncell] (r, p, Ap) are
    updated in place from the state cmp_residual_cg left; returns (iterations, error, error_ini)."""
    assert f.shape[0] == 3 and f.flags.c_contiguous and phi.flags.c_contiguous
    err = (C.c_double * 2)()
    L = lib()
    L.ora_cg_solve.restype = C.c_int
    L.ora_cg_solve.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p,
                               C.c_double, C.c_int, C.c_void_p]
    ig, so, nb = (np.ascontiguousarray(a, np.int32) for a in (igrid, son, nbor))
    it = L.ora_cg_solve(len(ig), ig.ctypes.data, so.ctypes.data, nb.ctypes.data, ngridmax, ncoarse, phi.ctypes.data,
                        f.ctypes.data, epsilon, itermax, C.addressof(err))
    return it, err[0], err[1]


TWOPI_REF = 6.2831853   # amr/constants.f90:5 (the reference's truncated 2*pi)


def mg_solve_uniform(rho, rho_tot, boxlen=1.0, epsilon=1e-4, phi0=None, safe_mode=False):
    """multigrid_fine on a fully refined periodic level: rho[n,n,n] -> dict(phi, f1, f2, iters, err)."""
    n = rho.shape[0]
    level = int(round(np.log2(n)))
    assert rho.shape == (n, n, n) and 2 ** level == n
    fourpi = 2 * TWOPI_REF * boxlen          # 2*twopi*scale, scale = boxlen/nx_loc, nx_loc = 1
    phi = np.zeros_like(rho) if phi0 is None else np.ascontiguousarray(phi0).copy()
    f1 = np.zeros_like(rho)
    f2 = np.zeros_like(rho)
    safe = C.c_int(1 if safe_mode else 0)
    err = C.c_double()
    it = lib().ora_mg_solve_uniform(np.ascontiguousarray(rho), rho_tot, level, fourpi, epsilon,
                                    C.byref(safe), phi, f1, f2, C.byref(err))
    return dict(phi=phi, f1=f1, f2=f2, iters=it, err=err.value, safe_mode=bool(safe.value))


def gradient_phi_uniform(phi):
    n = phi.shape[0]
    level = int(round(np.log2(n)))
    f = np.zeros((3,) + phi.shape)
    lib().ora_gradient_phi_uniform(np.ascontiguousarray(phi), level, f)
    return f


# --------------------------------------------------------------------------
# the reference's own routines (oracle/_ref, built from /root/reference)
# --------------------------------------------------------------------------
_ref = {}


def _ref_path(ndim, nvar=None):
    tag = "%dd" % ndim if nvar in (None, ndim + 2) else "%dd_v%d" % (ndim, nvar)
    return os.path.join(HERE, "_ref", "libref_kernels%s.so" % tag)


def ref_available(ndim=3, nvar=None):
    return os.path.exists(_ref_path(ndim, nvar))


def ref(ndim=3, nvar=None):
    key = (ndim, nvar if nvar else ndim + 2)
    if key not in _ref:
        L = C.CDLL(_ref_path(ndim, nvar))
        L.ref_get_dims.argtypes = [C.POINTER(C.c_int)] * 3
        L.ref_set_hydro_params.argtypes = [C.c_double, C.c_double, C.c_double, C.c_int,
                                           C.c_double, C.c_int, C.c_int, C.c_int,
                                           C.c_double, C.c_double]
        L.ref_unsplit.argtypes = [_dp, _dp, _dp, _dp, C.c_double, C.c_double, C.c_double,
                                  C.c_double, C.c_int]
        L.ref_riemann.argtypes = [_dp, _dp, _dp, C.c_int]
        L.ref_cmpdt.argtypes = [_dp, _dp, C.c_double, C.POINTER(C.c_double), C.c_int]
        if hasattr(L, "ref_interpol_hydro"):
            L.ref_interpol_hydro.argtypes = [_dp, _dp, C.c_int, C.c_int, C.c_int, C.c_double]
            L.ref_upl.argtypes = [_dp, _dp, C.c_int, C.c_int, C.c_double]
        _ref[key] = L
    return _ref[key]


def ref_dims(ndim=3, nvar=None):
    a, b, c = C.c_int(), C.c_int(), C.c_int()
    ref(ndim, nvar).ref_get_dims(C.byref(a), C.byref(b), C.byref(c))
    return a.value, b.value, c.value


def ref_set_params(p, courant_factor=0.5):
    ref(p.ndim, p.nvar).ref_set_hydro_params(p.gamma, p.smallr, p.smallc, p.slope_type,
                                     p.slope_theta, p.riemann, p.scheme,
                                     p.niter_riemann, p.difmag, courant_factor)


def ref_unsplit(p, uin, gravin, dx, dt, ngrid=None):
    nvector = uin.shape[-1]
    ngrid = nvector if ngrid is None else ngrid
    assert ref_dims(p.ndim, p.nvar) == (p.ndim, p.nvar, nvector)
    ref_set_params(p)
    _, _, fs, ts = patch_shapes(p.ndim, p.nvar, nvector)
    flux = np.zeros(fs)
    tmp = np.zeros(ts)
    ref(p.ndim, p.nvar).ref_unsplit(np.ascontiguousarray(uin), np.ascontiguousarray(gravin),
                            flux, tmp, dx, dx, dx, dt, ngrid)
    return flux, tmp


def ref_riemann(p, qleft, qright):
    nvector = qleft.shape[-1]
    assert ref_dims(p.ndim, p.nvar) == (p.ndim, p.nvar, nvector)
    ref_set_params(p)
    fg = np.zeros((p.nvar + 1, nvector))
    ref(p.ndim, p.nvar).ref_riemann(np.ascontiguousarray(qleft), np.ascontiguousarray(qright), fg, nvector)
    return fg


def ref_cmpdt(p, uu, gg, dx, courant_factor):
    """uu: (nvar, nvector) conservative, gg: (ndim, nvector)."""
    nvector = uu.shape[-1]
    ref_set_params(p, courant_factor)
    dt = C.c_double()
    ref(p.ndim, p.nvar).ref_cmpdt(np.ascontiguousarray(uu).copy(), np.ascontiguousarray(gg), dx,
                          C.byref(dt), nvector)
    return dt.value
