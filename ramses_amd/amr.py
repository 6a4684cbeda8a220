"""Host-side mirror of the reference's per-level interface on the tree arrays themselves
(AMR levels, MPI ranks, boxes with physical boundaries), on host numpy arrays in the reference's
layout (cell = ncoarse + ind*ngridmax + igrid, 1-based oct indices):

    godunov_fine(ilevel)            hydro/godunov_fine.f90:5-35, godfine1 :486-911
    phi_fine_cg(ilevel,icount)      poisson/phi_fine_cg.f90:88-187 (the iteration loop)

Each call stages its arrays to the device, runs the HIP kernels through the C ABI and copies the
results back (the staged entry points the Fortran shims use).  There is no CPU fallback.
"""
import ctypes as C

import numpy as np

from ._capi import RamsesAmdError, check, lib


def _vp(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _i32(a):
    a = np.ascontiguousarray(a, np.int32)
    return a


class AmrTree:
    """son(1:ncell), nbor(1:ngridmax,1:6) [stored (6, ngridmax)], father(1:ngridmax) of amr_commons
    (amr/amr_commons.f90:67-75), with ncoarse and ngridmax."""

    def __init__(self, son, nbor, father, ngridmax, ncoarse):
        self.son, self.nbor, self.father = _i32(son), _i32(nbor), _i32(father)
        self.ngridmax, self.ncoarse = int(ngridmax), int(ncoarse)
        self.ncell = self.ncoarse + 8 * self.ngridmax
        if self.son.size != self.ncell or self.nbor.size != 6 * self.ngridmax or self.father.size != self.ngridmax:
            raise RamsesAmdError("tree arrays do not match ncoarse/ngridmax")


def godunov_fine(params, tree, ilevel, igrid, uold, unew, dx, dt, nvector=32, interpol_var=0, interpol_type=1,
                 f=None, divu=None, enew=None):
    """godunov_fine(ilevel) on the octs `igrid` of a level: unew (nvar, ncell) is updated in place
    with the flux differences of the level's cells and the corrections owed to the coarser level;
    f = gravity (3, ncell) switches on the predictor's source term, divu/enew the pressure_fix
    bookkeeping.  nvector fixes the order of the coarse-level corrections (bit parity with a
    reference built with the same NVECTOR)."""
    igrid = _i32(igrid)
    for a in (uold, unew, f, divu, enew):
        if a is not None and not (a.dtype == np.float64 and a.flags.c_contiguous):
            raise RamsesAmdError("arrays must be C-contiguous float64")
    check(lib().ramses_amd_godunov_fine_amr_host(C.byref(params), int(ilevel), len(igrid), _vp(igrid), _vp(tree.son),
                                                 _vp(tree.nbor), _vp(tree.father), tree.ngridmax, tree.ncoarse,
                                                 _vp(uold), _vp(unew), _vp(f), _vp(divu), _vp(enew), float(dx), float(dt),
                                                 int(nvector), int(interpol_var), int(interpol_type)))


def phi_fine_cg(tree, ilevel, igrid, phi, f, epsilon, itermax=10000, ordered=False, rho=None, rho_tot=0.0,
                fact=1.0, ncell_level=None):
    """Iteration loop of phi_fine_cg on one level of one rank: phi (ncell) and f (3, ncell) = (r, p, Ap)
    hold the state cmp_residual_cg left and are updated in place.  ordered: dot products added in
    the reference's order (bit-identical, slow).  Returns (iterations, error, error_ini, rhs_norm)."""
    igrid = _i32(igrid)
    if not (phi.dtype == np.float64 and phi.flags.c_contiguous and f.dtype == np.float64 and f.flags.c_contiguous
            and f.shape == (3, tree.ncell)):
        raise RamsesAmdError("phi (ncell) and f (3, ncell) must be C-contiguous float64")
    it = C.c_int(0)
    err = (C.c_double * 3)()
    check(lib().ramses_amd_cg_solve_host(int(ilevel), len(igrid), _vp(igrid), _vp(tree.son), _vp(tree.nbor), tree.ngridmax,
                                         tree.ncoarse, _vp(phi), _vp(f), _vp(rho), float(rho_tot), float(fact),
                                         float(8 * len(igrid) if ncell_level is None else ncell_level), float(epsilon),
                                         int(itermax), int(ordered), C.byref(it), err))
    return it.value, err[0], err[1], err[2]
