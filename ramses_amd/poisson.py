"""Host-side mirror of the reference's per-level gravity interface on a fully
refined periodic level:

    multigrid_fine(ilevel,icount)   poisson/multigrid_fine_commons.f90:25-296
    force_fine(ilevel,icount)       poisson/force_fine.f90:5-194 (gradient_phi)

Device-resident; every compute call goes through the C ABI.
"""
import ctypes as C

import torch

from . import _capi
from ._capi import check, lib
from .hydro import _ptr, _stream

TWOPI = 6.2831853  # amr/constants.f90:5 -- the reference's (truncated) value, kept on purpose


class PoissonLevel:
    """phi, rho, f(:,1:3) of one fully refined periodic level (n = 2^level)."""

    def __init__(self, level, boxlen=1.0, epsilon=1e-4, device="cuda"):
        if not torch.cuda.is_available():
            raise _capi.RamsesAmdError("PoissonLevel needs a GPU; there is no CPU fallback")
        self.level = level
        self.n = n = 2 ** level
        self.boxlen = boxlen
        self.epsilon = epsilon
        self.fourpi = 2 * TWOPI * boxlen          # 2*twopi*scale, nx_loc = 1
        dev = torch.device(device)
        z = lambda *s: torch.zeros(s, dtype=torch.float64, device=dev)  # noqa: E731
        self.phi = z(n, n, n)
        self.rho = z(n, n, n)
        self.f = z(3, n, n, n)
        self._f1 = z(n, n, n)
        self._f2 = z(n, n, n)
        nwork = lib().ramses_amd_mg_workspace_doubles(level)
        if nwork < 0:
            check(int(nwork))
        self._work = z(int(nwork))
        self.safe_mode = C.c_int(0)
        self.rho_tot = 0.0
        self.last_iters = 0
        self.last_err = 0.0

    def multigrid_fine(self, rho_tot=None):
        """Solve on the level from a zero first guess (make_multipole_phi at levelmin)."""
        self.rho_tot = float(self.rho.mean().item()) if rho_tot is None else float(rho_tot)
        self.phi.zero_()
        it, err = C.c_int(), C.c_double()
        check(lib().ramses_amd_multigrid_fine_brick(self.level, _ptr(self.rho), self.rho_tot, self.fourpi,
                                                    self.epsilon, C.byref(self.safe_mode), _ptr(self.phi),
                                                    _ptr(self._f1), _ptr(self._f2), _ptr(self._work),
                                                    C.byref(it), C.byref(err), _stream()))
        self.last_iters, self.last_err = it.value, err.value
        return it.value, err.value

    def force_fine(self):
        check(lib().ramses_amd_gradient_phi_brick(self.level, _ptr(self.phi), _ptr(self.f), _stream()))
