"""Host-side mirror of the MHD sweep (SOLVER=mhd of the reference).

`MhdLevel` holds one fully refined periodic level as a device brick of the reference's eleven
fields -- uold(:,1:5) = rho, rho u, rho v, rho w, E; uold(:,6:8) = the field on the cell's LEFT
faces; uold(:,nvar+1:nvar+3) = the field on its RIGHT faces (mhd/godunov_fine.f90:40-110) -- and
exposes godunov_fine / set_uold with the reference's meaning.  Every compute call goes through
the C ABI (ramses_amd_mhd_godunov_brick); there is no CPU fallback.
"""
import ctypes as C

import torch

from . import _capi
from ._capi import check, lib

RIEMANN = {"llf": 0, "roe": 1, "hll": 2, "hlld": 3, "upwind": 4, "hydro": 5}       # hydro/read_hydro_params.f90:184-199
RIEMANN2D = {"llf": 0, "roe": 1, "upwind": 2, "hll": 3, "hlla": 4, "hlld": 5}     # :205-220


class MhdParams(C.Structure):
    """struct ramses_amd_mhd_params"""
    _fields_ = [("gamma", C.c_double), ("smallr", C.c_double), ("smallc", C.c_double), ("slope_theta", C.c_double),
                ("slope_type", C.c_int32), ("slope_mag_type", C.c_int32), ("riemann", C.c_int32), ("riemann2d", C.c_int32)]


def make_mhd_params(gamma=1.4, smallr=1e-10, smallc=1e-10, slope_type=1, slope_mag_type=-1, slope_theta=1.5, riemann="llf",
                    riemann2d="llf"):
    """the defaults of mhd/hydro_parameters.f90:75-112"""
    return MhdParams(gamma, smallr, smallc, slope_theta, slope_type, slope_mag_type,
                     RIEMANN[riemann] if isinstance(riemann, str) else int(riemann),
                     RIEMANN2D[riemann2d] if isinstance(riemann2d, str) else int(riemann2d))


class MhdLevel:
    def __init__(self, nx, ny, nz, dx, params=None, device="cuda"):
        if not torch.cuda.is_available():
            raise _capi.RamsesAmdError("MhdLevel needs a GPU (torch.cuda.is_available() is False); there is no CPU fallback")
        self.params = params if params is not None else make_mhd_params()
        self.nx, self.ny, self.nz, self.dx = nx, ny, nz, float(dx)
        self.device = torch.device(device)
        self.uold = torch.zeros((11, nz, ny, nx), dtype=torch.float64, device=self.device)
        self.unew = torch.zeros_like(self.uold)
        nbytes = lib().ramses_amd_mhd_workspace_bytes(nx, ny, nz)
        if nbytes < 0:
            check(int(nbytes))
        self._work = torch.empty(int(nbytes), dtype=torch.uint8, device=self.device)

    def upload(self, u):
        self.uold.copy_(torch.as_tensor(u, dtype=torch.float64))

    def download(self):
        return self.uold.cpu().numpy()

    def godunov_fine(self, dt):
        """set_unew + godunov_fine: unew = uold advanced by dt"""
        check(lib().ramses_amd_mhd_godunov_brick(C.byref(self.params), self.nx, self.ny, self.nz, C.c_void_p(self.uold.data_ptr()),
                                                 C.c_void_p(self.unew.data_ptr()), self.dx, float(dt), C.c_void_p(self._work.data_ptr()),
                                                 self._work.numel(), C.c_void_p(torch.cuda.current_stream().cuda_stream)))

    def set_uold(self):
        self.uold, self.unew = self.unew, self.uold

    def step(self, dt):
        self.godunov_fine(dt)
        self.set_uold()
