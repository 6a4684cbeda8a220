"""Host-side mirror of the reference's per-level hydro interface.

`HydroLevel` holds one fully refined level as a device-resident brick and
exposes the reference's routine names for this path with the same meaning:

    set_unew / godunov_fine / set_uold     hydro/godunov_fine.f90:5-232
    courant_fine                           hydro/courant_fine.f90:1-159
    make_virtual_fine_dp                   amr/virtual_boundaries.f90:373-528

PyTorch is plumbing only (device memory, streams); every compute call goes
through the C ABI of libramses_amd.so and fails loudly if that is missing.
"""
import ctypes as C

import torch

from . import _capi
from ._capi import check, lib


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def godunov_tune(tile_rows=0, zchunk=0):
    """Tuning knobs of the sweep kernel (0 = built-in default); results do not
    depend on them."""
    check(lib().ramses_amd_godunov_tune(int(tile_rows), int(zchunk)))


class HydroLevel:
    """One fully refined level (or one rank's share of it) on one MI355X.

    uold/unew: torch.float64 tensors [nvar, nz+2ng, ny+2ng, nx+2ng] on the GPU,
    variable order rho, rho*u, rho*v, rho*w, E (hydro/condinit.f90:17-20).
    ng=0: periodic wrap in-kernel (single rank); ng=2: one ghost oct per side,
    filled by make_virtual_fine_dp (periodic self-copy or the RCCL exchange).
    """

    def __init__(self, nx, ny, nz, dx, params=None, ng=0, device="cuda", poisson=False, bound_type=None,
                 bound_state=None, no_inflow=False):
        """bound_type: {face: code} with the reference's &BOUNDARY_PARAMS codes
        (face 0:-x 1:+x 2:-y 3:+y 4:-z 5:+z; code face+1 reflexive, 10+face+1
        outflow, 20+face+1 imposed with bound_state[face] = conserved state);
        faces without an entry are periodic.  Needs ng >= 2.
        NDIM < 3 (params.ndim): ny and/or nz = 1, the brick keeps 5 variables."""
        if not torch.cuda.is_available():
            raise _capi.RamsesAmdError("HydroLevel needs a GPU (torch.cuda.is_available() is False); "
                                       "there is no CPU fallback")
        self.params = params if params is not None else _capi.make_params()
        self.nx, self.ny, self.nz, self.ng, self.dx = nx, ny, nz, ng, float(dx)
        self.nvar = self.params.nvar
        self.brick = _capi.dense_brick(nx, ny, nz, ng)
        shape = (self.nvar, nz + 2 * ng, ny + 2 * ng, nx + 2 * ng)
        self.device = torch.device(device)
        self.uold = torch.zeros(shape, dtype=torch.float64, device=self.device)
        self.unew = torch.zeros(shape, dtype=torch.float64, device=self.device)
        self.f = None
        if poisson:
            self.f = torch.zeros((3,) + shape[1:], dtype=torch.float64, device=self.device)
        self._red = torch.zeros(4, dtype=torch.float64, device=self.device)
        self.dtnew = 0.0
        self.bound_type = dict(bound_type or {})
        self.bound_state = dict(bound_state or {})
        self.no_inflow = bool(no_inflow)
        if self.bound_type and ng < 2:
            raise _capi.RamsesAmdError("physical boundaries need ghost layers (ng >= 2)")
        for face in (0, 2, 4):
            if (face in self.bound_type) != (face + 1 in self.bound_type):
                raise _capi.RamsesAmdError("a direction is either periodic or has a boundary on both faces")
        self._periodic_axes = sum(1 << a for a in range(3) if 2 * a not in self.bound_type)

    # -- views ---------------------------------------------------------------
    def interior(self, t=None):
        t = self.uold if t is None else t
        g = self.ng
        if g == 0:
            return t
        return t[:, g:g + self.nz, g:g + self.ny, g:g + self.nx]

    def upload(self, u_host):
        """u_host: [nvar,nz,ny,nx] array-like of the interior state."""
        src = torch.as_tensor(u_host, dtype=torch.float64)
        self.interior(self.uold).copy_(src.to(self.device))

    def download(self, t=None):
        return self.interior(t).cpu().numpy()

    # -- the reference's call surface ------------------------------------------
    def make_virtual_fine_dp(self):
        """Single-rank periodic limit of the forward halo exchange: refresh the
        ghost octs of uold from the opposite interior face (no-op for ng=0)."""
        if self.ng:
            for axis in range(3):
                if self._periodic_axes & (1 << axis):
                    check(lib().ramses_amd_fill_ghosts_periodic(C.byref(self.brick), _ptr(self.uold),
                                                                self.nvar, 1 << axis, _stream()))
                    if self.f is not None:
                        check(lib().ramses_amd_fill_ghosts_periodic(C.byref(self.brick), _ptr(self.f),
                                                                    3, 1 << axis, _stream()))
                else:
                    self.make_boundary_hydro(axis)

    def make_boundary_hydro(self, axis):
        """make_boundary_hydro (hydro/hydro_boundary.f90:5-269) for the two faces of one direction."""
        for face in (2 * axis, 2 * axis + 1):
            st = self.bound_state.get(face)
            buf = (C.c_double * self.nvar)(*st) if st is not None else None
            check(lib().ramses_amd_make_boundary_hydro(C.byref(self.params), C.byref(self.brick), _ptr(self.uold), face,
                                                       int(self.bound_type[face]), buf, int(self.no_inflow), _stream()))

    def courant_fine(self):
        """CFL time step of the level -> (dt, mass, etot, eint); also sets dtnew."""
        check(lib().ramses_amd_courant_init(C.byref(self.params), self.dx, _ptr(self._red), _stream()))
        check(lib().ramses_amd_courant_brick(C.byref(self.params), C.byref(self.brick), _ptr(self.uold),
                                             _ptr(self.f), self.dx, _ptr(self._red), _stream()))
        dt, mass, etot, eint = self._red.cpu().tolist()
        self.dtnew = dt
        return dt, mass, etot, eint

    def godunov_fine(self, dt=None):
        """set_unew + godunov_fine fused: unew = uold + flux differences."""
        dt = self.dtnew if dt is None else dt
        check(lib().ramses_amd_godunov_brick(C.byref(self.params), C.byref(self.brick), _ptr(self.uold),
                                             _ptr(self.f), _ptr(self.unew), self.dx, float(dt), _stream()))

    def godunov_fine_shell(self, dt=None):
        """The part of godunov_fine that produces the cells the neighbour ranks
        receive (tiles/planes touching a brick face); see godunov_fine_interior."""
        dt = self.dtnew if dt is None else dt
        check(lib().ramses_amd_godunov_brick_shell(C.byref(self.params), C.byref(self.brick), _ptr(self.uold),
                                                   _ptr(self.f), _ptr(self.unew), self.dx, float(dt), _stream()))

    def godunov_fine_interior(self, dt=None):
        """The rest of godunov_fine: shell + interior == godunov_fine bit for bit."""
        dt = self.dtnew if dt is None else dt
        check(lib().ramses_amd_godunov_brick_interior(C.byref(self.params), C.byref(self.brick), _ptr(self.uold),
                                                      _ptr(self.f), _ptr(self.unew), self.dx, float(dt), _stream()))

    def set_uold(self):
        """uold = unew (hydro/godunov_fine.f90:193-197): a buffer swap on the device."""
        self.uold, self.unew = self.unew, self.uold

    def step(self, dt=None):
        """One fine step of amr_step's hydro branch on a single level:
        godunov_fine -> set_uold -> make_virtual_fine_dp (amr/amr_step.f90:388-510)."""
        self.godunov_fine(dt)
        self.set_uold()
        self.make_virtual_fine_dp()
