"""Multi-GPU multigrid on a periodic, fully refined level: the distributed form
of

    multigrid_fine(ilevel,icount)   poisson/multigrid_fine_commons.f90:25-296
    recursive_multigrid_coarse      :307-390
    force_fine / gradient_phi       poisson/force_fine.f90:5-324

The level is the reference's cubic periodic box (nx = ny = nz = 1 coarse cell,
amr/amr_parameters.f90:81: not a namelist item), N = 2^level cells per
direction, cut into px x py x pz bricks, each a power of two -- the shapes the
reference's Hilbert decomposition gives 2^k ranks on a uniform level (2 ranks:
two half boxes, 4 ranks: four quarter columns, 8 ranks: the octants).  One rank
(GPU) owns an (N/px) x (N/py) x (N/pz) brick.  MI355X-first choices (DESIGN.md
section 6):

* every multigrid level of a rank is a brick with NG = 5 ghost layers.  The
  fused smoother recomputes the neighbours' updates inside the ghost layers
  (the same cone it already recomputes in its tile halos), so ONE 5-cell-wide
  exchange per smoother call replaces the reference's exchange after every
  colour pass (make_virtual_mg_dp, 8 per V-cycle and level): fewer, larger
  messages for the point-to-point xGMI links;
* restriction is local (octs never straddle ranks); prolongation needs one
  ghost layer of the coarse correction;
* levels whose per-rank brick would fall below the smoother's 64-cell tile
  in any direction are REPLICATED: one all-gather of the restricted residual, then every rank
  runs the remaining V-cycle on the whole coarse level (the single-GPU code)
  and reads its part of the correction -- no latency-bound tiny exchanges;
* scalar reductions: the residual norms (sum) per iteration.

All arithmetic runs in the HIP kernels through the C ABI, in the reference's
operation order, so phi is bit-identical to the single-GPU solve (and to the
reference) whenever the iteration counts agree.
"""
import ctypes as C
import math

import numpy as np
import torch

from . import _capi
from ._capi import check, lib
from .hydro import _ptr, _stream
from .parallel import rank_coords
from .poisson import TWOPI
from .transport import DistTransport, RcclTransport

NG = 5                 # ghost layers: 4 colour passes + the residual's stencil
MIN_FUSED = 64         # the fused smoother's tile width


def brick_dims(level, pgrid):
    """extents (x, y, z) of one rank's brick of the 2^level cubic box"""
    N = 1 << level
    return tuple(N // p for p in pgrid)


def assemble_level(parts, pgrid, dims):
    """parts[rank] = the [nz][ny][nx] brick of rank = x + px*(y + py*z) (an all-gather in rank order);
    returns the whole [pz*nz][py*ny][px*nx] level (what the library's assemble kernel builds)."""
    (px, py, pz), (nx, ny, nz) = pgrid, dims
    g = parts.view(pz, py, px, nz, ny, nx).permute(0, 3, 1, 4, 2, 5)   # [pz, k, py, j, px, i]
    return g.reshape(pz * nz, py * ny, px * nx)


# ---- include/ramses_amd.h: ramses_amd_mg_transport ---------------------------------------------------------------
_I64P, _IP, _DP = C.POINTER(C.c_int64), C.POINTER(C.c_int), C.POINTER(C.c_double)
_EXCHANGE = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, _IP, _DP, _I64P, _I64P, _DP, _I64P, _I64P)
_ALLGATHER = C.CFUNCTYPE(C.c_int, C.c_void_p, _DP, C.c_int64, _DP)
_ALLREDUCE = C.CFUNCTYPE(C.c_int, C.c_void_p, _DP)


class MgTransport(C.Structure):
    _fields_ = [("user", C.c_void_p), ("exchange", _EXCHANGE), ("allgather", _ALLGATHER), ("allreduce_sum", _ALLREDUCE)]


def _host(ptr, off, cnt):
    """torch view of cnt doubles of a host buffer"""
    return torch.from_numpy(np.ctypeslib.as_array(C.cast(C.addressof(ptr.contents) + 8 * off, _DP), shape=(cnt,)))


class _Callbacks:
    """The caller's message layer behind the library's three host-buffer callbacks: a transport of
    ramses_amd/transport.py (torch.distributed gloo / nccl, or the virtual ranks of LocalWorld)."""

    def __init__(self, tr, device):
        self.tr, self.device = tr, device
        self.error = None
        # torch.distributed with a device backend moves device tensors only
        self.on_device = isinstance(tr, DistTransport) and not tr.staged and tr.world > 1
        self.table = MgTransport(None, _EXCHANGE(self._exchange), _ALLGATHER(self._allgather), _ALLREDUCE(self._allreduce))

    def _guard(self, fn):
        try:
            fn()
            return 0
        except BaseException as exc:      # noqa: BLE001  (an exception must not cross the C frames)
            self.error = exc
            return 1

    def _exchange(self, user, npeer, peer, h_send, soff, scnt, h_recv, roff, rcnt):
        def go():
            sends = [(_host(h_send, soff[i], scnt[i]), peer[i]) for i in range(npeer)]
            recvs = [(_host(h_recv, roff[i], rcnt[i]), peer[i]) for i in range(npeer)]
            if self.on_device:
                dsend = [(t.to(self.device), q) for t, q in sends]
                drecv = [(torch.empty(t.shape, dtype=t.dtype, device=self.device), q) for t, q in recvs]
                self.tr.sendrecv(dsend, drecv)
                for (t, _), (d, _) in zip(recvs, drecv):
                    t.copy_(d)
            else:
                self.tr.sendrecv(sends, recvs)
        return self._guard(go)

    def _allgather(self, user, h_send, count, h_recv):
        def go():
            src = _host(h_send, 0, count)
            out = self.tr.allgather(src.to(self.device) if self.on_device else src)
            _host(h_recv, 0, count * self.tr.world).copy_(out.reshape(-1))
        return self._guard(go)

    def _allreduce(self, user, value):
        def go():
            value[0] = self.tr.allreduce(float(value[0]), self.device)
        return self._guard(go)


class PoissonDecomposition:
    """Host mirror of ramses_amd_mgdist_* (csrc/mg_dist.hip): the V-cycle driver and its halo exchanges run behind
    the C ABI; this class owns the rank's density / force arrays and the transport."""

    def __init__(self, pgrid, rank, n=None, boxlen=1.0, epsilon=1e-4, transport=None, device="cuda", level=None):
        """pgrid = (px, py, pz) ranks (powers of two) on the 2^level box; `n` (kept for cubic rank grids):
        the brick's extent, level = log2(n * px)."""
        if not torch.cuda.is_available():
            raise _capi.RamsesAmdError("PoissonDecomposition needs a GPU; there is no CPU fallback")
        pgrid = tuple(int(p) for p in pgrid)
        if any(p < 1 or p & (p - 1) for p in pgrid):
            raise _capi.RamsesAmdError("distributed multigrid needs a power-of-two rank grid (got %r)" % (pgrid,))
        if level is None:
            if n is None or not (pgrid[0] == pgrid[1] == pgrid[2]) or n & (n - 1):
                raise _capi.RamsesAmdError("give the level of the box (a brick extent n only names it on a cubic rank grid)")
            level = int(round(math.log2(n * pgrid[0])))
        self.pgrid, self.level = pgrid, level
        self.boxlen, self.epsilon = boxlen, epsilon
        self.fourpi = 2 * TWOPI * boxlen
        self.tr = transport if transport is not None else DistTransport()
        self.dev = torch.device(device)
        # RCCL inside the library when the transport is the library's own communicator, the callbacks otherwise
        self._cb = None if isinstance(self.tr, RcclTransport) else _Callbacks(self.tr, self.dev)
        ctx = C.c_void_p()
        pg = (C.c_int * 3)(*pgrid)
        check(lib().ramses_amd_mgdist_create(level, pg, rank, None, C.byref(self._cb.table) if self._cb else None, C.byref(ctx)))
        self._ctx = ctx
        dims, coords = (C.c_int * 3)(), (C.c_int * 3)()
        nlev, lrep = C.c_int(), C.c_int()
        check(lib().ramses_amd_mgdist_info(ctx, dims, coords, C.byref(nlev), C.byref(lrep), None, None))
        self.dims, self.coords = tuple(dims), tuple(coords)
        assert self.dims == brick_dims(level, pgrid) and self.coords == rank_coords(rank, pgrid)
        self.nlev, self.lrep = nlev.value, lrep.value
        nx, ny, nz = self.dims
        self.rho = torch.zeros(nz, ny, nx, dtype=torch.float64, device=self.dev)
        self.phi = torch.zeros(nz, ny, nx, dtype=torch.float64, device=self.dev)
        self.f = torch.zeros(3, nz, ny, nx, dtype=torch.float64, device=self.dev)
        self.last_iters, self.last_err = 0, 0.0
        self._exch0 = 0

    def __del__(self):
        ctx = getattr(self, "_ctx", None)
        if ctx:
            try:
                lib().ramses_amd_mgdist_destroy(ctx)
            except Exception:       # noqa: BLE001  (interpreter shutdown)
                pass
            self._ctx = None

    def _call(self, rc):
        if rc and self._cb is not None and self._cb.error is not None:
            exc, self._cb.error = self._cb.error, None
            raise exc
        check(rc)

    def my_slices(self):
        """this rank's part of a [z][y][x] array of the whole level"""
        nx, ny, nz = self.dims
        cx, cy, cz = self.coords
        return (slice(cz * nz, (cz + 1) * nz), slice(cy * ny, (cy + 1) * ny), slice(cx * nx, (cx + 1) * nx))

    @property
    def exchanges(self):
        n = C.c_int64()
        check(lib().ramses_amd_mgdist_info(self._ctx, None, None, None, None, None, C.byref(n)))
        return n.value - self._exch0

    @exchanges.setter
    def exchanges(self, v):
        n = C.c_int64()
        check(lib().ramses_amd_mgdist_info(self._ctx, None, None, None, None, None, C.byref(n)))
        self._exch0 = n.value - int(v)

    @property
    def safe_mode(self):
        v = C.c_int()
        check(lib().ramses_amd_mgdist_info(self._ctx, None, None, None, None, C.byref(v), None))
        return v.value

    def phi_interior(self):
        self._call(lib().ramses_amd_mgdist_get_phi(self._ctx, _ptr(self.phi), _stream()))
        return self.phi

    # ------------------------------------------------------------------ API
    def multigrid_fine(self, rho_tot):
        """Solve for phi of the level from a zero first guess.  rho (this rank's
        brick) is self.rho; rho_tot the mean density of the whole box."""
        it, err = C.c_int(), C.c_double()
        self._call(lib().ramses_amd_mgdist_solve(self._ctx, _ptr(self.rho), float(rho_tot), self.fourpi, self.epsilon,
                                                 C.byref(it), C.byref(err), _stream()))
        self.last_iters, self.last_err = it.value, err.value
        return it.value, err.value

    def force_fine(self):
        self._call(lib().ramses_amd_mgdist_force(self._ctx, _ptr(self.f), _stream()))
