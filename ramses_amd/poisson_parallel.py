"""Multi-GPU multigrid on a periodic, fully refined level: the distributed form
of

    multigrid_fine(ilevel,icount)   poisson/multigrid_fine_commons.f90:25-296
    recursive_multigrid_coarse      :307-390
    force_fine / gradient_phi       poisson/force_fine.f90:5-324

One rank (GPU) owns an n^3 brick of the (n*p)^3 level, p^3 ranks.  MI355X-first
choices:

* every multigrid level of a rank is a brick with NG = 5 ghost layers.  The
  fused smoother recomputes the neighbours' updates inside the ghost layers
  (the same cone it already recomputes in its tile halos), so ONE 5-cell-wide
  exchange per smoother call replaces the reference's exchange after every
  colour pass (make_virtual_mg_dp, 8 per V-cycle and level): fewer, larger
  messages for the point-to-point xGMI links;
* restriction is local (octs never straddle ranks); prolongation needs one
  ghost layer of the coarse correction;
* levels whose per-rank brick would fall below the smoother's 64-cell tile
  are REPLICATED: one all-gather of the restricted residual, then every rank
  runs the remaining V-cycle on the whole coarse level (the single-GPU code)
  and reads its part of the correction -- no latency-bound tiny exchanges;
* scalar reductions: the residual norms (sum) per iteration.

All arithmetic runs in the HIP kernels through the C ABI, in the reference's
operation order, so phi is bit-identical to the single-GPU solve (and to the
reference) whenever the iteration counts agree.
"""
import ctypes as C
import math

import torch

from . import _capi
from ._capi import check, lib
from .hydro import _ptr, _stream
from .parallel import BrickDecomposition, rank_coords
from .poisson import TWOPI

NG = 5                 # ghost layers: 4 colour passes + the residual's stencil
MIN_FUSED = 64         # the fused smoother's tile width
MAXITER = 10           # multigrid_fine_commons.f90:34
SAFE_FACTOR = 0.5      # :35
MG_MAX_PARTIALS = 4096


class _Lev:
    """One multigrid level of this rank: local n^3 cells + NG ghost layers."""

    def __init__(self, l, n, dev):
        self.l, self.n = l, n
        self.nx = self.ny = self.nz = n
        self.brick = _capi.dense_brick(n, n, n, NG)
        p = n + 2 * NG
        z = lambda: torch.zeros(p, p, p, dtype=torch.float64, device=dev)  # noqa: E731
        self.u1, self.u2, self.u3, self.u4 = z(), z(), z(), z()
        self.ng = NG

    def interior(self, t):
        g, n = NG, self.n
        return t[g:g + n, g:g + n, g:g + n]


class PoissonDecomposition:
    def __init__(self, pgrid, rank, n, boxlen=1.0, epsilon=1e-4, transport=None, device="cuda"):
        if not torch.cuda.is_available():
            raise _capi.RamsesAmdError("PoissonDecomposition needs a GPU; there is no CPU fallback")
        px, py, pz = pgrid
        if not (px == py == pz) or px & (px - 1):
            raise _capi.RamsesAmdError("distributed multigrid needs a cubic power-of-two rank grid (got %r)" % (pgrid,))
        if n < MIN_FUSED or n & (n - 1):
            raise _capi.RamsesAmdError("per-rank brick must be a power of two >= %d (got %d)" % (MIN_FUSED, n))
        self.p = px
        self.n = n
        self.level = int(round(math.log2(n * px)))
        self.boxlen, self.epsilon = boxlen, epsilon
        self.fourpi = 2 * TWOPI * boxlen
        self.dec = BrickDecomposition(pgrid, rank, n, boxlen=boxlen, ng=NG, transport=transport)
        self.tr = self.dec.transport
        self.coords = rank_coords(rank, pgrid)
        dev = torch.device(device)
        self.dev = dev
        # distributed levels: local size >= MIN_FUSED
        self.lev = {}
        l, nl = self.level, n
        while nl >= MIN_FUSED and l >= 1:
            self.lev[l] = _Lev(l, nl, dev)
            l, nl = l - 1, nl // 2
        self.lrep = l                          # first replicated level (0: none)
        self.nrep_local = nl
        if self.lrep >= 1:
            self.rep_local = _Lev(self.lrep, nl, dev)   # target of the last distributed restriction
            ng_ = 1 << self.lrep
            self.rep_rhs = torch.zeros(ng_, ng_, ng_, dtype=torch.float64, device=dev)
            self.rep_u1 = torch.zeros_like(self.rep_rhs)
            nwork = lib().ramses_amd_mg_workspace_doubles(self.lrep + 1)
            if nwork < 0:
                check(int(nwork))
            self.rep_work = torch.zeros(int(nwork), dtype=torch.float64, device=dev)
        self._work = torch.zeros(MG_MAX_PARTIALS + 8, dtype=torch.float64, device=dev)
        self._norm = torch.zeros(2, dtype=torch.float64, device=dev)
        self._origin = (C.c_int * 3)()
        fine = self.lev[self.level]
        self.phi = fine.u1                     # with ghosts; interior via phi_interior()
        self.rho = torch.zeros(n, n, n, dtype=torch.float64, device=dev)
        self.f = torch.zeros(3, n, n, n, dtype=torch.float64, device=dev)
        self.safe_mode = 0
        self.last_iters, self.last_err = 0, 0.0
        self.exchanges = 0

    # ------------------------------------------------------------------ helpers
    def phi_interior(self):
        return self.lev[self.level].interior(self.lev[self.level].u1)

    def _exchange(self, L, t):
        self.dec.exchange_direct(L, t, 1)      # one round, one message per peer
        self.exchanges += 1

    def _fused(self, L, src, dst, rhs, res, norm_slot):
        dx = 2.0 ** (-L.l)
        check(lib().ramses_amd_mg_smooth_fused_ghost(
            _ptr(src), _ptr(dst), _ptr(rhs), _ptr(res) if res is not None else None, _ptr(self._work),
            C.c_void_p(self._norm.data_ptr() + 8 * norm_slot) if norm_slot is not None else None,
            L.n, NG, dx, 4, _stream()))

    def _restrict(self, Lf, res, Lc, rhs_c):
        check(lib().ramses_amd_mg_restrict_ghost(_ptr(res), _ptr(rhs_c), Lf.n, NG, NG, _stream()))

    def _interp_from(self, Lf, phi_f, l_coarse):
        """phi_f += prolongation of the correction of level l_coarse."""
        if l_coarse in self.lev:
            Lc = self.lev[l_coarse]
            self._exchange(Lc, Lc.u1)          # one ghost layer is needed; the slab mover sends all NG
            check(lib().ramses_amd_mg_interp_correct_ghost(_ptr(phi_f), Lf.n, NG, _ptr(Lc.u1), NG, 0, None, _stream()))
        else:
            for d in range(3):
                self._origin[d] = self.coords[d] * self.nrep_local
            check(lib().ramses_amd_mg_interp_correct_ghost(_ptr(phi_f), Lf.n, NG, _ptr(self.rep_u1), 0,
                                                           1 << self.lrep, self._origin, _stream()))

    def _coarse_cycle(self, l, safe):
        """recursive_multigrid_coarse (multigrid_fine_commons.f90:307-390); on entry
        the restricted residual is in the level's u2 interior and u1 is zero."""
        if l < 1:
            return
        if l not in self.lev:
            # replicated levels: gather the right-hand side, solve everywhere
            local = self.rep_local.interior(self.rep_local.u2).contiguous()
            parts = self.tr.allgather(local)                     # [world, nl, nl, nl], rank = x + p*(y + p*z)
            p, nl = self.p, self.nrep_local
            g = parts.view(p, p, p, nl, nl, nl).permute(0, 3, 1, 4, 2, 5)   # [pz, k, py, j, px, i]
            self.rep_rhs.copy_(g.reshape(p * nl, p * nl, p * nl))
            check(lib().ramses_amd_mg_coarse_solve_dense(l, _ptr(self.rep_rhs), _ptr(self.rep_u1), _ptr(self.rep_work),
                                                         safe, _stream()))
            return
        L = self.lev[l]
        self._exchange(L, L.u2)
        self._fused(L, L.u1, L.u4, L.u2, L.u3, None)              # pre-smoothing + residual
        self._restrict_to(l, L.u3)
        self._coarse_cycle(l - 1, safe)
        if l - 1 >= 1:
            self._interp_from(L, L.u4, l - 1)
        self._exchange(L, L.u4)
        check(lib().ramses_amd_mg_smooth_fused_ghost(_ptr(L.u4), _ptr(L.u1), _ptr(L.u2), None, None, None,
                                                     L.n, NG, 2.0 ** (-l), 4, _stream()))   # post-smoothing

    def _restrict_to(self, l, res):
        """restrict the residual of level l into level l-1 (u2) and zero its correction."""
        if l - 1 < 1:
            return
        Lf = self.lev[l]
        if (l - 1) in self.lev:
            Lc = self.lev[l - 1]
            Lc.u1.zero_()
            self._restrict(Lf, res, Lc, Lc.u2)
        else:
            self._restrict(Lf, res, self.rep_local, self.rep_local.u2)

    # ------------------------------------------------------------------ API
    def multigrid_fine(self, rho_tot):
        """Solve for phi of the level from a zero first guess.  rho (this rank's
        interior) is self.rho; rho_tot the mean density of the whole box."""
        L = self.lev[self.level]
        rho_tot = float(rho_tot)
        phi, phi2, f1, f2 = L.u1, L.u4, L.u3, L.u2
        phi.zero_()
        rhs = torch.empty_like(self.rho)
        check(lib().ramses_amd_mg_rhs(_ptr(self.rho), _ptr(rhs), self.rho.numel(), self.fourpi, rho_tot, _stream()))
        L.interior(f2).copy_(rhs)
        self._exchange(L, f2)
        it, err, i_res_norm2 = 0, 1.0, 0.0
        safe = self.safe_mode
        while True:
            it += 1
            if it > 1:
                self._exchange(L, phi)
            self._fused(L, phi, phi2, f2, f1, 0 if it == 1 else None)
            if it == 1:
                i_res_norm2 = self.tr.allreduce(float(self._norm[0].item()), self.dev)
            if self.level > 1:
                self._restrict_to(self.level, f1)
                self._coarse_cycle(self.level - 1, safe)
                self._interp_from(L, phi2, self.level - 1)
            self._exchange(L, phi2)
            # post-smoothing; only the norm of the residual is needed
            self._fused(L, phi2, phi, f2, None, 1)
            res_norm2 = self.tr.allreduce(float(self._norm[1].item()), self.dev)
            last_err = err
            err = math.sqrt(res_norm2 / (i_res_norm2 + 1e-20 * rho_tot * rho_tot))
            if err < self.epsilon or it >= MAXITER:
                break
            if err > last_err * SAFE_FACTOR and not safe:
                safe = 1
        self.safe_mode = safe
        self.last_iters, self.last_err = it, err
        return it, err

    def force_fine(self):
        L = self.lev[self.level]
        self._exchange(L, L.u1)
        check(lib().ramses_amd_gradient_phi_ghost(_ptr(L.u1), _ptr(self.f), L.n, NG, 2.0 ** (-self.level), _stream()))
