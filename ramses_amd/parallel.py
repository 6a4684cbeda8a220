"""Multi-GPU decomposition and the virtual-boundary halo exchange.

One process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI).  The
reference's Hilbert decomposition of a uniform level into 2^k ranks yields
axis-aligned bricks; `BrickDecomposition` reproduces that
partition (x fastest) and implements

    make_virtual_fine_dp      amr/virtual_boundaries.f90:373-528

GPU-resident: pack kernels gather the 2-cell (one oct) face slabs of ALL nvar
fields into one message per peer (the reference sends nvar separate rounds,
amr/amr_step.f90:503-510), grouped RCCL send/recv moves them, unpack kernels
scatter into the ghost octs.  Axes are exchanged x, y, z with slabs spanning
the already-filled ghosts, which also fills edge and corner octs (the
reference's reception lists contain those octs explicitly).

There is no reverse (accumulating) exchange on a single fully refined level:
make_virtual_reverse_dp only carries coarse-fine flux corrections, which are
zero here because set_unew zeroes the reception cells
(hydro/godunov_fine.f90:92-104).
"""
import ctypes as C

import torch

from ._capi import check, lib
from .hydro import HydroLevel, _ptr, _stream
from .transport import DistTransport


def rank_coords(rank, pgrid):
    px, py, pz = pgrid
    return (rank % px, (rank // px) % py, rank // (px * py))


def coords_rank(c, pgrid):
    px, py, pz = pgrid
    return (c[0] % px) + px * ((c[1] % py) + py * (c[2] % pz))


class BrickDecomposition:
    """pgrid=(px,py,pz) ranks, each owning an n^3 brick (weak scaling) of the
    periodic (n*px, n*py, n*pz) level -- or, n = (nx, ny, nz), a brick of those
    extents (the parts of a cubic box on 2 or 4 ranks); boxlen is the x extent
    of the box."""

    def __init__(self, pgrid, rank, n, boxlen=0.5, ng=2, transport=None):
        self.pgrid = tuple(pgrid)
        self.rank = rank
        self.transport = transport if transport is not None else DistTransport()
        self.dims = (n, n, n) if isinstance(n, int) else tuple(int(v) for v in n)
        self.n = self.dims[0]
        self.ng = ng
        self.coords = rank_coords(rank, self.pgrid)
        self.boxlen = boxlen
        self.dx = boxlen / (self.dims[0] * self.pgrid[0])
        self.lo = tuple(c * d for c, d in zip(self.coords, self.dims))
        self._bufs = {}

    # -- construction ----------------------------------------------------------
    def make_level(self, params, poisson=False):
        nx, ny, nz = self.dims
        return HydroLevel(nx, ny, nz, self.dx, params=params, ng=self.ng, poisson=poisson)

    def init_sedov(self, lev, gamma=1.4):
        """namelist/sedov3d.nml on the global level, restricted to this brick
        (hydro/init_flow_fine.f90:455-596: the 'point' region deposits into the
        cell whose centre is within dx of the origin -- global cell (0,0,0))."""
        from . import ic
        # the level is a uniform background plus ONE cell (ic.sedov3d_corner_and_background): the reference's own values
        corner, back, _ = ic.sedov3d_corner_and_background(self.dims[0] * self.pgrid[0], boxlen=self.boxlen, gamma=gamma)
        u = lev.interior(lev.uold)
        for v in range(5):
            u[v].fill_(float(back[v]))
            if self.lo == (0, 0, 0):
                u[v, 0, 0, 0] = float(corner[v])

    def neighbour(self, axis, direction):
        c = list(self.coords)
        c[axis] += direction
        return coords_rank(c, self.pgrid)

    # -- slab movers (C ABI; overridable in CPU protocol tests) ------------------
    def _slab_size(self, lev, nvar, face):
        n = lib().ramses_amd_halo_slab_size(C.byref(lev.brick), nvar, face)
        if n < 0:
            check(int(n))
        return int(n)

    def _pack(self, lev, t, nvar, face, buf):
        check(lib().ramses_amd_halo_pack(C.byref(lev.brick), _ptr(t), nvar, face, _ptr(buf), _stream()))

    def _unpack(self, lev, t, nvar, face, buf):
        check(lib().ramses_amd_halo_unpack(C.byref(lev.brick), _ptr(t), nvar, face, _ptr(buf), _stream()))

    def _fill_periodic(self, lev, t, nvar, axes):
        check(lib().ramses_amd_fill_ghosts_periodic(C.byref(lev.brick), _ptr(t), nvar, axes, _stream()))

    def _buffers(self, lev, t, nvar, axis):
        key = (id(lev), nvar, axis)
        if key not in self._bufs:
            n = self._slab_size(lev, nvar, 2 * axis)
            mk = lambda: torch.empty(n, dtype=torch.float64, device=t.device)  # noqa: E731
            self._bufs[key] = (mk(), mk(), mk(), mk())
        return self._bufs[key]

    # -- the exchange ------------------------------------------------------------
    def exchange(self, lev, t, nvar):
        """Forward halo of the nvar-field brick tensor t (all fields fused)."""
        for axis in range(3):
            if self.pgrid[axis] == 1:
                self._fill_periodic(lev, t, nvar, 1 << axis)
                continue
            send_lo, send_hi, recv_lo, recv_hi = self._buffers(lev, t, nvar, axis)
            lo_nbr = self.neighbour(axis, -1)
            hi_nbr = self.neighbour(axis, +1)
            self._pack(lev, t, nvar, 2 * axis, send_lo)       # my low interior slab -> low neighbour's high ghosts
            self._pack(lev, t, nvar, 2 * axis + 1, send_hi)   # my high interior slab -> high neighbour's low ghosts
            # One grouped launch (ncclGroupStart/End).  With 2 ranks on the axis
            # both messages go to the same peer; sends and receives are posted in
            # matching order (peer's first send = its low slab = my high ghosts).
            self.transport.sendrecv([(send_lo, lo_nbr), (send_hi, hi_nbr)],
                                    [(recv_hi, hi_nbr), (recv_lo, lo_nbr)])
            self._unpack(lev, t, nvar, 2 * axis, recv_lo)
            self._unpack(lev, t, nvar, 2 * axis + 1, recv_hi)

    # -- one-shot exchange: all 26 neighbour regions at once ------------------------
    def _direct_plan(self, lev, nvar, t):
        """Per (level, nvar): the 26 regions ordered by peer, so that every peer gets ONE
        message.  Region g (an offset in {-1,0,1}^3) of MY ghost layers is filled from the
        neighbour at coords+g, which sends its interior region next to its side -g; both
        sides order a peer's regions by the sender's offset index."""
        key = ("direct", id(lev), nvar)
        if key in self._bufs:
            return self._bufs[key]
        import numpy as np
        ng = lev.ng
        n = (lev.nx, lev.ny, lev.nz)
        offs = [(ox, oy, oz) for oz in (-1, 0, 1) for oy in (-1, 0, 1) for ox in (-1, 0, 1) if (ox, oy, oz) != (0, 0, 0)]
        index = {o: i for i, o in enumerate(offs)}

        def send_box(o):      # interior cells next to side o
            org = [ng if o[d] <= 0 else ng + n[d] - ng for d in range(3)]
            ext = [n[d] if o[d] == 0 else ng for d in range(3)]
            return org, ext

        def recv_box(g):      # ghost cells on side g
            org = [0 if g[d] < 0 else (ng if g[d] == 0 else ng + n[d]) for d in range(3)]
            ext = [n[d] if g[d] == 0 else ng for d in range(3)]
            return org, ext

        def peer(o):
            return coords_rank([self.coords[d] + o[d] for d in range(3)], self.pgrid)

        sends = sorted(offs, key=lambda o: (peer(o), index[o]))
        recvs = sorted(offs, key=lambda g: (peer(g), index[tuple(-x for x in g)]))
        plan = {}
        for name, lst, box in (("send", sends, send_box), ("recv", recvs, recv_box)):
            boxes, offsets, segs = [], [], []
            pos = 0
            for o in lst:
                org, ext = box(o)
                boxes += org + ext
                offsets.append(pos)
                size = nvar * ext[0] * ext[1] * ext[2]
                q = peer(o)
                if segs and segs[-1][0] == q:
                    segs[-1][2] += size
                else:
                    segs.append([q, pos, size])
                pos += size
            plan[name] = dict(boxes=np.array(boxes, np.int32), offsets=np.array(offsets, np.int64), segs=segs,
                              buf=torch.empty(pos, dtype=torch.float64, device=t.device))
        self._bufs[key] = plan
        return plan

    def _multi(self, lev, t, nvar, part, pack):
        check(lib().ramses_amd_halo_multi(C.byref(lev.brick), _ptr(t), nvar, len(part["offsets"]),
                                          part["boxes"].ctypes.data_as(C.c_void_p),
                                          part["offsets"].ctypes.data_as(C.c_void_p), _ptr(part["buf"]),
                                          1 if pack else 0, _stream()))

    def exchange_direct(self, lev, t, nvar):
        """Forward halo in ONE round: one pack launch, one grouped send/recv with one
        message per peer (7 peers on a 2x2x2 node: every xGMI link carries its share at
        the same time), one unpack launch.  Same ghost values as exchange()."""
        plan = self._direct_plan(lev, nvar, t)
        S, R = plan["send"], plan["recv"]
        self._multi(lev, t, nvar, S, True)
        sends, recvs = [], []
        for (q, pos, size), (q2, pos2, size2) in zip(S["segs"], R["segs"]):
            assert q == q2 and size == size2
            if q == self.rank:
                R["buf"][pos2:pos2 + size2].copy_(S["buf"][pos:pos + size])   # periodic wrap onto myself
            else:
                sends.append((S["buf"][pos:pos + size], q))
                recvs.append((R["buf"][pos2:pos2 + size2], q))
        self.transport.sendrecv(sends, recvs)
        self._multi(lev, t, nvar, R, False)

    def exchange_direct_profiled(self, lev, t, nvar):
        """exchange_direct with an event after every stage on the current stream: returns
        (pack_ms, sendrecv_ms, unpack_ms, bytes sent to other ranks) -- bench.py's N>1 line
        explains its own efficiency with them."""
        plan = self._direct_plan(lev, nvar, t)
        S, R = plan["send"], plan["recv"]
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        ev[0].record()
        self._multi(lev, t, nvar, S, True)
        ev[1].record()
        sends, recvs, nbytes = [], [], 0
        self.last_bytes_per_peer = {}
        for (q, pos, size), (q2, pos2, size2) in zip(S["segs"], R["segs"]):
            if q == self.rank:
                R["buf"][pos2:pos2 + size2].copy_(S["buf"][pos:pos + size])
            else:
                sends.append((S["buf"][pos:pos + size], q))
                recvs.append((R["buf"][pos2:pos2 + size2], q))
                nbytes += 8 * size
                self.last_bytes_per_peer[int(q)] = self.last_bytes_per_peer.get(int(q), 0) + 8 * size
        self.transport.sendrecv(sends, recvs)
        ev[2].record()
        self._multi(lev, t, nvar, R, False)
        ev[3].record()
        ev[3].synchronize()
        return ev[0].elapsed_time(ev[1]), ev[1].elapsed_time(ev[2]), ev[2].elapsed_time(ev[3]), nbytes

    def make_virtual_fine_dp(self, lev, direct=True):
        """Refresh the ghost octs of uold (and of f when poisson) on every rank."""
        ex = self.exchange_direct if direct else self.exchange
        ex(lev, lev.uold, lev.nvar)
        if lev.f is not None:
            ex(lev, lev.f, 3)

    def step_overlapped(self, lev, dt):
        """One hydro step with the halo exchange of the NEW state hidden behind the
        interior sweep (the reference runs them back to back, amr/amr_step.f90:388-510):

            compute stream:  shell sweep | interior sweep ................ | swap
            comm stream:                 | pack, RCCL send/recv, unpack    |

        The shell launches produce every cell within 2 of a brick face, i.e. all the
        data the face slabs carry; the interior launch writes only cells the exchange
        never touches, and both read the old state, so the result equals
        godunov_fine -> set_uold -> make_virtual_fine_dp bit for bit."""
        if lev.uold.device.type != "cuda":
            # the schedule without streams (the CPU protocol tests, tests/test_halo_gloo.py): the same order of operations
            lev.godunov_fine_shell(dt)
            self.exchange_direct(lev, lev.unew, lev.nvar)
            lev.godunov_fine_interior(dt)
            lev.set_uold()
            return
        if getattr(self, "_comm_stream", None) is None:
            self._comm_stream = torch.cuda.Stream(device=lev.uold.device)
        comp = torch.cuda.current_stream()
        lev.godunov_fine_shell(dt)
        shell_done = torch.cuda.Event()
        shell_done.record(comp)
        with torch.cuda.stream(self._comm_stream):
            self._comm_stream.wait_event(shell_done)
            self.exchange_direct(lev, lev.unew, lev.nvar)  # ghosts of the new state
            comm_done = torch.cuda.Event()
            comm_done.record(self._comm_stream)
        lev.godunov_fine_interior(dt)
        comp.wait_event(comm_done)
        lev.set_uold()

    def allreduce_min(self, value, device):
        """dt = min over ranks (MPI_ALLREDUCE MIN of hydro/courant_fine.f90:140)."""
        return self.transport.allreduce(value, device, op="min")
