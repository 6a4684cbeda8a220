"""Multi-rank halo exchange protocol on CPU (gloo): the message pattern of
BrickDecomposition.exchange (who sends which face slab to whom, in which
order, including the 2-ranks-per-axis case where both faces go to one peer)
with the HIP slab movers replaced by an independent torch-slicing mover that
lives in this test.  The HIP movers themselves are checked against the same
slicing code in test_halo_gpu.py."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ramses_amd.parallel import BrickDecomposition, rank_coords


def slab_slices(n, ng, face, ghost):
    """Python restatement of the slab geometry of include/ramses_amd.h:
    returns (z,y,x) slices in allocated coordinates."""
    axis, hi = face // 2, face & 1
    sl = []
    for d in range(3):
        if d < axis:
            sl.append(slice(0, n[d] + 2 * ng))
        elif d > axis:
            sl.append(slice(ng, ng + n[d]))
        else:
            if ghost:
                o = n[d] + ng if hi else 0
            else:
                o = n[d] if hi else ng
            sl.append(slice(o, o + ng))
    return (sl[2], sl[1], sl[0])


class FakeLevel:
    def __init__(self, n, ng, nvar):
        self.nx = self.ny = self.nz = n
        self.ng, self.nvar = ng, nvar
        self.brick = None
        self.f = None
        self.uold = torch.zeros((nvar, n + 2 * ng, n + 2 * ng, n + 2 * ng), dtype=torch.float64)


class TorchMoverDecomposition(BrickDecomposition):
    def _n(self, lev):
        return (lev.nx, lev.ny, lev.nz)

    def _slab_size(self, lev, nvar, face):
        s = slab_slices(self._n(lev), lev.ng, face, False)
        return nvar * int(np.prod([x.stop - x.start for x in s]))

    def _pack(self, lev, t, nvar, face, buf):
        s = slab_slices(self._n(lev), lev.ng, face, False)
        buf.copy_(t[(slice(None),) + s].reshape(-1))

    def _unpack(self, lev, t, nvar, face, buf):
        s = slab_slices(self._n(lev), lev.ng, face, True)
        view = t[(slice(None),) + s]
        view.copy_(buf.reshape(view.shape))

    def _multi(self, lev, t, nvar, part, pack):
        """torch stand-in of ramses_amd_halo_multi (the one-shot 26-region pack/unpack)"""
        b = part["boxes"].reshape(-1, 6)
        buf = part["buf"]
        for (ox, oy, oz, ex, ey, ez), off in zip(b, part["offsets"]):
            view = t[:, oz:oz + ez, oy:oy + ey, ox:ox + ex]
            m = nvar * ex * ey * ez
            if pack:
                buf[off:off + m] = view.reshape(-1)
            else:
                view.copy_(buf[off:off + m].reshape(view.shape))

    def _fill_periodic(self, lev, t, nvar, axes):
        for axis in range(3):
            if axes & (1 << axis):
                for hi in (0, 1):
                    dst = slab_slices(self._n(lev), lev.ng, 2 * axis + hi, True)
                    src = slab_slices(self._n(lev), lev.ng, 2 * axis + (1 - hi), False)
                    t[(slice(None),) + dst] = t[(slice(None),) + src]


def global_field(nvar, gz, gy, gx):
    """Unique value per (var, global cell)."""
    z, y, x = np.meshgrid(np.arange(gz), np.arange(gy), np.arange(gx), indexing="ij")
    base = (z * gy + y) * gx + x
    return np.stack([base + 1e7 * v for v in range(nvar)]).astype(np.float64)


def _worker(rank, world, pgrid, n, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ng, nvar = 2, 3
        dec = TorchMoverDecomposition(pgrid, rank, n, boxlen=1.0, ng=ng)
        lev = FakeLevel(n, ng, nvar)
        G = global_field(nvar, n * pgrid[2], n * pgrid[1], n * pgrid[0])
        cx, cy, cz = rank_coords(rank, pgrid)
        own = G[:, cz * n:(cz + 1) * n, cy * n:(cy + 1) * n, cx * n:(cx + 1) * n]
        lev.uold[:, ng:ng + n, ng:ng + n, ng:ng + n] = torch.from_numpy(own.copy())
        dec.exchange(lev, lev.uold, nvar)
        # expected: the periodic global field around this brick, ghosts included
        idx = lambda c, ext: (np.arange(c * n - ng, (c + 1) * n + ng)) % ext  # noqa: E731
        exp = G[:, idx(cz, G.shape[1])][:, :, idx(cy, G.shape[2])][:, :, :, idx(cx, G.shape[3])]
        ok = np.array_equal(lev.uold.numpy(), exp)
        # the one-shot exchange (one message per peer) fills the same ghosts
        lev1 = FakeLevel(n, ng, nvar)
        lev1.uold[:, ng:ng + n, ng:ng + n, ng:ng + n] = torch.from_numpy(own.copy())
        dec.exchange_direct(lev1, lev1.uold, nvar)
        ok = ok and np.array_equal(lev1.uold.numpy(), exp)
        # deep halo of the distributed multigrid (5 ghost layers, one field)
        ng5 = 5 if n >= 5 else n
        lev5 = FakeLevel(n, ng5, 1)
        lev5.uold[:, ng5:ng5 + n, ng5:ng5 + n, ng5:ng5 + n] = torch.from_numpy(own[:1].copy())
        dec5 = TorchMoverDecomposition(pgrid, rank, n, boxlen=1.0, ng=ng5)
        dec5.exchange(lev5, lev5.uold, 1)
        idx5 = lambda c, ext: (np.arange(c * n - ng5, (c + 1) * n + ng5)) % ext  # noqa: E731
        exp5 = G[:1, idx5(cz, G.shape[1])][:, :, idx5(cy, G.shape[2])][:, :, :, idx5(cx, G.shape[3])]
        ok = ok and np.array_equal(lev5.uold.numpy(), exp5)
        # the collectives of the multigrid driver
        tr = dec.transport
        ok = ok and tr.allreduce(float(rank + 1), "cpu") == world * (world + 1) / 2
        ok = ok and tr.allreduce(float(rank + 1), "cpu", op="min") == 1.0
        g = tr.allgather(torch.full((2, 3), float(rank)))
        ok = ok and g.shape == (world, 2, 3) and all(bool((g[r] == r).all()) for r in range(world))
        # the same through an explicit second process group (what bench.py falls back to when the
        # RCCL self-test fails: a gloo group next to the default one, host-staged tensors)
        from ramses_amd.transport import DistTransport
        tr2 = DistTransport(group=dist.new_group(backend="gloo"), staged=True)
        dec2 = TorchMoverDecomposition(pgrid, rank, n, boxlen=1.0, ng=ng, transport=tr2)
        lev2 = FakeLevel(n, ng, nvar)
        lev2.uold[:, ng:ng + n, ng:ng + n, ng:ng + n] = torch.from_numpy(own.copy())
        dec2.exchange_direct(lev2, lev2.uold, nvar)
        ok = ok and np.array_equal(lev2.uold.numpy(), exp)
        ok = ok and tr2.allreduce(float(rank + 1), "cpu", op="max") == float(world)
        g2 = tr2.allgather(torch.full((3,), float(rank)))
        ok = ok and g2.shape == (world, 3) and all(bool((g2[r] == r).all()) for r in range(world))
        tr2.barrier()
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("pgrid", [(2, 1, 1), (1, 2, 1), (2, 2, 1), (2, 2, 2)])
def test_halo_exchange_fills_all_ghosts(pgrid):
    world = pgrid[0] * pgrid[1] * pgrid[2]
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, pgrid, 6, _free_port(), ret), nprocs=world, join=True)
    assert len(ret) == world and all(ret.values()), dict(ret)


@pytest.mark.parametrize("pgrid", [(2, 2, 2), (2, 1, 1)])
def test_virtual_ranks_exchange_like_processes(pgrid):
    """LocalWorld (threads + mailboxes, used by the GPU tests to run multi-rank
    code on one device) moves the same slabs as the torch.distributed transport."""
    from ramses_amd.transport import LocalWorld
    world = pgrid[0] * pgrid[1] * pgrid[2]
    n, ng, nvar = 6, 2, 2
    G = global_field(nvar, n * pgrid[2], n * pgrid[1], n * pgrid[0])

    def body(tr):
        dec = TorchMoverDecomposition(pgrid, tr.rank, n, boxlen=1.0, ng=ng, transport=tr)
        lev = FakeLevel(n, ng, nvar)
        cx, cy, cz = rank_coords(tr.rank, pgrid)
        own = G[:, cz * n:(cz + 1) * n, cy * n:(cy + 1) * n, cx * n:(cx + 1) * n]
        lev.uold[:, ng:ng + n, ng:ng + n, ng:ng + n] = torch.from_numpy(own.copy())
        dec.exchange(lev, lev.uold, nvar)
        idx = lambda c, ext: (np.arange(c * n - ng, (c + 1) * n + ng)) % ext  # noqa: E731
        exp = G[:, idx(cz, G.shape[1])][:, :, idx(cy, G.shape[2])][:, :, :, idx(cx, G.shape[3])]
        s = tr.allreduce(float(tr.rank), "cpu")
        g = tr.allgather(torch.full((2,), float(tr.rank)))
        return bool(np.array_equal(lev.uold.numpy(), exp)) and s == world * (world - 1) / 2 and \
            [float(x) for x in g[:, 0]] == [float(r) for r in range(world)]

    assert all(LocalWorld(world).run(body))


def test_weak_scaling_rank_grid_matches_bench():
    import bench
    assert bench.rank_grid(1) == (1, 1, 1)
    assert bench.rank_grid(2) == (2, 1, 1)
    assert bench.rank_grid(4) == (2, 2, 1)
    assert bench.rank_grid(8) == (2, 2, 2)


class FakeBrick(FakeLevel):
    def __init__(self, dims, ng, nvar):
        self.nx, self.ny, self.nz = dims
        self.ng, self.nvar = ng, nvar
        self.brick = None
        self.f = None
        self.uold = torch.zeros((nvar, self.nz + 2 * ng, self.ny + 2 * ng, self.nx + 2 * ng), dtype=torch.float64)


@pytest.mark.parametrize("pgrid", [(1, 1, 2), (1, 2, 2), (2, 1, 1), (1, 2, 4)])
def test_bricks_that_are_not_cubes_exchange_their_deep_halo(pgrid):
    """The distributed multigrid on 2 or 4 ranks: the cubic box in bricks of unequal extents, the 5-cell halo moved in
    one round (one message per peer, self-wrap along the uncut axes), and the all-gather assembly of a replicated level."""
    from ramses_amd.poisson_parallel import assemble_level, brick_dims
    from ramses_amd.transport import LocalWorld
    world = pgrid[0] * pgrid[1] * pgrid[2]
    level, ng = 4, 3
    N = 1 << level
    dims = brick_dims(level, pgrid)
    assert tuple(d * p for d, p in zip(dims, pgrid)) == (N, N, N)
    G = global_field(1, N, N, N)

    def body(tr):
        dec = TorchMoverDecomposition(pgrid, tr.rank, dims, boxlen=1.0, ng=ng, transport=tr)
        lev = FakeBrick(dims, ng, 1)
        nx, ny, nz = dims
        cx, cy, cz = rank_coords(tr.rank, pgrid)
        own = G[:, cz * nz:(cz + 1) * nz, cy * ny:(cy + 1) * ny, cx * nx:(cx + 1) * nx]
        lev.uold[:, ng:ng + nz, ng:ng + ny, ng:ng + nx] = torch.from_numpy(own.copy())
        dec.exchange_direct(lev, lev.uold, 1)
        idx = lambda c, n: (np.arange(c * n - ng, (c + 1) * n + ng)) % N  # noqa: E731
        exp = G[:, idx(cz, nz)][:, :, idx(cy, ny)][:, :, :, idx(cx, nx)]
        parts = tr.allgather(torch.from_numpy(own[0].copy()))
        whole = assemble_level(parts, pgrid, dims)
        return bool(np.array_equal(lev.uold.numpy(), exp)) and bool(np.array_equal(whole.numpy(), G[0]))

    assert all(LocalWorld(world).run(body))


def test_multigrid_rank_grid_cuts_z_first():
    import bench
    assert bench.mg_rank_grid(1) == (1, 1, 1)
    assert bench.mg_rank_grid(2) == (1, 1, 2)
    assert bench.mg_rank_grid(4) == (1, 2, 2)
    assert bench.mg_rank_grid(8) == (2, 2, 2)


# ---- the overlapped step's ORDER of operations on 8 processes ---------------------------------------
class StencilLevel(FakeLevel):
    """a level whose "sweep" is a 13-point stencil of reach 2 (what the Godunov sweep reads: two cells per side), split like
    HydroLevel into the shell (cells within 2 of a brick face: what the neighbours receive) and the interior"""

    def __init__(self, n, ng, nvar):
        super().__init__(n, ng, nvar)
        self.unew = torch.zeros_like(self.uold)

    def _update(self, dt, mask):
        n, g, u = self.nx, self.ng, self.uold
        c = u[:, g:g + n, g:g + n, g:g + n]
        acc = torch.zeros_like(c)
        for d in range(3):
            for o, w in ((-2, 0.25), (-1, 1.0), (1, 1.0), (2, 0.25)):
                sl = [slice(g, g + n)] * 3
                sl[d] = slice(g + o, g + o + n)
                acc = acc + w * u[(slice(None),) + tuple(sl)]
        new = c + dt * (acc - 7.5 * c)
        tgt = self.unew[:, g:g + n, g:g + n, g:g + n]
        tgt[:, mask] = new[:, mask]

    def _shell_mask(self):
        n = self.nx
        i = torch.arange(n)
        near = (i < 2) | (i >= n - 2)
        return near[:, None, None] | near[None, :, None] | near[None, None, :]

    def godunov_fine(self, dt):
        self._update(dt, torch.ones((self.nx,) * 3, dtype=torch.bool))

    def godunov_fine_shell(self, dt):
        self._update(dt, self._shell_mask())

    def godunov_fine_interior(self, dt):
        self._update(dt, ~self._shell_mask())

    def set_uold(self):
        self.uold, self.unew = self.unew, self.uold


def _overlap_worker(rank, world, pgrid, n, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ng, nvar, dt, nsteps = 2, 2, 0.01, 3
        G = global_field(nvar, n * pgrid[2], n * pgrid[1], n * pgrid[0]) * 1e-6
        cx, cy, cz = rank_coords(rank, pgrid)
        own = G[:, cz * n:(cz + 1) * n, cy * n:(cy + 1) * n, cx * n:(cx + 1) * n]
        out = {}
        for mode in ("serial", "overlapped"):
            dec = TorchMoverDecomposition(pgrid, rank, n, boxlen=1.0, ng=ng)
            lev = StencilLevel(n, ng, nvar)
            lev.uold[:, ng:ng + n, ng:ng + n, ng:ng + n] = torch.from_numpy(own.copy())
            dec.exchange_direct(lev, lev.uold, nvar)
            for _ in range(nsteps):
                if mode == "serial":
                    lev.godunov_fine(dt)               # amr/amr_step.f90:388-510: sweep, set_uold, make_virtual_fine_dp
                    lev.set_uold()
                    dec.make_virtual_fine_dp(lev)
                else:
                    dec.step_overlapped(lev, dt)       # shell, exchange of the NEW state, interior, swap
            out[mode] = lev.uold.numpy().copy()
        # the single-process answer: the same stencil on the periodic global field
        U = G.copy()
        for _ in range(nsteps):
            acc = np.zeros_like(U)
            for d in range(3):
                for o, w in ((-2, 0.25), (-1, 1.0), (1, 1.0), (2, 0.25)):
                    acc = acc + w * np.roll(U, -o, axis=1 + d)
            U = U + dt * (acc - 7.5 * U)
        idx = lambda c, ext: (np.arange(c * n - ng, (c + 1) * n + ng)) % ext  # noqa: E731
        exp = U[:, idx(cz, U.shape[1])][:, :, idx(cy, U.shape[2])][:, :, :, idx(cx, U.shape[3])]
        ret[rank] = bool(np.array_equal(out["serial"], out["overlapped"])) and bool(np.allclose(out["overlapped"], exp, rtol=1e-13, atol=0.0))
    finally:
        dist.destroy_process_group()


def test_overlapped_step_equals_the_serial_schedule_on_8_processes():
    """BrickDecomposition.step_overlapped (shell sweep, exchange of the new state's ghosts, interior sweep, swap) against the
    reference's order (sweep, set_uold, make_virtual_fine_dp) on a 2 x 2 x 2 grid of processes: every ghost cell after every
    step, bit for bit, and the single-process result of the same stencil"""
    pgrid, world = (2, 2, 2), 8
    ret = mp.Manager().dict()
    mp.spawn(_overlap_worker, args=(world, pgrid, 6, _free_port(), ret), nprocs=world, join=True)
    assert len(ret) == world and all(ret.values()), dict(ret)
