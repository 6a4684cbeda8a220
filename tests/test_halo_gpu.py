"""GPU tests of the halo slab movers (pack / unpack / periodic fill) through
the C ABI, against the torch-slicing restatement in test_halo_gloo.py."""
import ctypes as C

import numpy as np
import pytest

from test_halo_gloo import slab_slices

pytestmark = pytest.mark.gpu


def _level(n, nvar=5, ng=2):
    import torch
    import ramses_amd
    from ramses_amd.hydro import HydroLevel
    lev = HydroLevel(n[0], n[1], n[2], 1.0 / n[0], params=ramses_amd.make_params(), ng=ng)
    g = torch.Generator(device="cpu").manual_seed(7)
    lev.uold.copy_(torch.rand(lev.uold.shape, generator=g, dtype=torch.float64))
    return lev


@pytest.mark.parametrize("n", [(8, 6, 10), (70, 12, 6), (16, 16, 16)])
def test_pack_unpack_match_slicing(gpu_lib, n):
    import torch
    from ramses_amd.parallel import BrickDecomposition
    lev = _level(n)
    dec = BrickDecomposition((1, 1, 1), 0, n[0])
    ref = lev.uold.cpu()
    for face in range(6):
        size = dec._slab_size(lev, lev.nvar, face)
        s = slab_slices(n, 2, face, False)
        assert size == ref[(slice(None),) + s].numel()
        buf = torch.empty(size, dtype=torch.float64, device="cuda")
        dec._pack(lev, lev.uold, lev.nvar, face, buf)
        torch.cuda.synchronize()
        assert torch.equal(buf.cpu(), ref[(slice(None),) + s].reshape(-1))
        # unpack a marker buffer into the ghost slab and check only that slab changed
        mark = torch.arange(size, dtype=torch.float64, device="cuda") + 1000.0
        before = lev.uold.clone()
        dec._unpack(lev, lev.uold, lev.nvar, face, mark)
        torch.cuda.synchronize()
        g = slab_slices(n, 2, face, True)
        after = lev.uold.cpu()
        exp = before.cpu()
        exp[(slice(None),) + g] = mark.cpu().reshape(exp[(slice(None),) + g].shape)
        assert torch.equal(after, exp)
        lev.uold.copy_(before)


def test_periodic_fill_matches_global_wrap(gpu_lib):
    import torch
    n = (20, 10, 12)
    lev = _level(n)
    inner = lev.interior(lev.uold).cpu().numpy()
    lev.make_virtual_fine_dp()
    torch.cuda.synchronize()
    got = lev.uold.cpu().numpy()
    iz = np.arange(-2, n[2] + 2) % n[2]
    iy = np.arange(-2, n[1] + 2) % n[1]
    ix = np.arange(-2, n[0] + 2) % n[0]
    exp = inner[:, iz][:, :, iy][:, :, :, ix]
    assert np.array_equal(got, exp)


@pytest.mark.parametrize("n", [(200, 40, 24), (64, 16, 16), (130, 30, 12)])
def test_shell_plus_interior_equals_full_sweep(gpu_lib, n):
    """The boundary-shell / interior split used for comm/compute overlap is a
    partition of the sweep: same bits as the single launch."""
    import torch
    import ramses_amd
    from ramses_amd.hydro import HydroLevel
    from helpers import random_brick
    u = random_brick(n[0], n[1], n[2], seed=n[0])
    dx, dt = 1.0 / 64, 0.02 / 64
    outs = []
    for split in (False, True):
        lev = HydroLevel(n[0], n[1], n[2], dx, params=ramses_amd.make_params(riemann="hllc", slope_type=2), ng=2)
        lev.upload(u)
        lev.make_virtual_fine_dp()
        lev.unew.fill_(-7.0)
        if split:
            lev.godunov_fine_interior(dt)
            lev.godunov_fine_shell(dt)
        else:
            lev.godunov_fine(dt)
        torch.cuda.synchronize()
        outs.append(lev.download(lev.unew))
    assert np.array_equal(outs[0], outs[1])
    assert (outs[0] != -7.0).all()


def test_overlapped_step_equals_plain_step(gpu_lib):
    """step_overlapped (second stream, events) == godunov_fine; set_uold; make_virtual_fine_dp
    on a single-rank periodic decomposition (the exchange is the periodic self-fill)."""
    import torch
    import ramses_amd
    from ramses_amd.parallel import BrickDecomposition
    from helpers import random_brick
    n = 64
    u = random_brick(n, n, n, seed=12)
    outs = []
    for overlapped in (False, True):
        dec = BrickDecomposition((1, 1, 1), 0, n, boxlen=1.0)
        lev = dec.make_level(ramses_amd.make_params(courant_factor=0.8))
        lev.upload(u)
        dec.make_virtual_fine_dp(lev)
        for _ in range(3):
            dt = lev.courant_fine()[0]
            if overlapped:
                dec.step_overlapped(lev, dt)
            else:
                lev.godunov_fine(dt)
                lev.set_uold()
                dec.make_virtual_fine_dp(lev)
        torch.cuda.synchronize()
        outs.append(lev.uold.cpu().numpy())
    assert np.array_equal(outs[0], outs[1])     # ghosts included


@pytest.mark.parametrize("pgrid", [(1, 1, 1), (2, 2, 2), (2, 1, 1)])
def test_one_shot_exchange_equals_axis_exchange(gpu_lib, pgrid):
    """ramses_amd_halo_multi + one grouped send/recv (exchange_direct) fills exactly the
    ghosts of the three-round axis exchange; multi-rank grids run as virtual ranks."""
    import torch
    import ramses_amd
    from ramses_amd.parallel import BrickDecomposition, rank_coords
    from ramses_amd.transport import LocalWorld
    world = pgrid[0] * pgrid[1] * pgrid[2]
    n = 12
    rng = np.random.default_rng(2)
    G = rng.normal(size=(5, n * pgrid[2], n * pgrid[1], n * pgrid[0]))

    def body(tr):
        dec = BrickDecomposition(pgrid, tr.rank, n, boxlen=1.0, transport=tr)
        lev = dec.make_level(ramses_amd.make_params())
        cx, cy, cz = rank_coords(tr.rank, pgrid)
        own = np.ascontiguousarray(G[:, cz * n:(cz + 1) * n, cy * n:(cy + 1) * n, cx * n:(cx + 1) * n])
        lev.upload(own)
        dec.exchange_direct(lev, lev.uold, lev.nvar)
        torch.cuda.synchronize()
        a = lev.uold.cpu().numpy()
        g = lev.ng
        idx = lambda c, ext: (np.arange(c * n - g, (c + 1) * n + g)) % ext  # noqa: E731
        exp = G[:, idx(cz, G.shape[1])][:, :, idx(cy, G.shape[2])][:, :, :, idx(cx, G.shape[3])]
        return bool(np.array_equal(a, exp))

    assert all(LocalWorld(world).run(body))
