"""GPU tests of the distributed multigrid driver (ramses_amd/poisson_parallel.py).

The multi-rank code path runs on ONE GPU with virtual ranks (LocalWorld: one
thread and one brick per rank, mailboxes instead of RCCL) and must reproduce the
single-brick solve of the same global level bit for bit: ghost-brick kernels,
deep-halo exchange, replicated coarse levels and the all-gather assembly."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _density(n, seed):
    rng = np.random.default_rng(seed)
    rho = 1.0 + 0.5 * rng.random((n, n, n))
    a, b, c = n // 8, n // 3, n // 2
    rho[a:a + n // 6, b:b + n // 4, c:c + n // 5] += 20.0
    rho[-n // 10:, :n // 7, n // 3:n // 2] += 7.0      # straddles the periodic / rank boundary
    return rho


def _single(rho, eps):
    import torch
    from ramses_amd.poisson import PoissonLevel
    n = rho.shape[0]
    lev = PoissonLevel(int(np.log2(n)), boxlen=1.0, epsilon=eps)
    lev.rho.copy_(torch.from_numpy(rho).cuda())
    it, err = lev.multigrid_fine(float(rho.mean()))
    lev.force_fine()
    torch.cuda.synchronize()
    return it, err, lev.phi.cpu().numpy(), lev.f.cpu().numpy()


@pytest.mark.parametrize("n", [64, 128])
def test_one_rank_ghost_bricks_equal_dense_solve(gpu_lib, n):
    import torch
    from ramses_amd.poisson_parallel import PoissonDecomposition
    rho = _density(n, 3)
    it0, err0, phi0, f0 = _single(rho, 1e-6)
    pd = PoissonDecomposition((1, 1, 1), 0, n, boxlen=1.0, epsilon=1e-6)
    pd.rho.copy_(torch.from_numpy(rho).cuda())
    it, err = pd.multigrid_fine(float(rho.mean()))
    pd.force_fine()
    torch.cuda.synchronize()
    assert it == it0
    assert np.array_equal(pd.phi_interior().cpu().numpy(), phi0)
    assert np.array_equal(pd.f.cpu().numpy(), f0)
    assert abs(err - err0) <= 1e-12 * err0


@pytest.mark.parametrize("n", [64, 128])
def test_eight_virtual_ranks_equal_single_brick(gpu_lib, n):
    """n=64: one distributed level above the replicated ones; n=128: two
    distributed levels (distributed coarse cycle, coarse halo for the prolongation)."""
    import torch
    from ramses_amd.parallel import rank_coords
    from ramses_amd.poisson_parallel import PoissonDecomposition
    from ramses_amd.transport import LocalWorld
    p = 2
    N = n * p
    rho = _density(N, 11)
    rho_tot = float(rho.mean())
    it0, err0, phi0, f0 = _single(rho, 1e-6)

    def body(tr):
        cx, cy, cz = rank_coords(tr.rank, (p, p, p))
        pd = PoissonDecomposition((p, p, p), tr.rank, n, boxlen=1.0, epsilon=1e-6, transport=tr)
        sub = rho[cz * n:(cz + 1) * n, cy * n:(cy + 1) * n, cx * n:(cx + 1) * n]
        pd.rho.copy_(torch.from_numpy(np.ascontiguousarray(sub)).cuda())
        it, err = pd.multigrid_fine(rho_tot)
        pd.force_fine()
        torch.cuda.synchronize()
        return (it, err, pd.phi_interior().cpu().numpy(), pd.f.cpu().numpy(), pd.exchanges, (cx, cy, cz))

    out = LocalWorld(p ** 3).run(body)
    for it, err, phi, f, nex, (cx, cy, cz) in out:
        assert it == it0
        sl = (slice(cz * n, (cz + 1) * n), slice(cy * n, (cy + 1) * n), slice(cx * n, (cx + 1) * n))
        assert np.array_equal(phi, phi0[sl]), (cx, cy, cz)
        assert np.array_equal(f, f0[(slice(None),) + sl]), (cx, cy, cz)
        assert abs(err - err0) <= 1e-10 * err0
    # communication-avoiding: per V-cycle the fine level exchanges phi twice (+ the coarse
    # correction once); the reference exchanges after each of the 8 colour passes
    nlev = 1 if n == 64 else 2
    assert out[0][4] <= 2 + it0 * (2 + 4 * (nlev - 1) + 1) + 1


@pytest.mark.parametrize("level,pgrid", [(7, (2, 1, 1)), (7, (1, 1, 2)), (7, (2, 2, 1)), (7, (1, 2, 2)),
                                         (8, (1, 2, 4)), (8, (2, 1, 1)), (8, (1, 2, 2))])
def test_bricks_of_two_and_four_ranks_equal_single_brick(gpu_lib, level, pgrid):
    """The reference's box is a cube, so 2 or 4 ranks own bricks that are not cubes (half boxes, quarter columns):
    non-cubic ghost bricks through the fused smoother, the ghost restriction / prolongation / gradient kernels, the
    halo plan and the all-gather assembly of the replicated levels.  Level 8 with >= 128-cell extents: two distributed
    levels (the coarse halo of the prolongation too)."""
    import torch
    from ramses_amd.poisson_parallel import PoissonDecomposition
    from ramses_amd.transport import LocalWorld
    N = 1 << level
    rho = _density(N, 5)
    rho_tot = float(rho.mean())
    it0, err0, phi0, f0 = _single(rho, 1e-6)
    world = pgrid[0] * pgrid[1] * pgrid[2]

    def body(tr):
        pd = PoissonDecomposition(pgrid, tr.rank, level=level, boxlen=1.0, epsilon=1e-6, transport=tr)
        sl = pd.my_slices()
        pd.rho.copy_(torch.from_numpy(np.ascontiguousarray(rho[sl])).cuda())
        it, err = pd.multigrid_fine(rho_tot)
        pd.force_fine()
        torch.cuda.synchronize()
        return (it, err, pd.phi_interior().cpu().numpy(), pd.f.cpu().numpy(), sl, pd.nlev)

    out = LocalWorld(world).run(body)
    for it, err, phi, f, sl, nlev in out:
        assert it == it0
        assert np.array_equal(phi, phi0[sl]), (pgrid, sl)
        assert np.array_equal(f, f0[(slice(None),) + sl]), (pgrid, sl)
        assert abs(err - err0) <= 1e-10 * err0
    if level == 8 and pgrid != (1, 2, 4):
        assert out[0][5] == 2
