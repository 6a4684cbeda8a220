"""Shared test helpers: seeded synthetic states in the brick layout."""
import numpy as np


def random_brick(nx, ny, nz, seed, gamma=1.4, contrast=True, nvar=5):
    """Positive-density/pressure random state u[nvar,nz,ny,nx] with shocks:
    piecewise-constant blocks + noise, pressure spanning several decades."""
    rng = np.random.default_rng(seed)
    shp = (nz, ny, nx)
    rho = rng.uniform(0.2, 2.0, shp)
    vel = rng.normal(0.0, 0.7, (3,) + shp)
    p = rng.uniform(0.05, 2.0, shp)
    if contrast:
        # blocky jumps so the limiters and Riemann branches all fire
        bz, by, bx = max(nz // 3, 1), max(ny // 3, 1), max(nx // 3, 1)
        jump = 10.0 ** rng.uniform(-3, 2, (nz // bz + 1, ny // by + 1, nx // bx + 1))
        jump = np.repeat(np.repeat(np.repeat(jump, bz, 0), by, 1), bx, 2)[:nz, :ny, :nx]
        p = p * jump
        rho = rho * np.sqrt(jump)
    u = np.zeros((nvar,) + shp)
    u[0] = rho
    u[1:4] = rho * vel
    u[4] = p / (gamma - 1.0) + 0.5 * rho * (vel ** 2).sum(0)
    for n in range(5, nvar):
        u[n] = rho * rng.uniform(0, 1, shp)
    return u


def rel_linf(a, b):
    """max |a-b| / max |b| per variable, maximised over variables."""
    worst = 0.0
    for n in range(a.shape[0]):
        scale = np.abs(b[n]).max()
        if scale == 0.0:
            scale = 1.0
        worst = max(worst, np.abs(a[n] - b[n]).max() / scale)
    return worst


from ramses_amd.ic import _morton_rank, uniform_tree  # noqa: E402,F401  (synthetic trees live with the other synthetic inputs)
