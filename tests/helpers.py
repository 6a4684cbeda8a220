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


def _morton_rank(no):
    """rank of every oct position (oz,oy,ox) of an no^3 lattice along the Z-order curve"""
    oz, oy, ox = np.meshgrid(np.arange(no), np.arange(no), np.arange(no), indexing="ij")
    key = np.zeros_like(ox, dtype=np.int64)
    for b in range(max(1, int(np.log2(no)))):
        key |= ((ox >> b) & 1) << (3 * b) | ((oy >> b) & 1) << (3 * b + 1) | ((oz >> b) & 1) << (3 * b + 2)
    return np.argsort(np.argsort(key.reshape(-1))).reshape(no, no, no)


def uniform_tree(L, slack=7, order="scrambled", refine_box=None):
    """RAMSES tree arrays (amr/amr_commons.f90:67-75) of a periodic nx=ny=nz=1 box
    whose levels 1..L are fully refined; octs are numbered level by level in a
    scrambled order (the reference's lists are not lexicographic either).
    Returns dict(son, nbor[6, ngridmax], father, igrid (level L, list order),
    ncoarse, ngridmax, cells: function u[nvar,n,n,n] <-> cell vector)."""
    rng = np.random.default_rng(L)
    ncoarse = 1
    counts = [8 ** (l - 1) for l in range(1, L + 1)]
    nextra = 0
    if refine_box is not None:
        (x0, x1), (y0, y1), (z0, z1) = refine_box
        nextra = (x1 - x0) * (y1 - y0) * (z1 - z0)
    ngridmax = sum(counts) + nextra + slack
    ncell = ncoarse + 8 * ngridmax
    son = np.zeros(ncell, np.int32)
    nbor = np.zeros((6, ngridmax), np.int32)
    father = np.zeros(ngridmax, np.int32)
    ids = []                              # ids[l-1][oz,oy,ox] = 1-based oct index
    free = rng.permutation(ngridmax) + 1
    used = 0

    def cell_of(l, cx, cy, cz):
        """1-based cell index of level-l cell (cx,cy,cz); l = 0 is the coarse cell."""
        if l == 0:
            return np.ones_like(cx, dtype=np.int64)
        g = ids[l - 1][cz >> 1, cy >> 1, cx >> 1].astype(np.int64)
        ind = (cx & 1) + 2 * (cy & 1) + 4 * (cz & 1)
        return ncoarse + ind * ngridmax + g

    for l in range(1, L + 1):
        no = 2 ** (l - 1)
        n_oct = no ** 3
        if order == "morton":
            # siblings contiguous, levels one after the other: what refine_fine produces on a fresh grid
            idl = (used + 1 + _morton_rank(no)).astype(np.int32)
        else:
            idl = free[used:used + n_oct].reshape(no, no, no).astype(np.int32)
        used += n_oct
        oz, oy, ox = np.meshgrid(np.arange(no), np.arange(no), np.arange(no), indexing="ij")
        ids.append(idl)
        fcell = cell_of(l - 1, ox, oy, oz)                  # father cell: level l-1 cell at the oct's position
        father[idl - 1] = fcell
        son[fcell - 1] = idl
        for d in range(6):
            axis, up = d >> 1, d & 1
            c = [ox.copy(), oy.copy(), oz.copy()]
            c[axis] = (c[axis] + (1 if up else -1)) % no
            nbor[d, idl - 1] = cell_of(l - 1, c[0], c[1], c[2])
    idL = ids[-1]
    igrid = np.sort(idL.reshape(-1)).astype(np.int32) if order == "morton" else rng.permutation(idL.reshape(-1)).astype(np.int32)
    n = 2 ** L

    def to_cells(u, vec):
        kz, ky, kx = np.meshgrid(np.arange(n), np.arange(n), np.arange(n), indexing="ij")
        vec[:, cell_of(L, kx, ky, kz) - 1] = u

    def from_cells(vec):
        kz, ky, kx = np.meshgrid(np.arange(n), np.arange(n), np.arange(n), indexing="ij")
        return vec[:, cell_of(L, kx, ky, kz) - 1]

    out = dict(son=son, nbor=nbor, father=father, igrid=igrid, ncoarse=ncoarse, ngridmax=ngridmax, ncell=ncell,
               to_cells=to_cells, from_cells=from_cells)
    if refine_box is not None:
        # level L+1 octs in every level-L cell of the (periodic) box: a partially refined level whose
        # father cells all have their 3^3 neighbours (level L is fully refined)
        (x0, x1), (y0, y1), (z0, z1) = refine_box
        cz, cy, cx = np.meshgrid(np.arange(z0, z1) % n, np.arange(y0, y1) % n, np.arange(x0, x1) % n, indexing="ij")
        cz, cy, cx = cz.reshape(-1), cy.reshape(-1), cx.reshape(-1)
        idf = free[used:used + cx.size].astype(np.int32)
        fcell = cell_of(L, cx, cy, cz)
        father[idf - 1] = fcell
        son[fcell - 1] = idf
        for d in range(6):
            axis, up = d >> 1, d & 1
            c = [cx.copy(), cy.copy(), cz.copy()]
            c[axis] = (c[axis] + (1 if up else -1)) % n
            nbor[d, idf - 1] = cell_of(L, c[0], c[1], c[2])
        out["igrid_fine"] = rng.permutation(idf).astype(np.int32)
        out["fine_cells"] = lambda: np.concatenate([ncoarse + ind * ngridmax + idf for ind in range(8)])
    return out
