"""Initial conditions from &INIT_PARAMS regions on a uniform level brick.

Host-side mirror (numpy; not on the hot path) of the reference's
region_condinit (hydro/init_flow_fine.f90:455-596) and condinit
(hydro/condinit.f90:1-73) evaluated at the cell centres of a fully refined
level: x = (i+1/2)*dx.  Produces the conserved brick u[nvar,nz,ny,nx].
"""
import numpy as np


def region_condinit(x, y, z, dx, regions, gamma=1.4, smallr=1e-10, smallc=1e-10, ndim=3):
    """Primitive q = (rho,u,v,w,P) on broadcastable coordinate arrays.  ndim < 3:
    the unused directions drop out as in the reference (xn,yn,zn start at 0 for
    'square' and at 1 for 'point', vol = dx**ndim)."""
    shape = np.broadcast(x, y, z).shape
    q = np.zeros((5,) + shape)
    q[0] = smallr
    q[4] = smallr * smallc ** 2 / gamma
    for r in regions:
        if r["type"] == "square":
            xn = 2.0 * np.abs(x - r["x_center"]) / r["length_x"]
            yn = 2.0 * np.abs(y - r["y_center"]) / r["length_y"] if ndim > 1 else 0.0 * y
            zn = 2.0 * np.abs(z - r["z_center"]) / r["length_z"] if ndim > 2 else 0.0 * z
            en = r.get("exp_region", 2.0)
            if en < 10:
                rad = (xn ** en + yn ** en + zn ** en) ** (1.0 / en)
            else:
                rad = np.maximum(np.maximum(xn, yn), zn)
            inside = np.broadcast_to(rad < 1.0, shape)
            q[0][inside] = r.get("d_region", 0.0)
            q[1][inside] = r.get("u_region", 0.0)
            q[2][inside] = r.get("v_region", 0.0)
            q[3][inside] = r.get("w_region", 0.0)
            q[4][inside] = r.get("p_region", 0.0)
        elif r["type"] == "point":
            vol = dx ** ndim
            xn = np.maximum(1.0 - np.abs(x - r["x_center"]) / dx, 0.0)
            yn = np.maximum(1.0 - np.abs(y - r["y_center"]) / dx, 0.0) if ndim > 1 else 1.0 + 0.0 * y
            zn = np.maximum(1.0 - np.abs(z - r["z_center"]) / dx, 0.0) if ndim > 2 else 1.0 + 0.0 * z
            w = xn * yn * zn
            q[0] = q[0] + r.get("d_region", 0.0) * w / vol
            q[1] = q[1] + r.get("u_region", 0.0) * w
            q[2] = q[2] + r.get("v_region", 0.0) * w
            q[3] = q[3] + r.get("w_region", 0.0) * w
            q[4] = q[4] + r.get("p_region", 0.0) * w / vol
        else:
            raise ValueError("unknown region type %r" % r["type"])
    return q


def condinit(q, gamma=1.4):
    """Primitive -> conservative, same summation order as condinit.f90:44-57."""
    u = np.zeros_like(q)
    u[0] = q[0]
    u[1] = q[0] * q[1]
    u[2] = q[0] * q[2]
    u[3] = q[0] * q[3]
    e = np.zeros_like(q[0])
    e = e + 0.5 * q[0] * q[1] ** 2
    e = e + 0.5 * q[0] * q[2] ** 2
    e = e + 0.5 * q[0] * q[3] ** 2
    e = e + q[4] / (gamma - 1.0)
    u[4] = e
    return u


# namelist/sedov3d.nml &INIT_PARAMS
SEDOV3D_REGIONS = [
    dict(type="square", x_center=0.5, y_center=0.5, z_center=0.5, length_x=10.0, length_y=10.0,
         length_z=10.0, exp_region=10.0, d_region=1.0, u_region=0.0, v_region=0.0, p_region=1e-5),
    dict(type="point", x_center=0.0, y_center=0.0, z_center=0.0, length_x=1.0, length_y=1.0,
         length_z=1.0, exp_region=10.0, d_region=0.0, u_region=0.0, v_region=0.0, p_region=0.4),
]


def uniform_brick_ic(n, boxlen, regions, gamma=1.4, lo=(0, 0, 0), shape=None):
    """Conserved state of a sub-brick [lo, lo+shape) of the n^3 level."""
    dx = boxlen / n
    shape = (n, n, n) if shape is None else shape
    xs = (np.arange(lo[0], lo[0] + shape[0]) + 0.5) * dx
    ys = (np.arange(lo[1], lo[1] + shape[1]) + 0.5) * dx
    zs = (np.arange(lo[2], lo[2] + shape[2]) + 0.5) * dx
    q = region_condinit(xs[None, None, :], ys[None, :, None], zs[:, None, None], dx, regions, gamma)
    return condinit(q, gamma), dx


def sedov3d(n, boxlen=0.5, gamma=1.4, lo=(0, 0, 0), shape=None):
    """namelist/sedov3d.nml on a uniform n^3 level (levelmin=levelmax=log2 n)."""
    return uniform_brick_ic(n, boxlen, SEDOV3D_REGIONS, gamma, lo, shape)


def sedov3d_corner_and_background(n, boxlen=0.5, gamma=1.4):
    """sedov3d.nml on an n^3 level WITHOUT building the level on the host (bench.py at 512^3: 5 GB): the 'point'
    region sits at the box corner (x_center = y_center = z_center = 0) and region_condinit's CIC weights
    max(1 - |x - x_center|/dx, 0) are not periodic (hydro/init_flow_fine.f90:555-594), so of the eight cells of its
    cloud only cell (0,0,0) lies inside the box: the level is a uniform background plus that ONE cell.  Returns
    (u_corner[5], u_background[5], dx), the reference's own arithmetic (condinit.f90:44-57) on the 2^3 corner brick;
    tests/test_ic_sedov.py checks the statement against the full construction."""
    u, dx = uniform_brick_ic(n, boxlen, SEDOV3D_REGIONS, gamma, (0, 0, 0), (2, 2, 2))
    corner, back = u[:, 0, 0, 0].copy(), u[:, 1, 1, 1].copy()
    assert all(np.array_equal(u[:, k, j, i], back) for k in range(2) for j in range(2) for i in range(2) if (i, j, k) != (0, 0, 0))
    return corner, back, dx


# namelist/sedov1d.nml &INIT_PARAMS
SEDOV1D_REGIONS = [
    dict(type="square", x_center=0.5, y_center=0.0, z_center=0.0, length_x=1.0, length_y=1.0, length_z=1.0,
         exp_region=2.0, d_region=1.0, u_region=0.0, v_region=0.0, p_region=1e-5),
    dict(type="point", x_center=0.0, y_center=0.0, z_center=0.0, length_x=1.0, length_y=1.0, length_z=1.0,
         exp_region=2.0, d_region=0.0, u_region=0.0, v_region=0.0, p_region=0.4),
]


def sedov1d(n, boxlen=0.5, gamma=1.4):
    """namelist/sedov1d.nml on a uniform level of n cells, as a [5,1,1,n] brick
    (NDIM=1 embedded: v = w = 0)."""
    dx = boxlen / n
    xs = (np.arange(n) + 0.5) * dx
    zero = np.zeros(1)
    q = region_condinit(xs[None, None, :], zero[None, :, None], zero[:, None, None], dx, SEDOV1D_REGIONS, gamma, ndim=1)
    return condinit(q, gamma), dx


# ---------------------------------------------------------------------------
# synthetic octrees (tests, scripts/amr_probe.py, bench.py's tree-walking sweep line)
# ---------------------------------------------------------------------------
def _morton_rank(no):
    """rank of every oct position (oz,oy,ox) of an no^3 lattice along the Z-order curve"""
    oz, oy, ox = np.meshgrid(np.arange(no), np.arange(no), np.arange(no), indexing="ij")
    key = np.zeros_like(ox, dtype=np.int64)
    for b in range(max(1, int(np.log2(no)))):
        key |= ((ox >> b) & 1) << (3 * b) | ((oy >> b) & 1) << (3 * b + 1) | ((oz >> b) & 1) << (3 * b + 2)
    return np.argsort(np.argsort(key.reshape(-1))).reshape(no, no, no)


def uniform_tree(L, slack=7, order="scrambled", refine_box=None, refine_mask=None, refine_mask2=None, refine_mask3=None):
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
    if refine_mask is not None:
        nextra = int(np.count_nonzero(refine_mask))
    if refine_mask2 is not None:
        nextra += int(np.count_nonzero(refine_mask2))
    if refine_mask3 is not None:
        nextra += int(np.count_nonzero(refine_mask3))
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
    if refine_box is not None or refine_mask is not None:
        # level L+1 octs in every level-L cell of the (periodic) box / of the mask[z,y,x]: a partially refined level whose
        # father cells all have their 3^3 neighbours (level L is fully refined)
        if refine_mask is not None:
            cz, cy, cx = np.nonzero(refine_mask)
        else:
            (x0, x1), (y0, y1), (z0, z1) = refine_box
            cz, cy, cx = np.meshgrid(np.arange(z0, z1) % n, np.arange(y0, y1) % n, np.arange(x0, x1) % n, indexing="ij")
            cz, cy, cx = cz.reshape(-1), cy.reshape(-1), cx.reshape(-1)
        if order == "morton":
            # numbered along the Z-order curve of their father cells (siblings contiguous), after the coarser levels
            key = np.zeros(cx.size, dtype=np.int64)
            for b in range(L):
                key |= ((cx >> b) & 1) << (3 * b) | ((cy >> b) & 1) << (3 * b + 1) | ((cz >> b) & 1) << (3 * b + 2)
            o = np.argsort(key, kind="stable")
            cz, cy, cx = cz[o], cy[o], cx[o]
            idf = (used + 1 + np.arange(cx.size)).astype(np.int32)
        else:
            idf = free[used:used + cx.size].astype(np.int32)
        fcell = cell_of(L, cx, cy, cz)
        father[idf - 1] = fcell
        son[fcell - 1] = idf
        for d in range(6):
            axis, up = d >> 1, d & 1
            c = [cx.copy(), cy.copy(), cz.copy()]
            c[axis] = (c[axis] + (1 if up else -1)) % n
            nbor[d, idf - 1] = cell_of(L, c[0], c[1], c[2])
        out["igrid_fine"] = idf.copy() if order == "morton" else rng.permutation(idf).astype(np.int32)
        out["fine_cells"] = lambda: np.concatenate([ncoarse + ind * ngridmax + idf for ind in range(8)])
        # further levels: level L+1+k octs in the level-(L+k) cells of a mask over that level's (2^k n)^3 cell grid (cells that
        # exist and whose 3^3 neighbours exist: the caller keeps each mask inside the level below)
        used += cx.size
        prev_pos, prev_id, npos = (cz, cy, cx), idf, n          # oct positions of the level below, in units of its octs
        for k, (mk, name) in enumerate(((refine_mask2, "igrid_fine2"), (refine_mask3, "igrid_fine3")), start=1):
            if mk is None:
                break
            assert mk.shape == (2 * npos,) * 3
            idsf = np.zeros((npos, npos, npos), np.int32)
            idsf[prev_pos[0], prev_pos[1], prev_pos[2]] = prev_id
            ez, ey, ex = np.nonzero(mk)
            nn = 2 * npos

            def cell_f(fx, fy, fz, idsf=idsf, npos=npos):
                g = idsf[(fz >> 1) % npos, (fy >> 1) % npos, (fx >> 1) % npos].astype(np.int64)
                assert (g > 0).all(), "a refinement mask reaches outside the level below"
                return ncoarse + ((fx & 1) + 2 * (fy & 1) + 4 * (fz & 1)) * ngridmax + g
            if order == "morton":
                key = np.zeros(ex.size, dtype=np.int64)
                for b in range(L + k):
                    key |= ((ex >> b) & 1) << (3 * b) | ((ey >> b) & 1) << (3 * b + 1) | ((ez >> b) & 1) << (3 * b + 2)
                o = np.argsort(key, kind="stable")
                ez, ey, ex = ez[o], ey[o], ex[o]
                idg = (used + 1 + np.arange(ex.size)).astype(np.int32)
            else:
                idg = free[used:used + ex.size].astype(np.int32)
            fc2 = cell_f(ex, ey, ez)
            father[idg - 1] = fc2
            son[fc2 - 1] = idg
            for d in range(6):
                axis, up = d >> 1, d & 1
                c = [ex.copy(), ey.copy(), ez.copy()]
                c[axis] = (c[axis] + (1 if up else -1)) % nn
                nbor[d, idg - 1] = cell_f(c[0], c[1], c[2])
            out[name] = idg.copy() if order == "morton" else rng.permutation(idg).astype(np.int32)
            used += ex.size
            prev_pos, prev_id, npos = (ez, ey, ex), idg, nn
    return out
