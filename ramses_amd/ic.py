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
