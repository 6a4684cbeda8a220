#!/usr/bin/env python
"""Generate tests/golden/hydro_unsplit_*.npz from the REFERENCE ITSELF.

Runs only in the build container (needs /root/reference and amdflang):
oracle/build_ref.sh compiles the unmodified reference sources into
oracle/_ref/libref_kernels*.so; this script feeds seeded random 6^ndim patches
to the reference's own unsplit() / riemann_*() / cmpdt() and stores inputs and
outputs.  The committed .npz files are what travels to the GPU box.

    python tests/golden/make_golden_hydro.py
"""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pyoracle as po  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
NV = 32      # NVECTOR of the reference build
NGRID = 2    # lanes actually filled and stored (keeps the fixture small)


def rand_patch(ndim, nvar, rng):
    us, gs, _, _ = po.patch_shapes(ndim, nvar, NV)
    rho = rng.uniform(0.1, 2.0, us[1:])
    vel = rng.normal(0, 1.0, (ndim,) + us[1:])
    p = rng.uniform(0.01, 3.0, us[1:]) * 10 ** rng.uniform(-5, 2, us[1:])
    u = np.zeros(us)
    u[0] = rho
    for d in range(ndim):
        u[1 + d] = rho * vel[d]
    u[ndim + 1] = p / 0.4 + 0.5 * rho * (vel ** 2).sum(0)
    for n in range(ndim + 2, nvar):
        u[n] = rho * rng.uniform(0, 1, us[1:])
    g = rng.normal(0, 1.0, gs)
    return u, g


def main():
    for ndim, nvar in ((1, 3), (2, 4), (3, 5), (3, 7)):
        if not po.ref_available(ndim, nvar):
            subprocess.check_call([os.path.join(ROOT, "oracle", "build_ref.sh"), "kernels", str(ndim), str(nvar)])
    rng = np.random.default_rng(20250117)
    cases = {}
    idx = 0
    for ndim, nvar in ((1, 3), (2, 4), (3, 5), (3, 7)):
        slopes = [1, 2, 3, 7, 8] + ([4, 5, 6] if ndim == 1 else [])
        for scheme in ("muscl", "plmde"):
            for riem in ("llf", "hllc", "hll", "acoustic", "exact"):
                # one slope type / difmag per (scheme, riemann) cell, cycling: keeps the file small
                st = slopes[idx % len(slopes)]
                difmag = 0.1 if idx % 3 == 0 else 0.0
                idx += 1
                p = po.make_params(ndim=ndim, nvar=nvar, riemann=riem, slope_type=st, scheme=scheme, difmag=difmag)
                u, g = rand_patch(ndim, nvar, rng)
                dx = 1.0 / 64
                dt = 0.3 * dx
                u[..., NGRID:] = 0.0
                g[..., NGRID:] = 0.0
                flux, tmp = po.ref_unsplit(p, u, g, dx, dt, ngrid=NGRID)
                u, g, flux, tmp = (a[..., :NGRID].copy() for a in (u, g, flux, tmp))
                key = "d%d_v%d_%s_%s_s%d_m%d" % (ndim, nvar, scheme, riem, st, int(difmag > 0))
                cases[key + "_uin"] = u
                cases[key + "_grav"] = g
                cases[key + "_flux"] = flux
                cases[key + "_tmp"] = tmp
                cases[key + "_dxdt"] = np.array([dx, dt])
    # cmpdt known answers (3-D)
    p = po.make_params(ndim=3)
    uu = np.zeros((5, NV))
    uu[0] = rng.uniform(0.1, 2, NV)
    uu[1:4] = rng.normal(0, 1, (3, NV)) * uu[0]
    uu[4] = rng.uniform(0.1, 3, NV) / 0.4 + 0.5 * (uu[1:4] ** 2).sum(0) / uu[0]
    gg = rng.normal(0, 1, (3, NV))
    cases["cmpdt_uu"] = uu
    cases["cmpdt_gg"] = gg
    cases["cmpdt_dt"] = np.array([po.ref_cmpdt(p, uu, gg, 1.0 / 64, 0.8)])
    np.savez_compressed(os.path.join(OUT, "hydro_unsplit_ref.npz"), **cases)
    print("wrote", len(cases), "arrays")


if __name__ == "__main__":
    main()
