#!/usr/bin/env python
"""Generate tests/golden/amr_ops_ref.npz from the REFERENCE's own interpol_hydro
and upl (oracle/_ref/libref_kernels3d.so, built from /root/reference by
oracle/build_ref.sh).  Runs only in the build container."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pyoracle as po  # noqa: E402


def state(rng, shape):
    rho = rng.uniform(0.2, 2, shape)
    vel = rng.normal(0, 1, (3,) + shape)
    p = rng.uniform(0.1, 2, shape) * 10 ** rng.uniform(-2, 1, shape)
    u = np.zeros((5,) + shape)
    u[0] = rho
    u[1:4] = rho * vel
    u[4] = p / 0.4 + 0.5 * rho * (vel ** 2).sum(0)
    return u


def main():
    R = po.ref(3)
    rng = np.random.default_rng(777)
    nv = 32
    out = {}
    for iv in (0, 1, 2):
        for it in (1, 2, 3, 4):
            if it == 4 and iv != 2:
                continue
            smallr = 0.5 if it % 2 == 0 else 1e-10
            u1 = state(rng, (7, nv))
            u2 = np.zeros((5, 8, nv))
            R.ref_interpol_hydro(u1.copy(), u2, nv, iv, it, smallr)
            key = "interp_v%d_t%d" % (iv, it)
            out[key + "_u1"], out[key + "_u2"], out[key + "_smallr"] = u1, u2, np.array([smallr])
        child = state(rng, (8, nv))
        parent = np.zeros((5, nv))
        smallr = 0.6 if iv == 1 else 1e-10
        R.ref_upl(child.copy(), parent, nv, iv, smallr)
        out["upl_v%d_child" % iv], out["upl_v%d_parent" % iv], out["upl_v%d_smallr" % iv] = child, parent, np.array([smallr])
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "amr_ops_ref.npz"), **out)
    print("wrote", len(out), "arrays")


if __name__ == "__main__":
    main()
