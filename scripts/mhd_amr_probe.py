"""Crash / sanity probe of the MHD sweep of AMR levels (csrc/mhd_amr.hip): a synthetic tree -- level L complete, level L+1 in a
spherical shell -- with a random state, ramses_amd_mhd_godunov_fine_amr_f90 on level L+1 then L.  RAMSES_AMD_DEBUG_SYNC=1 names
the stage a device fault belongs to.   python scripts/mhd_amr_probe.py [L]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ramses_amd import ic                      # noqa: E402
from ramses_amd._capi import lib               # noqa: E402

L = int(sys.argv[1]) if len(sys.argv) > 1 else 4
n = 2 ** L
z, y, x = np.meshgrid(np.arange(n), np.arange(n), np.arange(n), indexing="ij")
r = np.sqrt((x - n / 2 + 0.5) ** 2 + (y - n / 2 + 0.5) ** 2 + (z - n / 2 + 0.5) ** 2)
mask = (r >= 0.2 * n) & (r <= 0.38 * n)
T = ic.uniform_tree(L, order="morton", refine_mask=mask)
ncell = T["ncell"]
rng = np.random.default_rng(1)
u = np.zeros((11, ncell))
u[0] = 1.0 + rng.random(ncell)
for d in (1, 2, 3):
    u[d] = u[0] * (rng.random(ncell) - 0.5)
u[5:11] = 0.3 * (rng.random((6, ncell)) - 0.5)
u[4] = 2.0 + rng.random(ncell) + 0.5 * (u[1] ** 2 + u[2] ** 2 + u[3] ** 2) / u[0] + 0.125 * ((u[5:8] + u[8:11]) ** 2).sum(0)
unew = u.copy()


class P(C.Structure):
    _fields_ = [("gamma", C.c_double), ("smallr", C.c_double), ("smallc", C.c_double), ("slope_theta", C.c_double),
                ("slope_type", C.c_int32), ("slope_mag_type", C.c_int32), ("riemann", C.c_int32), ("riemann2d", C.c_int32)]


p = P(1.4, 1e-10, 1e-10, 1.5, 1, 1, 3, 5)
vp = lambda a: a.ctypes.data_as(C.c_void_p)      # noqa: E731
Lb = lib()
for lev, ig in ((L + 1, T["igrid_fine"]), (L, T["igrid"])):
    ig = np.ascontiguousarray(ig, np.int32)
    dx = 0.5 ** lev
    rc = Lb.ramses_amd_mhd_godunov_fine_amr_f90(C.byref(p), lev, L, len(ig), vp(ig), vp(T["son"]), vp(np.ascontiguousarray(T["nbor"])), vp(T["father"]),
                                                C.c_int64(T["ngridmax"]), C.c_int64(T["ncoarse"]), vp(u), vp(unew), None, 0, C.c_double(dx),
                                                C.c_double(0.05 * dx), 32, 0, 1, 1)
    print("level", lev, "octs", len(ig), "rc", rc, Lb.ramses_amd_last_error() if rc else "", "max |unew-u|", np.abs(unew - u).max(), flush=True)
print("finite:", np.isfinite(unew).all(), " sweeps", Lb.ramses_amd_mhd_amr_sweeps(), "octs", Lb.ramses_amd_mhd_amr_octs())
