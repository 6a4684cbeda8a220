#!/usr/bin/env python
"""Time of rho_fine's hydro deposit on the device (ramses_amd_resident_rho_fine_f90: order-tagged CIC gather +
sequential multipole sums) on a synthetic fully refined level:  python scripts/rho_probe.py [level]
Run under `rocprofv3 --kernel-trace --stats` for the per-kernel split."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ramses_amd  # noqa: E402

L = int(sys.argv[1]) if len(sys.argv) > 1 else 7
lib = ramses_amd.lib()
n, no = 2 ** L, 2 ** (L - 1)
ngrid = no ** 3
ngridmax = ngrid + 5
ncoarse = 1
ncell = ncoarse + 8 * ngridmax
oz, oy, ox = np.meshgrid(np.arange(no), np.arange(no), np.arange(no), indexing="ij")
igrid = np.arange(1, ngrid + 1, dtype=np.int32)
xg = np.zeros((3, ngridmax))
for d, o in enumerate((ox, oy, oz)):
    xg[d, :ngrid] = (o.reshape(-1) + 0.5) / no
rng = np.random.default_rng(0)
uold = rng.uniform(0.5, 2.0, (5, ncell))
p = ramses_amd.make_params()
vp = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
mp = np.zeros(4)
assert lib.ramses_amd_resident_invalidate() == 0


def call():
    rc = lib.ramses_amd_resident_rho_fine_f90(C.byref(p), L, ngrid, vp(igrid), vp(xg), ngridmax, ncoarse, 1, vp(uold), 1.0, 32, vp(mp))
    assert rc == 0, lib.ramses_amd_last_error()


call()
t0 = time.perf_counter()
K = 3
for _ in range(K):
    call()
ms = (time.perf_counter() - t0) / K * 1e3
print(json.dumps({"level": L, "cells": n ** 3, "rho_fine_ms_per_call": ms, "multipole": mp.tolist()}))
