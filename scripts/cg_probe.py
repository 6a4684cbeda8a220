#!/usr/bin/env python
"""Time per iteration of the conjugate-gradient solver (ramses_amd_cg_solve_host) on a fully
refined synthetic level, staging excluded: two solves with different iteration caps, the
difference divided by the extra iterations.  Algorithmic traffic 80 B per cell and iteration.
    python scripts/cg_probe.py [level] [morton|scrambled]"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ramses_amd  # noqa: E402,F401
from helpers import uniform_tree  # noqa: E402
from ramses_amd._capi import check, lib  # noqa: E402

L = int(sys.argv[1]) if len(sys.argv) > 1 else 7
ORDER = sys.argv[2] if len(sys.argv) > 2 else "morton"
T = uniform_tree(L, order=ORDER)
ncell = T["ncell"]
igrid = np.ascontiguousarray(T["igrid"], np.int32)
rng = np.random.default_rng(3)
lev = np.concatenate([T["ncoarse"] + ind * T["ngridmax"] + igrid - 1 for ind in range(8)])
r0 = np.zeros(ncell)
r0[lev] = rng.normal(size=lev.size)
r0[lev] -= r0[lev].mean()          # periodic level: the right-hand side must have zero mean
vp = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731


def solve(itermax, ordered=0):
    phi = np.zeros(ncell)
    f = np.zeros((3, ncell))
    f[0] = r0
    f[1] = r0
    it = C.c_int(0)
    err = (C.c_double * 3)()
    t0 = time.perf_counter()
    check(lib().ramses_amd_cg_solve_host(L, len(igrid), vp(igrid), vp(T["son"]), vp(T["nbor"]), T["ngridmax"], T["ncoarse"],
                                         vp(phi), vp(f), None, 0.0, 1.0, 8.0 * len(igrid), 1e-300, itermax, ordered,
                                         C.byref(it), err))
    return time.perf_counter() - t0, it.value, err[0] / err[1]


solve(5)
out = {"level": L, "cells": int(lev.size), "order": ORDER}
for ordered in (0, 1):
    lo, hi = (20, 220) if not ordered else (4, 24)
    t_lo, it_lo, _ = solve(lo, ordered)
    t_hi, it_hi, red = solve(hi, ordered)
    ms = (t_hi - t_lo) / (it_hi - it_lo) * 1e3
    out["ordered" if ordered else "parallel"] = {"ms_per_iteration": ms, "GB_per_s_at_80B": lev.size * 80 / ms / 1e6,
                                                  "residual_reduction_after_%d" % it_hi: red,
                                                  "staging_and_setup_s": t_lo - it_lo * ms * 1e-3}
print(json.dumps(out))
