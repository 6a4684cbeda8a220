#!/usr/bin/env python
"""Throughput of the AMR (tree-walking, one wavefront per oct) sweep on a fully
refined synthetic tree, next to the dense brick sweep of the same level.
    python scripts/amr_probe.py [level]"""
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ramses_amd  # noqa: E402
from ramses_amd import ic  # noqa: E402
from ramses_amd.ic import uniform_tree  # noqa: E402
from ramses_amd._capi import check, lib  # noqa: E402

L = int(sys.argv[1]) if len(sys.argv) > 1 else 7
n = 2 ** L
ORDER = sys.argv[2] if len(sys.argv) > 2 else "morton"
T = uniform_tree(L, order=ORDER)
u, dx = ic.sedov3d(n)
uold = np.zeros((5, T["ncell"]))
T["to_cells"](u, uold)
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731
d_uold, d_unew = dev(uold), dev(uold)
d_son, d_nbor, d_father, d_igrid = dev(T["son"]), dev(T["nbor"]), dev(T["father"]), dev(T["igrid"])
nw = lib().ramses_amd_godunov_fine_amr_workspace(len(T["igrid"]), T["ngridmax"])
d_work = torch.zeros(int(nw), dtype=torch.uint8, device="cuda")
d_err = torch.zeros(1, dtype=torch.int32, device="cuda")
p = ramses_amd.make_params(courant_factor=0.8)
ptr = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
dt = 1e-6


def run():
    check(lib().ramses_amd_godunov_fine_amr_device(C.byref(p), L, len(T["igrid"]), ptr(d_igrid), ptr(d_son), ptr(d_nbor),
                                                   ptr(d_father), T["ngridmax"], T["ncoarse"], ptr(d_uold), ptr(d_unew),
                                                   None, None, None, dx, dt, 32, 0, 1, ptr(d_work), ptr(d_err),
                                                   C.c_void_p(torch.cuda.current_stream().cuda_stream)))


for _ in range(2):
    run()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
K = 5
for _ in range(K):
    run()
b.record()
torch.cuda.synchronize()
ms = a.elapsed_time(b) / K
from ramses_amd.hydro import HydroLevel  # noqa: E402
lev = HydroLevel(n, n, n, dx, params=p, ng=0)
lev.upload(u)
for _ in range(2):
    lev.godunov_fine(dt)
torch.cuda.synchronize()
a.record()
for _ in range(K):
    lev.godunov_fine(dt)
b.record()
torch.cuda.synchronize()
msd = a.elapsed_time(b) / K
print(json.dumps({"level": L, "oct_order": ORDER, "cells": n ** 3, "amr_sweep_ms": ms, "amr_cell_updates_per_s": n ** 3 / ms * 1e3,
                  "dense_sweep_ms_strict": msd, "dense_cell_updates_per_s": n ** 3 / msd * 1e3,
                  "tree_errors": int(d_err.item())}))
