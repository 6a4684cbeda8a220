#!/usr/bin/env python
"""Goldens of the conjugate-gradient Poisson solver (phi_fine_cg, poisson/phi_fine_cg.f90) on
partially refined levels: the self-gravity AMR run of make_golden_amr.py with cg_levelmin=4, so
levels 4 and 5 are solved by CG (level 3 = levelmin by multigrid).
  * kernel level: state at the start of the iteration loop (after the reference's cmp_residual_cg)
    and at its end, dumped by oracle/dump_patch/phi_fine_cg.f90 from the UNMODIFIED reference;
  * end to end: leaf cells (hydro + gravity) of the last snapshot and the solver's log lines.
    oracle/build_ref.sh ramses 3 serial oracle/dump_patch
    python tests/golden/make_golden_cg.py   -> tests/golden/cg_ref.npz"""
import importlib.util
import os
import re
import shutil
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ramses_snapshot as rs  # noqa: E402

SOLVES = (1, 2, 6)


def cg_namelist(eps="1e-5"):
    spec = importlib.util.spec_from_file_location("mka", os.path.join(ROOT, "tests", "golden", "make_golden_amr.py"))
    mka = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mka)
    return mka.selfgrav_namelist(eps).replace("epsilon=%s" % eps, "epsilon=%s\ncg_levelmin=4" % eps)


def cg_walls_namelist(eps="1e-5"):
    """The same run inside six reflexive walls: cmp_Ap_cg reads p in physical-boundary octs, which
    exist in the tree but are not in the level's list (constants of the solve)."""
    spec = importlib.util.spec_from_file_location("mka", os.path.join(ROOT, "tests", "golden", "make_golden_amr.py"))
    mka = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mka)
    return mka.walls_selfgrav_namelist().replace("epsilon=%s" % eps, "epsilon=%s\ncg_levelmin=4" % eps)


def read(work, k):
    with open(os.path.join(work, "cg_%04d_in.bin" % k), "rb") as fh:
        ilevel, ngrid, ngridmax, ncoarse = [int(x) for x in np.fromfile(fh, np.int32, 4)]
        epsilon, rho_tot, boxlen = [float(x) for x in np.fromfile(fh, np.float64, 3)]
        ncell = ncoarse + 8 * ngridmax
        igrid = np.fromfile(fh, np.int32, ngrid)
        son = np.fromfile(fh, np.int32, ncell)
        nbor = np.fromfile(fh, np.int32, 6 * ngridmax).reshape(6, ngridmax)
        phi = np.fromfile(fh, np.float64, ncell)
        rho = np.fromfile(fh, np.float64, ncell)
        f = np.fromfile(fh, np.float64, 3 * ncell).reshape(3, ncell)
        assert fh.read() == b""
    with open(os.path.join(work, "cg_%04d_out.bin" % k), "rb") as fh:
        phi_out = np.fromfile(fh, np.float64, ncell)
        f_out = np.fromfile(fh, np.float64, 3 * ncell).reshape(3, ncell)
    return dict(meta=np.array([ilevel, ngrid, ngridmax, ncoarse], np.int64), real=np.array([epsilon, rho_tot, boxlen]),
                igrid=igrid, son=son, nbor=nbor, phi=phi, rho=rho, f=f[:2].copy(), phi_out=phi_out, f_out=f_out)


def main():
    os.environ["RAMSES_DUMP_CG"] = ",".join(str(c) for c in SOLVES)
    work, log = rs.run_reference(cg_namelist(), binary=os.path.join(ROOT, "oracle", "_ref", "ramses3d_dump_patch"))
    out = {}
    try:
        solves = re.findall(r"==> Level=\s*(\d+) Step=\s*(\d+) Error=\s*(\S+)[ \t]+(\S+)[ \t]*\n", log)   # CG prints two errors
        print("CG solves (level, iterations)", [(a, b) for a, b, _, _ in solves])
        out["solves"] = np.array([[int(a), int(b)] for a, b, _, _ in solves])
        out["errors"] = np.array([[float(c), float(d)] for _, _, c, d in solves])
        for c in SOLVES:
            d = read(work, c)
            print(c, "level", d["meta"][0], "ngrid", d["meta"][1], "phi changed", int((d["phi"] != d["phi_out"]).sum()))
            for k, v in d.items():
                out["s%d_%s" % (c, k)] = v
        snap = rs.load_leaf_cells(os.path.join(work, "output_00002"), with_grav=True)
        order = np.lexsort((snap["x"][:, 0], snap["x"][:, 1], snap["x"][:, 2], snap["level"]))
        out["level"] = snap["level"][order].astype(np.int8)
        out["x"] = snap["x"][order]
        out["prim"] = snap["prim"][:, order]
        out["grav"] = snap["grav"][:, order]
    finally:
        shutil.rmtree(work, ignore_errors=True)
    # end to end inside walls (no dumps)
    os.environ.pop("RAMSES_DUMP_CG", None)
    work, log = rs.run_reference(cg_walls_namelist())
    try:
        solves = re.findall(r"==> Level=\s*(\d+) Step=\s*(\d+) Error=\s*(\S+)[ \t]+(\S+)[ \t]*\n", log)
        print("CG solves inside walls (level, iterations)", [(a, b) for a, b, _, _ in solves])
        out["w_solves"] = np.array([[int(a), int(b)] for a, b, _, _ in solves])
        snap = rs.load_leaf_cells(os.path.join(work, "output_00002"), with_grav=True)
        order = np.lexsort((snap["x"][:, 0], snap["x"][:, 1], snap["x"][:, 2], snap["level"]))
        out["w_level"] = snap["level"][order].astype(np.int8)
        out["w_prim"] = snap["prim"][:, order]
        out["w_grav"] = snap["grav"][:, order]
    finally:
        shutil.rmtree(work, ignore_errors=True)
    out["dumped"] = np.array(SOLVES)
    path = os.path.join(ROOT, "tests", "golden", "cg_ref.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path))


if __name__ == "__main__":
    main()
