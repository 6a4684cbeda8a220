#!/usr/bin/env python
"""Kernel-level goldens of rho_fine's hydro deposit (multipole_fine + cic_from_multipole / cic_cell,
pm/rho_fine.f90) on a uniform periodic level: the tree, the oct list, the density uold(:,1) it
reads and the rho / multipole / rho_tot it leaves, dumped by oracle/dump_patch/rho_fine.f90 from
the UNMODIFIED reference in the three-step self-gravity run of make_golden_poisson.py (the gas
moves from the second step on, so the deposit differs from the density in the last bits).
    oracle/build_ref.sh ramses 3 serial oracle/dump_patch
    python tests/golden/make_golden_rho.py   -> tests/golden/rho_fine_ref.npz"""
import importlib.util
import os
import shutil
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ramses_snapshot as rs  # noqa: E402


def read(work, k):
    with open(os.path.join(work, "rho_%04d_in.bin" % k), "rb") as fh:
        ilevel, icount, ngrid, ngridmax, ncoarse, levelmin, nvector = [int(x) for x in np.fromfile(fh, np.int32, 7)]
        boxlen, smallr = [float(x) for x in np.fromfile(fh, np.float64, 2)]
        ncell = ncoarse + 8 * ngridmax
        igrid = np.fromfile(fh, np.int32, ngrid)
        xg = np.fromfile(fh, np.float64, 3 * ngridmax).reshape(3, ngridmax)
        son = np.fromfile(fh, np.int32, ncell)
        nbor = np.fromfile(fh, np.int32, 6 * ngridmax).reshape(6, ngridmax)
        father = np.fromfile(fh, np.int32, ngridmax)
        dens = np.fromfile(fh, np.float64, ncell)
        assert fh.read() == b""
    with open(os.path.join(work, "rho_%04d_out.bin" % k), "rb") as fh:
        rho = np.fromfile(fh, np.float64, ncell)
        multipole = np.fromfile(fh, np.float64, 4)
        rho_tot = np.fromfile(fh, np.float64, 1)
        assert fh.read() == b""
    return dict(meta=np.array([ilevel, icount, ngrid, ngridmax, ncoarse, levelmin, nvector], np.int64),
                real=np.array([boxlen, smallr]), igrid=igrid, xg=xg, son=son, nbor=nbor, father=father, dens=dens,
                rho=rho, multipole=multipole, rho_tot=rho_tot)


def main():
    spec = importlib.util.spec_from_file_location("mkp", os.path.join(ROOT, "tests", "golden", "make_golden_poisson.py"))
    mkp = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mkp)
    key, level, boxlen, eps, blob = mkp.CASES[-1]
    nml = rs.sedov3d_namelist(level=level, nstepmax=4, foutput=10, boxlen=boxlen, poisson=True,
                              init=mkp.BLOB.format(**blob), extra="&POISSON_PARAMS\nepsilon=%s\n/\n" % eps)
    os.environ["RAMSES_DUMP_RHO"] = ",".join(str(c) for c in range(1, 12))
    work, log = rs.run_reference(nml, binary=os.path.join(ROOT, "oracle", "_ref", "ramses3d_dump_patch"))
    out = {}
    try:
        calls = sorted(int(f[4:8]) for f in os.listdir(work) if f.startswith("rho_") and f.endswith("_in.bin"))
        keep = []
        for c in calls:
            d = read(work, c)
            lev = np.zeros(d["dens"].size, bool)
            for ind in range(8):
                lev[d["meta"][4] + ind * d["meta"][3] + d["igrid"] - 1] = True
            diff = np.abs(d["rho"][lev] - d["dens"][lev]).max()
            print("call", c, "level", d["meta"][0], "icount", d["meta"][1], "ngrid", d["meta"][2],
                  "max |rho - density|", diff, "rho_tot", d["rho_tot"][0])
            if d["meta"][0] == d["meta"][5]:
                keep.append((c, d, diff))
        # the last two calls on levelmin: the gas moves, the deposit is not the density bit for bit
        for c, d, diff in keep[-2:]:
            assert diff > 0
            for k, v in d.items():
                out["c%d_%s" % (c, k)] = v
        out["calls"] = np.array([c for c, _, _ in keep[-2:]])
    finally:
        shutil.rmtree(work, ignore_errors=True)
    path = os.path.join(ROOT, "tests", "golden", "rho_fine_ref.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path))


if __name__ == "__main__":
    main()
