#!/usr/bin/env python
"""Kernel-level goldens of rho_fine's hydro deposit on AMR levels (pm/rho_fine.f90: multipole_fine over the tree -- leaf
cells and the sums of the children of split cells -- and cic_from_multipole / cic_cell on partially refined levels, where a
CIC corner without an oct is dropped): the tree, the oct lists of every level a call visits, the density it reads and the
rho / multipole / rho_tot it leaves, dumped by oracle/dump_patch/rho_fine.f90 from the UNMODIFIED reference in the
self-gravitating AMR run of make_golden_amr.py (levels 3-5, sub-cycling, regridding).
    oracle/build_ref.sh ramses 3 serial oracle/dump_patch
    python tests/golden/make_golden_rho_amr.py   -> tests/golden/rho_fine_amr_ref.npz"""
import importlib.util
import os
import shutil
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ramses_snapshot as rs  # noqa: E402


def _load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def read_lists(work, k, ilevel):
    with open(os.path.join(work, "rho_%04d_lists.bin" % k), "rb") as fh:
        nlevelmax = int(np.fromfile(fh, np.int32, 1)[0])
        first, lists = [0], []
        for _ in range(ilevel, nlevelmax + 1):
            n = int(np.fromfile(fh, np.int32, 1)[0])
            lists.append(np.fromfile(fh, np.int32, n))
            first.append(first[-1] + n)
        assert fh.read() == b""
    return nlevelmax, np.array(first, np.int32), np.concatenate(lists).astype(np.int32)


def main():
    mkr = _load(os.path.join(ROOT, "tests", "golden", "make_golden_rho.py"), "mkr")
    mka = _load(os.path.join(ROOT, "tests", "golden", "make_golden_amr.py"), "mka")
    nml = mka.selfgrav_namelist().replace("foutput=%d" % mka.SELFGRAV_NSTEP, "foutput=1000")
    os.environ["RAMSES_DUMP_RHO"] = ",".join(str(c) for c in range(1, 80))
    work, log = rs.run_reference(nml, binary=os.path.join(ROOT, "oracle", "_ref", "ramses3d_dump_patch"))
    out = {}
    try:
        calls = sorted(int(f[4:8]) for f in os.listdir(work) if f.startswith("rho_") and f.endswith("_in.bin"))
        keep = {}
        for c in calls:
            d = mkr.read(work, c)
            ilevel, icount, ngrid, ngridmax, ncoarse, levelmin, nvector = [int(x) for x in d["meta"]]
            nlevelmax, first, igrid_all = read_lists(work, c, ilevel)
            nl = [int(first[i + 1] - first[i]) for i in range(len(first) - 1)]
            deposits = ilevel == levelmin or icount > 1
            print("call", c, "level", ilevel, "icount", icount, "octs per level", nl, "deposits" if deposits else "-")
            if deposits and sum(1 for n in nl if n > 0) >= 2:
                d["first"], d["igrid_all"], d["nlevelmax"] = first, igrid_all, np.array([nlevelmax])
                keep[(ilevel, c)] = d               # the LAST qualifying call of every level (the gas has moved)
        chosen = {}
        for (lev, c), d in sorted(keep.items()):
            chosen[lev] = (c, d)
        for lev, (c, d) in sorted(chosen.items()):
            for k, v in d.items():
                out["c%d_%s" % (c, k)] = v
        out["calls"] = np.array([c for _, (c, _) in sorted(chosen.items())])
        print("kept calls", out["calls"])
    finally:
        shutil.rmtree(work, ignore_errors=True)
    path = os.path.join(ROOT, "tests", "golden", "rho_fine_amr_ref.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path))


if __name__ == "__main__":
    main()
