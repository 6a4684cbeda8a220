#!/usr/bin/env python
"""Kernel-level goldens of the AMR multigrid: inputs/outputs of the UNMODIFIED reference's
gauss_seidel_mg_fine (red, black) and cmp_residual_mg_fine on a partially refined level,
dumped by oracle/dump_patch/multigrid_fine_fine.f90 inside the self-gravity AMR run of
make_golden_amr.py.
    oracle/build_ref.sh ramses 3 serial oracle/dump_patch
    python tests/golden/make_golden_amr_mg.py   -> tests/golden/amr_mg_ref.npz"""
import importlib.util
import os
import shutil
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ramses_snapshot as rs  # noqa: E402

CALLS = (81, 82, 1021)      # first red/black sweep and first residual of the first level-5 solve


def read(work, k):
    with open(os.path.join(work, "mgfine_%04d_in.bin" % k), "rb") as fh:
        kind, ilevel, ngrid, ngridmax, ncoarse, iflag = [int(x) for x in np.fromfile(fh, np.int32, 6)]
        ncell = ncoarse + 8 * ngridmax
        igrid = np.fromfile(fh, np.int32, ngrid)
        son = np.fromfile(fh, np.int32, ncell)
        nbor = np.fromfile(fh, np.int32, 6 * ngridmax).reshape(6, ngridmax)
        flag2 = np.fromfile(fh, np.int32, ncell + 1)[1:]      # flag2(0:ncell)
        phi = np.fromfile(fh, np.float64, ncell)
        f = np.fromfile(fh, np.float64, 3 * ncell).reshape(3, ncell)
        assert fh.read() == b""
    with open(os.path.join(work, "mgfine_%04d_out.bin" % k), "rb") as fh:
        phi_out = np.fromfile(fh, np.float64, ncell)
        f_out = np.fromfile(fh, np.float64, 3 * ncell).reshape(3, ncell)
    return dict(meta=np.array([kind, ilevel, ngrid, ngridmax, ncoarse, iflag], np.int64), igrid=igrid, son=son, nbor=nbor,
                flag2=flag2, phi=phi, f=f, phi_out=phi_out, f1_out=f_out[0])


def main():
    spec = importlib.util.spec_from_file_location("mka", os.path.join(ROOT, "tests", "golden", "make_golden_amr.py"))
    mka = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mka)
    os.environ["RAMSES_DUMP_MG"] = ",".join(str(c) for c in CALLS)
    work, log = rs.run_reference(mka.selfgrav_namelist(), binary=os.path.join(ROOT, "oracle", "_ref", "ramses3d_dump_patch"))
    out = {}
    try:
        for c in CALLS:
            d = read(work, c)
            lev = np.zeros(d["phi"].size, bool)
            for ind in range(8):
                lev[d["meta"][4] + ind * d["meta"][3] + d["igrid"] - 1] = True
            scan = d["flag2"][lev] // d["meta"][3]
            print(c, "kind", d["meta"][0], "level", d["meta"][1], "ngrid", d["meta"][2], "flags", d["meta"][5],
                  "scan cells", int(scan.sum()), "masked", int((d["f"][2][lev] <= 0).sum()),
                  "phi changed", int((d["phi"] != d["phi_out"]).sum()), "f1 changed", int((d["f"][0] != d["f1_out"]).sum()))
            for k, v in d.items():
                out["c%d_%s" % (c, k)] = v
    finally:
        shutil.rmtree(work, ignore_errors=True)
    out["calls"] = np.array(CALLS)
    path = os.path.join(ROOT, "tests", "golden", "amr_mg_ref.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path))


if __name__ == "__main__":
    main()
