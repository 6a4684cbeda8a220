#!/usr/bin/env python
"""Golden vectors of a uniform 2-D Sedov run (sedov3d.nml's setup with NDIM=2,
64^2, periodic), UNMODIFIED reference oracle/_ref/ramses2d.
    python tests/golden/make_golden_sedov2d.py -> tests/golden/sedov2d_ref_run.npz"""
import os
import shutil
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ramses_snapshot as rs  # noqa: E402

CASES = [("hllc_s2", "hllc", 2), ("llf_s1", "llf", 1)]
LEVEL, NSTEP = 6, 6


def main():
    out = {}
    for tag, riemann, slope in CASES:
        nml = rs.sedov3d_namelist(level=LEVEL, nstepmax=NSTEP, foutput=NSTEP, riemann=riemann, slope_type=slope)
        nml = nml.replace("ngridtot=", "ngridtot=20000 !")
        work, log = rs.run_reference(nml, ndim=2)
        try:
            for k in (1, 2):
                s = rs.load_leaf_cells(os.path.join(work, "output_%05d" % k))
                order = np.lexsort((s["x"][:, 0], s["x"][:, 1]))
                n = 2 ** LEVEL
                out["%s_prim%d" % (tag, k - 1)] = s["prim"][:, order].reshape(-1, n, n)
        finally:
            shutil.rmtree(work, ignore_errors=True)
    path = os.path.join(ROOT, "tests", "golden", "sedov2d_ref_run.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), out["llf_s1_prim1"].shape)


if __name__ == "__main__":
    main()
