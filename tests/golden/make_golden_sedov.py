#!/usr/bin/env python
"""Generate tests/golden/sedov3d_ref_runs.npz from END-TO-END runs of the
reference program itself (oracle/_ref/ramses3d, built from /root/reference by
oracle/build_ref.sh): namelist/sedov3d.nml scaled to levelmin=levelmax=4
(16^3) with several solver settings; the stored arrays are the per-cell
primitive fields of the reference's own hydro_NNNNN.out snapshots after
0,1,2,3 coarse steps, and the fine-step dt sequence from its log.

Runs only in the build container.   python tests/golden/make_golden_sedov.py
"""
import os
import re
import shutil
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ramses_snapshot as rs  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
CASES = [("llf", 1, "muscl"), ("hllc", 2, "muscl"), ("hll", 7, "muscl"), ("acoustic", 8, "muscl"),
         ("exact", 1, "muscl"), ("hllc", 1, "plmde"), ("llf", 3, "muscl")]


def main():
    level = 4
    arrays = {}
    for riemann, slope, scheme in CASES:
        nml = rs.sedov3d_namelist(level=level, nstepmax=4, foutput=1, riemann=riemann, slope_type=slope, scheme=scheme)
        work, out = rs.run_reference(nml)
        dts = [float(m.group(1)) for m in re.finditer(r"Fine step=\s+\d+ t=\s*\S+ dt=\s*(\S+)", out)]
        key = "%s_s%d_%s" % (riemann, slope, scheme)
        for k in range(1, 5):
            snap = rs.load_uniform_level(os.path.join(work, "output_%05d" % k), level)
            arrays["%s_prim%d" % (key, k - 1)] = snap["prim"]
            arrays["%s_t%d" % (key, k - 1)] = np.array([snap["info"]["t"]])
        arrays[key + "_dtlog"] = np.array(dts[:4])
        shutil.rmtree(work)
        print(key, "dt log", dts[:4])
    np.savez_compressed(os.path.join(OUT, "sedov3d_ref_runs.npz"), **arrays)


if __name__ == "__main__":
    main()
