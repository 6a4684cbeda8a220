#!/usr/bin/env python
"""Golden vectors of BASELINE config C1: namelist/sedov1d.nml on a uniform level
(levelmin = levelmax = 7, 128 cells, NDIM=1, HLLC, slope_type=2, reflexive
boundaries), run with the UNMODIFIED reference (oracle/_ref/ramses1d).
    python tests/golden/make_golden_sedov1d.py  -> tests/golden/sedov1d_ref_run.npz
"""
import os
import re
import shutil
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ramses_snapshot as rs  # noqa: E402

NML = """
&RUN_PARAMS
hydro=.true.
nremap=10
ncontrol=1
nsubcycle=10*1
nstepmax={nstep}
/

&AMR_PARAMS
levelmin={level}
levelmax={level}
ngridmax=1000
nexpand=1
boxlen=0.5
/

&INIT_PARAMS
nregion=3
region_type(1)='square'
region_type(2)='point'
region_type(3)='point'
x_center=0.5,0.0,0.5
length_x=1.0,1.0,1.0
d_region=1.0,0.0,0.0
u_region=0.0,0.0,0.0
p_region=1e-5,0.4,0.3
/

&OUTPUT_PARAMS
foutput={foutput}
noutput=1
tout=1000.0
/

&HYDRO_PARAMS
gamma=1.4
courant_factor=0.8
slope_type={slope}
riemann='hllc'
/

&BOUNDARY_PARAMS
nboundary = 2
ibound_min=-1,+1
ibound_max=-1,+1
bound_type= {bt}
/
"""
# sedov1d.nml plus a second blast at the right wall, so that both boundaries are exercised
# (&BOUNDARY_PARAMS bound_type: 1 reflexive, 2 outflow, 3 imposed; the direction comes from
# ibound_min/max, hydro/read_hydro_params.f90:352-365)
CASES = [("reflexive", "1, 1", 40, (1, 11, 41)), ("outflow", "2, 2", 60, (61,))]
# the slope types only the NDIM=1 branch of uslope has (hydro/umuscl.f90:1030-1090: 4 superbee, 5 ultrabee, 6 central
# difference on the density) and type 3, which means type 2 there (:1014-1023): reflexive walls, snapshots at steps 0, 10, 20
SLOPE_CASES = [("st3", 3), ("st4", 4), ("st5", 5), ("st6", 6)]


def main():
    out = {}
    for tag, bt, nstep, snaps in CASES:
        work, log = rs.run_reference(NML.format(level=7, nstep=nstep, foutput=10 if tag == "reflexive" else nstep, bt=bt, slope=2), ndim=1)
        try:
            outs = sorted(d for d in os.listdir(work) if d.startswith("output_"))
            print(tag, outs)
            for k, d in enumerate(outs):
                s = rs.load_leaf_cells(os.path.join(work, d))
                order = np.argsort(s["x"][:, 0])
                out["%s_x%d" % (tag, k)] = s["x"][order, 0]
                out["%s_prim%d" % (tag, k)] = s["prim"][:, order]
        finally:
            shutil.rmtree(work, ignore_errors=True)
    for tag, slope in SLOPE_CASES:
        work, log = rs.run_reference(NML.format(level=7, nstep=20, foutput=10, bt="1, 1", slope=slope), ndim=1)
        try:
            outs = sorted(d for d in os.listdir(work) if d.startswith("output_"))
            print(tag, outs)
            for k, d in enumerate(outs):
                s = rs.load_leaf_cells(os.path.join(work, d))
                order = np.argsort(s["x"][:, 0])
                out["%s_prim%d" % (tag, k)] = s["prim"][:, order]
        finally:
            shutil.rmtree(work, ignore_errors=True)
    path = os.path.join(ROOT, "tests", "golden", "sedov1d_ref_run.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path))


if __name__ == "__main__":
    main()
