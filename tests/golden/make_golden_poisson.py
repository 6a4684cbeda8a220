#!/usr/bin/env python
"""Generate tests/golden/poisson_ref_runs.npz from END-TO-END runs of the
reference program (oracle/_ref/ramses3d_rho = the reference compiled with its
own -DOUTPUT_PARTICLE_DENSITY so that backup_poisson also writes rho):
hydro + self-gravity, periodic uniform level, 'square' over-density blobs
(the namelist-only stand-in for cosmo.nml of SURVEY.md 8d).  Stored per case:
rho (the multigrid source), rho_tot, phi and f(1:3) of the first solve, and
the '==> Level= Step= Error=' line of the reference's log.

Runs only in the build container.   python tests/golden/make_golden_poisson.py
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

BLOB = """nregion=3
region_type(1)='square'
region_type(2)='square'
region_type(3)='square'
x_center=0.5,{xc},0.8
y_center=0.5,{yc},0.2
z_center=0.5,{zc},0.7
length_x=10.0,{lx},0.1
length_y=10.0,0.25,0.2
length_z=10.0,0.25,0.15
exp_region=10.0,10.0,2.0
d_region=1.0,{d2},3.0
u_region=0.0,0.0,0.0
v_region=0.0,0.0,0.0
p_region=1.0,1.0,1.0"""

CASES = [  # (key, level, boxlen, epsilon, blob parameters)
    ("l4_b1_e4", 4, 1.0, "1d-4", dict(xc=0.5, yc=0.5, zc=0.5, lx=0.25, d2=10.0)),
    ("l4_b2_e6", 4, 2.0, "1d-6", dict(xc=0.6, yc=0.8, zc=1.1, lx=0.5, d2=10.0)),
    ("l5_b1_e6", 5, 1.0, "1d-6", dict(xc=0.3, yc=0.55, zc=0.4, lx=0.4, d2=50.0)),
]


def main():
    binary = os.path.join(ROOT, "oracle", "_ref", "ramses3d_rho")
    arrays = {}
    for key, level, boxlen, eps, blob in CASES:
        nml = rs.sedov3d_namelist(level=level, nstepmax=2, foutput=1, boxlen=boxlen, poisson=True,
                                  init=BLOB.format(**blob), extra="&POISSON_PARAMS\nepsilon=%s\n/\n" % eps)
        work, out = rs.run_reference(nml, binary=binary)
        m = re.search(r"==> Level=\s*(\d+) Step=\s*(\d+) Error=\s*(\S+)", out)
        snap = rs.load_uniform_level(os.path.join(work, "output_00002"), level, with_grav=True)
        g = snap["grav"]
        assert g.shape[0] == 5
        arrays[key + "_rho"] = g[0]
        arrays[key + "_phi"] = g[1]
        arrays[key + "_f"] = g[2:5]
        arrays[key + "_prim2"] = snap["prim"]     # hydro state after one step WITH gravity
        arrays[key + "_meta"] = np.array([snap["info"]["rho_tot"], boxlen, float(eps.replace("d", "e")),
                                          float(m.group(2)), float(m.group(3))])
        print(key, "iters", m.group(2), "err", m.group(3), "rho_tot", snap["info"]["rho_tot"])
        shutil.rmtree(work)
    # three coarse steps of the last case (the gravity branch of amr_step three times over:
    # synchro_hydro_fine with the old and the new force, courant_fine/godunov_fine/set_uold with gravity)
    key, level, boxlen, eps, blob = CASES[-1]
    nml = rs.sedov3d_namelist(level=level, nstepmax=4, foutput=1, boxlen=boxlen, poisson=True,
                              init=BLOB.format(**blob), extra="&POISSON_PARAMS\nepsilon=%s\n/\n" % eps)
    work, out = rs.run_reference(nml, binary=binary)
    snap = rs.load_uniform_level(os.path.join(work, "output_00004"), level, with_grav=True)   # after 3 steps
    arrays[key + "_s3_grav"] = snap["grav"]
    arrays[key + "_s3_prim"] = snap["prim"]
    arrays[key + "_s3_iters"] = np.array([int(b) for _, b, _ in re.findall(r"==> Level=\s*(\d+) Step=\s*(\d+) Error=\s*(\S+)", out)])
    # the multigrid source of each of the three steps: rho_fine's hydro deposit (multipole_fine +
    # cic_from_multipole, pm/rho_fine.f90:666-891) is NOT the cell density bit for bit -- the mass of
    # a cell is CIC-deposited at its centre of mass (m*x)/m -- and it stays the reference's host code
    steps = [rs.load_uniform_level(os.path.join(work, "output_%05d" % k), level, with_grav=True) for k in (2, 3, 4)]
    arrays[key + "_s3_rho"] = np.stack([sn["grav"][0] for sn in steps])
    # rho_tot is recomputed by rho_fine every step from the summed multipole (:90-100): last bits move
    arrays[key + "_s3_rho_tot"] = np.array([sn["info"]["rho_tot"] for sn in steps])
    print(key, "3 steps: V-cycles per solve", arrays[key + "_s3_iters"])
    shutil.rmtree(work)
    np.savez_compressed(os.path.join(OUT, "poisson_ref_runs.npz"), **arrays)


if __name__ == "__main__":
    main()
