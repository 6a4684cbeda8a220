"""make_golden_fast_cert.py -- TEST INFRASTRUCTURE ONLY.

The certificate of the drop-in's DEFAULT arithmetic (the fast build of the dense sweep: written-out FMAs,
v_rcp_f64 / v_rsq_f64 + Newton steps) against the REFERENCE PROGRAM at BASELINE config C2's size with a
developed shock: the unmodified reference (oracle/_ref/ramses3d_mpi) runs sedov3d.nml at 256^3 for 100
coarse steps (the blast wave is ~45 cells wide by then, limiter and floor branches all warm); kept are

  * sha256 of the whole assembled primitive state (the STRICT mode must reproduce it bit for bit),
  * t and nstep,
  * the max |value| of every snapshot variable over the whole level (the rel-Linf normalisation),
  * three full planes of every variable (k = 0: through the blast centre, which sits on the box corner;
    k = 12 and k = 30: through the shocked shell) -- the fast mode is compared with them to north_star's
    1e-12 relative L-infinity.  256^3 x 5 doubles do not fit in the repository; three planes do.

A 256^3 x 100-step run of the reference costs ~6 min on this container's 8 cores and ~3 min on the GPU box's
64, so the golden is made here once and the GPU test (tests/test_fast_certificate_gpu.py) runs only the
patched program at that size; the same test also runs a LIVE A/B at 128^3 over 120 steps on the box.

Run:  python tests/golden/make_golden_fast_cert.py [level=8] [nstep=100] [nproc=8]
Writes tests/golden/fast_cert_<n>_<nstep>.npz
"""
import hashlib
import os
import shutil
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ramses_snapshot as rs  # noqa: E402

PLANES = (0, 12, 30)


def namelist(level, nstep):
    return rs.sedov3d_namelist(level=level, nstepmax=nstep, foutput=nstep, mem_factor=3.0)


def digest(prim):
    return hashlib.sha256(np.ascontiguousarray(prim).tobytes()).hexdigest()


def main():
    level = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    nstep = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    nproc = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    binary = os.path.join(ROOT, "oracle", "_ref", "ramses3d_mpi")
    os.environ["RAMSES_AMD"] = "0"
    if os.environ.get("FAST_CERT_OUTPUT"):          # an output directory of the same run kept from an earlier invocation
        snap = rs.load_uniform_level(os.environ["FAST_CERT_OUTPUT"], level)
    else:
        work, out = rs.run_reference(namelist(level, nstep), binary=binary, nproc=nproc, timeout=7200)
        try:
            snap = rs.load_uniform_level(os.path.join(work, "output_00002"), level)
        finally:
            shutil.rmtree(work, ignore_errors=True)
    prim = snap["prim"]
    n = 2 ** level
    assert int(np.ravel(snap["info"]["nstep"])[0]) == nstep, snap["info"]["nstep"]
    path = os.path.join(ROOT, "tests", "golden", "fast_cert_%d_%d.npz" % (n, nstep))
    np.savez_compressed(path, planes=prim[:, list(PLANES), :, :], plane_k=np.array(PLANES),
                        vmax=np.abs(prim).reshape(prim.shape[0], -1).max(axis=1), t=snap["info"]["t"],
                        nstep=nstep, sha256=digest(prim))
    print("wrote", path, os.path.getsize(path), "bytes; t =", snap["info"]["t"], "rho range", prim[0].min(), prim[0].max())


if __name__ == "__main__":
    main()
