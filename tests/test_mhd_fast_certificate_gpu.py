"""Certificate of the FAST arithmetic of the MHD sweep (ramses_amd_mhd_godunov_brick_fast, csrc/mhd_sweep.hip compiled with
-fapprox-func -ffp-contract=fast: divisions as v_rcp_f64 + Newton steps, contracted multiply-adds; RAMSES_AMD_MHD_FAST=1).

It is held to north_star's tolerance -- 1e-12 relative L-infinity per snapshot variable -- against the REFERENCE PROGRAM
(oracle/_ref/ramses3d_mhd, the unmodified SOLVER=mhd build) run live beside the patched one on a magnetised blast, and
against the strict build through the C ABI on a sheared, magnetised box, where the structural properties are checked too:
div B stays at rounding and the right-face field of a cell stays the left-face field of its neighbour bit for bit
(every EMF is computed once and shared by the four faces around its edge in either build).  The strict build stays the
default of the MHD drop-in; the bit-for-bit tests (tests/test_mhd_gpu.py, tests/test_mhd_dropin_gpu.py) are unchanged."""
import os
import shutil
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
REF = os.path.join(ROOT, "oracle", "_ref", "ramses3d_mhd")
PATCHED = os.path.join(ROOT, "oracle", "_ref", "ramses3d_patch_mhd_mhd")
TOL = 1e-12


def _rel(got, ref):
    """rel-Linf per snapshot variable (rho, u, v, w, the six face fields, P); velocities and fields share one scale each"""
    scale = np.abs(ref).reshape(ref.shape[0], -1).max(axis=1)
    scale[1:4] = scale[1:4].max()
    scale[4:10] = scale[4:10].max()
    return np.abs(got - ref).reshape(ref.shape[0], -1).max(axis=1) / np.maximum(scale, 1e-300)


@pytest.mark.parametrize("level,nstep,riemann,riemann2d,slope_type", [
    (6, 12, "hlld", "hlld", 2), (5, 40, "hlld", "hlld", 2), (5, 40, "llf", "llf", 1), (5, 30, "hll", "hll", 8), (5, 30, "roe", "roe", 1),
])
def test_fast_mhd_program_within_1e12_of_the_reference(gpu_lib, monkeypatch, level, nstep, riemann, riemann2d, slope_type):
    if not (os.path.exists(REF) and os.path.exists(PATCHED)):
        pytest.skip("oracle/_ref/ramses3d_mhd / ramses3d_patch_mhd_mhd not built")
    from mhd_common import mhd_namelist
    from oracle import ramses_snapshot as rs
    nml = mhd_namelist(level, nstep, riemann, riemann2d, slope_type)
    monkeypatch.setenv("RAMSES_AMD", "1")
    monkeypatch.setenv("RAMSES_AMD_MHD_RESIDENT", "1")
    monkeypatch.setenv("RAMSES_AMD_MHD_FAST", "1")
    work, out = rs.run_reference(nml, binary=PATCHED)
    try:
        assert "the MHD level stays resident on the GPU" in out, out[-1500:]
        got = rs.load_uniform_level(os.path.join(work, "output_00002"), level)
    finally:
        shutil.rmtree(work, ignore_errors=True)
    monkeypatch.setenv("RAMSES_AMD", "0")
    monkeypatch.delenv("RAMSES_AMD_MHD_FAST")
    work, out = rs.run_reference(nml, binary=REF)
    try:
        ref = rs.load_uniform_level(os.path.join(work, "output_00002"), level)
    finally:
        shutil.rmtree(work, ignore_errors=True)
    assert int(np.ravel(got["info"]["nstep"])[0]) == int(np.ravel(ref["info"]["nstep"])[0]) == nstep
    tr = float(np.ravel(ref["info"]["t"])[0])
    assert abs(float(np.ravel(got["info"]["t"])[0]) - tr) <= TOL * tr
    assert np.abs(ref["prim"][4:7]).max() > 0.5          # a magnetised run
    assert not np.array_equal(got["prim"], ref["prim"])   # (the fast build did run: it is not the reference's bits)
    err = _rel(got["prim"], ref["prim"])
    print("fast MHD vs the live reference, %d^3, %s/%s slope %d, %d steps: rel-Linf = %s" % (2 ** level, riemann, riemann2d, slope_type, nstep, err))
    assert (err <= TOL).all(), err


@pytest.mark.parametrize("riemann,riemann2d,slope_type", [("hlld", "hlld", 2), ("llf", "llf", 1), ("hll", "hlla", 7), ("upwind", "upwind", 2)])
def test_fast_brick_against_strict_brick_and_div_b(gpu_lib, monkeypatch, riemann, riemann2d, slope_type):
    import torch
    from ramses_amd.mhd import MhdLevel, make_mhd_params
    from test_mhd_gpu import mhd_ic
    n, gamma, nstep = 32, 5.0 / 3.0, 12
    u0 = mhd_ic(n, gamma)
    dt = 0.1 / n
    res = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("RAMSES_AMD_MHD_FAST", mode)
        lev = MhdLevel(n, n, n, 1.0 / n, params=make_mhd_params(gamma=gamma, slope_type=slope_type, riemann=riemann, riemann2d=riemann2d))
        lev.upload(u0)
        for _ in range(nstep):
            lev.step(dt)        # (the entry point itself refuses a level whose right faces are not its neighbours' left faces)
        torch.cuda.synchronize()
        res[mode] = lev.download()
    strict, fast = res["0"], res["1"]
    assert not np.array_equal(strict, fast)
    scale = np.abs(strict).reshape(11, -1).max(axis=1)
    scale[1:4] = scale[1:4].max()
    scale[5:11] = scale[5:11].max()
    err = np.abs(fast - strict).reshape(11, -1).max(axis=1) / scale
    print("fast vs strict brick, %s/%s slope %d, %d steps: rel-Linf = %s" % (riemann, riemann2d, slope_type, nstep, err))
    assert (err <= TOL).all(), err
    # right face of a cell == left face of its neighbour, bit for bit; div B at rounding
    for c, ax in ((0, 2), (1, 1), (2, 0)):
        assert np.array_equal(fast[8 + c], np.roll(fast[5 + c], -1, axis=ax))
    div = sum((fast[8 + c] - fast[5 + c]) for c in range(3)) * n
    assert np.abs(div).max() < 1e-9
