"""bench.py builds sedov3d.nml's initial state at 512^3 on the device from two cell values (the corner cell and the
background): that is the reference's initial state, not a simplification -- region_condinit's CIC cloud of the 'point'
region at the box corner is clipped by the box (its weights are not periodic, hydro/init_flow_fine.f90:555-594), so
only cell (0,0,0) receives the blast energy."""
import numpy as np
import pytest


@pytest.mark.parametrize("n", [16, 64])
def test_sedov3d_is_background_plus_the_corner_cell(n):
    from ramses_amd import ic
    u, dx = ic.sedov3d(n)
    corner, back, dx2 = ic.sedov3d_corner_and_background(n)
    assert dx == dx2
    v = np.empty_like(u)
    v[:] = back[:, None, None, None]
    v[:, 0, 0, 0] = corner
    assert np.array_equal(u, v)
    # the blast energy: p_region * (1/2)^3 / dx^3 on top of the background pressure, in one cell
    assert corner[4] > 1e3 * back[4] and np.count_nonzero(u[4] != back[4]) == 1


def test_the_reference_program_starts_from_the_same_state():
    """output_00001 of the unmodified reference (sedov3d.nml at 32^3) == ic.sedov3d(32), bit for bit"""
    import os
    import shutil
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    binary = os.path.join(ROOT, "oracle", "_ref", "ramses3d")
    if not os.path.exists(binary):
        pytest.skip("oracle/_ref/ramses3d not built")
    from oracle import ramses_snapshot as rs
    from ramses_amd import ic
    work, _ = rs.run_reference(rs.sedov3d_namelist(level=5, nstepmax=1, foutput=1), binary=binary)
    try:
        snap = rs.load_uniform_level(os.path.join(work, "output_00001"), 5)
    finally:
        shutil.rmtree(work, ignore_errors=True)
    u, _ = ic.sedov3d(32)
    rho, p = snap["prim"][0], snap["prim"][4]
    assert np.array_equal(rho, u[0])
    assert np.count_nonzero(p != p[1, 1, 1]) == 1 and p[0, 0, 0] == p.max()
    assert np.array_equal(p, (u[4] - 0.0) * (1.4 - 1.0)) or np.allclose(p, u[4] * 0.4, rtol=1e-15)
