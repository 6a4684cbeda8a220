"""CPU tests of the drop-in boundary: the C-ABI library loads and exports every
symbol include/ramses_amd.h declares; argument validation fails loudly.
No compute entry point is executed without a GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built_lib():
    from ramses_amd import build
    path = build.build()
    assert os.path.exists(path)
    return C.CDLL(path)


def _declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "ramses_amd.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(ramses_amd_[a-z0-9_]+)\s*\(", hdr)))


def test_header_symbols_exported(built_lib):
    names = _declared_symbols()
    assert len(names) >= 10
    for n in names:
        assert hasattr(built_lib, n), "libramses_amd.so does not export %s" % n


def test_python_binding_covers_header():
    from ramses_amd import _capi
    bound = {s[0] for s in _capi.SYMBOLS}
    assert bound == set(_declared_symbols())


def test_struct_layout_matches_header(built_lib):
    from ramses_amd import _capi
    # ramses_amd_brick_dense fills pitches: checks the Brick layout end to end
    b = _capi.dense_brick(10, 6, 4, 2)
    assert (b.nx, b.ny, b.nz, b.ng) == (10, 6, 4, 2)
    assert b.pitch_y == 14 and b.pitch_z == 14 * 10 and b.pitch_var == 14 * 10 * 8
    assert built_lib.ramses_amd_abi_check(C.c_size_t(C.sizeof(_capi.HydroParams)),
                                          C.c_size_t(C.sizeof(_capi.Brick))) == 0
    assert built_lib.ramses_amd_abi_check(C.c_size_t(8), C.c_size_t(8)) != 0


def test_argument_validation_is_loud():
    from ramses_amd import _capi
    L = _capi.lib()
    p = _capi.make_params(ndim=2)
    b = _capi.dense_brick(8, 8, 8, 0)
    rc = L.ramses_amd_godunov_brick(C.byref(p), C.byref(b), C.c_void_p(8), None, C.c_void_p(16), 0.1, 0.1, None)
    # NDIM=2 is an embedded problem: it needs nz=1 and ghost layers, anything else is refused
    assert rc == -1 and b"NDIM=2 needs a brick with nz=1" in L.ramses_amd_last_error()
    p = _capi.make_params(difmag=0.1)
    rc = L.ramses_amd_godunov_brick(C.byref(p), C.byref(b), C.c_void_p(8), None, C.c_void_p(16), 0.1, 0.1, None)
    assert rc == -2
    p = _capi.make_params()
    rc = L.ramses_amd_godunov_brick(C.byref(p), C.byref(b), C.c_void_p(8), None, C.c_void_p(8), 0.1, 0.1, None)
    assert rc == -1  # uold == unew
    with pytest.raises(_capi.RamsesAmdError):
        _capi.check(rc)


def test_a_brick_beyond_the_32_bit_offsets_of_the_sweep_is_refused_by_name():
    """hydro_sweep.hip addresses a plane with a 32-bit lane offset and a variable's planes with a 32-bit scalar offset: a brick
    beyond that (a 2 GiB plane, a 4 GiB variable) must be refused loudly before anything is launched"""
    from ramses_amd import _capi
    L = _capi.lib()
    p = _capi.make_params()
    for dims in ((900, 900, 900), (20000, 20000, 4)):          # 5.8 GiB per variable; 3.2 GB planes
        b = _capi.dense_brick(dims[0], dims[1], dims[2], 0)
        rc = L.ramses_amd_godunov_brick(C.byref(p), C.byref(b), C.c_void_p(8), None, C.c_void_p(16), 0.1, 0.1, None)
        msg = L.ramses_amd_last_error()
        assert rc == -2 and b"beyond the 32-bit offsets of the sweep kernel" in msg and b"split the level" in msg, (rc, msg)
    b = _capi.dense_brick(512, 512, 512, 0)                      # the bench's brick: 1 GiB per variable, fine
    rc = L.ramses_amd_godunov_brick(C.byref(p), C.byref(b), None, None, None, 0.1, 0.1, None)
    assert rc == -1 and b"NULL" in L.ramses_amd_last_error()


def test_no_gpu_means_no_silent_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from ramses_amd import RamsesAmdError
    from ramses_amd.hydro import HydroLevel
    with pytest.raises(RamsesAmdError):
        HydroLevel(8, 8, 8, 0.1)


def test_which_column_recognises_an_anonymous_array_by_address():
    """make_virtual_fine_dp(xx,ilevel) receives an anonymous array (amr/amr_step.f90:61,397,505: uold(1,ivar),
    unew(1,ivar)); the shim asks which column of a module array it is.  Host pointers only: no GPU needed."""
    import numpy as np
    from ramses_amd import _capi
    L = _capi.lib()
    ncell, nvar = 1000, 5
    base = np.zeros(ncell * nvar)
    other = np.zeros(ncell)
    p = base.ctypes.data
    for ivar in range(nvar):
        assert L.ramses_amd_which_column(C.c_void_p(p + 8 * ncell * ivar), C.c_void_p(p), ncell, nvar) == ivar + 1
    assert L.ramses_amd_which_column(C.c_void_p(p + 8), C.c_void_p(p), ncell, nvar) == 0              # inside a column
    assert L.ramses_amd_which_column(C.c_void_p(p + 8 * ncell * nvar), C.c_void_p(p), ncell, nvar) == 0  # past the end
    assert L.ramses_amd_which_column(C.c_void_p(other.ctypes.data), C.c_void_p(p), ncell, nvar) == 0
    assert L.ramses_amd_which_column(None, C.c_void_p(p), ncell, nvar) == 0
    # no communicators on the device before ramses_amd_amrres_comm_set
    assert L.ramses_amd_amrres_comm_epoch(3) == -1
    assert L.ramses_amd_amrres_comm_set(0, 1, 2, None, None, None, None) != 0
    assert b"comm_set" in L.ramses_amd_last_error()


def test_mpi_entry_points_fail_loudly_without_their_state():
    """The stepwise entry points of the MPI paths refuse to run out of order (before any device call)."""
    from ramses_amd import _capi
    L = _capi.lib()
    assert L.ramses_amd_amrres_zero_unew_virtual(3) != 0
    assert b"no resident AMR state" in L.ramses_amd_last_error()
    assert L.ramses_amd_amrres_halo_stage_in(3, 0) != 0
    assert L.ramses_amd_amrres_halo_rccl(3, 1, 1) != 0
    assert L.ramses_amd_cgmpi_step(0, 1) != 0
    assert b"no CG solve is open" in L.ramses_amd_last_error()
    v = C.c_double(0.0)
    assert L.ramses_amd_cgmpi_get(0, C.byref(v)) != 0
    assert L.ramses_amd_cgmpi_set(0, 1.0) != 0


def test_mgdist_oct_box_finds_the_box_a_rank_fills(built_lib):
    """ramses_amd_mgdist_oct_box (host only): the octs of a rank's domain, in any order, fill a box -- or the call says
    they do not (the Fortran shim then keeps the multigrid of AMR levels)."""
    import ctypes as C
    from ramses_amd import _capi
    L = _capi.lib()
    level, n = 4, 16
    ngridmax = 700
    rng = np.random.default_rng(2)
    # a quarter column: cells [8,16) x [0,8) x [0,16)  ->  octs 4 x 4 x 8
    octs = [(i, j, k) for k in range(0, 16, 2) for j in range(0, 8, 2) for i in range(8, 16, 2)]
    slots = rng.permutation(ngridmax)[:len(octs)] + 1
    xg = np.zeros((3, ngridmax))
    for s, (i, j, k) in zip(slots, octs):
        xg[:, s - 1] = [(i + 1) / n, (j + 1) / n, (k + 1) / n]
    igrid = slots.astype(np.int32)
    lo, dims = (C.c_int * 3)(), (C.c_int * 3)()
    rc = L.ramses_amd_mgdist_oct_box(level, len(octs), igrid.ctypes.data_as(C.c_void_p), xg.ctypes.data_as(C.c_void_p), ngridmax, lo, dims)
    assert rc == 0, L.ramses_amd_last_error()
    assert tuple(lo) == (8, 0, 0) and tuple(dims) == (8, 8, 16)
    # one oct missing: not a box
    rc = L.ramses_amd_mgdist_oct_box(level, len(octs) - 1, igrid.ctypes.data_as(C.c_void_p), xg.ctypes.data_as(C.c_void_p), ngridmax, lo, dims)
    assert rc != 0 and b"do not fill" in L.ramses_amd_last_error()
    # a rank grid that is not a power of two, a brick below the smoother's tile: loud
    ctx = C.c_void_p()
    assert L.ramses_amd_mgdist_create(7, (C.c_int * 3)(3, 1, 1), 0, None, None, C.byref(ctx)) != 0
    assert L.ramses_amd_mgdist_create(7, (C.c_int * 3)(4, 1, 1), 0, None, None, C.byref(ctx)) != 0
    assert b"every extent must be >= 64" in L.ramses_amd_last_error()


def test_mgdist_fortran_entries_refuse_to_run_without_a_context(built_lib):
    """ramses_amd_mgdist_multigrid_f90 / _force_f90 without a context (or with NULL arrays): an error code and a message,
    never a crash or a silent return."""
    from ramses_amd import _capi
    L = _capi.lib()
    buf = (C.c_double * 8)()
    ints = (C.c_int * 8)()
    it, err, safe = C.c_int(), C.c_double(), C.c_int()
    rc = L.ramses_amd_mgdist_multigrid_f90(None, 7, 8, ints, buf, 100, 1, ints, buf, buf, 1.0, 1.0, 1e-4,
                                           C.byref(safe), C.byref(it), C.byref(err))
    assert rc != 0 and L.ramses_amd_last_error()
    rc = L.ramses_amd_mgdist_force_f90(None, 7, 8, ints, buf, 100, 1, ints, buf, buf, ints, 32, 1.0, buf)
    assert rc != 0 and L.ramses_amd_last_error()
    assert L.ramses_amd_mgdist_destroy(None) == 0
