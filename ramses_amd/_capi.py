"""ctypes binding of libramses_amd.so (the C ABI in include/ramses_amd.h).

The library is the product: if it is missing or cannot be loaded this module
raises -- there is no Python/CPU fallback for any compute entry point.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# RAMSES_AMD_LIB: load another build of the same library (kernel A/B measurements)
LIB_PATH = os.environ.get("RAMSES_AMD_LIB") or os.path.join(HERE, "lib", "libramses_amd.so")

RIEMANN = {"llf": 0, "hllc": 1, "hll": 2, "acoustic": 3, "exact": 4}
SCHEME = {"muscl": 0, "plmde": 1}


class RamsesAmdError(RuntimeError):
    pass


class HydroParams(C.Structure):
    """struct ramses_amd_hydro_params (= &HYDRO_PARAMS, hydro/hydro_parameters.f90:75-89)."""
    _fields_ = [
        ("ndim", C.c_int32), ("nvar", C.c_int32),
        ("gamma", C.c_double), ("smallr", C.c_double), ("smallc", C.c_double),
        ("slope_type", C.c_int32), ("riemann", C.c_int32),
        ("slope_theta", C.c_double),
        ("scheme", C.c_int32), ("niter_riemann", C.c_int32),
        ("difmag", C.c_double), ("courant_factor", C.c_double),
        ("fast_math", C.c_int32), ("reserved", C.c_int32),
    ]


class Brick(C.Structure):
    """struct ramses_amd_brick."""
    _fields_ = [
        ("nx", C.c_int32), ("ny", C.c_int32), ("nz", C.c_int32), ("ng", C.c_int32),
        ("pitch_y", C.c_int64), ("pitch_z", C.c_int64), ("pitch_var", C.c_int64),
    ]


_lib = None

# every symbol include/ramses_amd.h declares: (name, restype, argtypes)
_vp, _i, _d, _i64 = C.c_void_p, C.c_int, C.c_double, C.c_int64
_PP, _PB = C.POINTER(HydroParams), C.POINTER(Brick)
SYMBOLS = [
    ("ramses_amd_last_error", C.c_char_p, []),
    ("ramses_amd_abi_check", _i, [C.c_size_t, C.c_size_t]),
    ("ramses_amd_brick_dense", None, [_PB, _i, _i, _i, _i]),
    ("ramses_amd_device_info", _i, [C.c_char_p, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_size_t)]),
    ("ramses_amd_set_device_auto", _i, [_i]),
    ("ramses_amd_godunov_brick", _i, [_PP, _PB, _vp, _vp, _vp, _d, _d, _vp]),
    ("ramses_amd_godunov_brick_shell", _i, [_PP, _PB, _vp, _vp, _vp, _d, _d, _vp]),
    ("ramses_amd_godunov_brick_interior", _i, [_PP, _PB, _vp, _vp, _vp, _d, _d, _vp]),
    ("ramses_amd_godunov_tune", _i, [_i, _i]),
    ("ramses_amd_courant_init", _i, [_PP, _d, _vp, _vp]),
    ("ramses_amd_courant_brick", _i, [_PP, _PB, _vp, _vp, _d, _vp, _vp]),
    ("ramses_amd_fill_ghosts_periodic", _i, [_PB, _vp, _i, _i, _vp]),
    ("ramses_amd_halo_slab_size", _i64, [_PB, _i, _i]),
    ("ramses_amd_halo_pack", _i, [_PB, _vp, _i, _i, _vp, _vp]),
    ("ramses_amd_halo_unpack", _i, [_PB, _vp, _i, _i, _vp, _vp]),
    ("ramses_amd_mg_workspace_doubles", _i64, [_i]),
    ("ramses_amd_multigrid_fine_brick", _i, [_i, _vp, _d, _d, _d, C.POINTER(C.c_int), _vp, _vp, _vp, _vp,
                                             C.POINTER(C.c_int), C.POINTER(C.c_double), _vp]),
    ("ramses_amd_gradient_phi_brick", _i, [_i, _vp, _vp, _vp]),
    ("ramses_amd_mg_gauss_seidel", _i, [_vp, _vp, _i, _d, _i, _vp]),
    ("ramses_amd_mg_residual", _i, [_vp, _vp, _vp, _i, _d, _vp, _vp, _vp]),
    ("ramses_amd_mg_restrict", _i, [_vp, _vp, _vp, _i, _vp]),
    ("ramses_amd_mg_interp_correct", _i, [_vp, _vp, _i, _vp]),
    ("ramses_amd_mg_smooth_fused", _i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _d, _i, _vp]),
    ("ramses_amd_mg_tune", _i, [_i]),
    ("ramses_amd_godunov_fine_host", _i, [_PP, _i, _i, _vp, _vp, _i64, _i64, _i, _vp, _vp, _vp, _d, _d]),
    ("ramses_amd_interpol_hydro_brick", _i, [_i, _i, _i, _i, _d, _vp, _vp, _vp]),
    ("ramses_amd_upload_fine_brick", _i, [_i, _i, _i, _d, _vp, _vp, _vp]),
    ("ramses_amd_multigrid_fine_f90", _i, [_i, _i, _vp, _vp, _i64, _i64, _i, _vp, _vp, _d, _d, _d,
                                           C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_double)]),
    ("ramses_amd_godunov_fine_f90", _i, [_PP, _i, _i, _vp, _vp, _i64, _i64, _i, _vp, _vp, _vp, _i, _d, _d]),
    ("ramses_amd_mg_smooth_fused_ghost", _i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _d, _i, _vp]),
    ("ramses_amd_godunov_fine_amr_host", _i, [_PP, _i, _i, _vp, _vp, _vp, _vp, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _d, _d, _i, _i, _i]),
    ("ramses_amd_godunov_fine_amr_f90", _i, [_PP, _i, _i, _vp, _vp, _vp, _vp, _i64, _i64, _vp, _vp, _vp, _i, _vp, _vp, _i, _d, _d, _i, _i, _i]),
    ("ramses_amd_godunov_fine_amr_workspace", _i64, [_i, _i64]),
    ("ramses_amd_godunov_fine_amr_device", _i, [_PP, _i, _i, _vp, _vp, _vp, _vp, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _d, _d, _i, _i, _i, _vp, _vp, _vp]),
    ("ramses_amd_halo_multi", _i, [_PB, _vp, _i, _i, _vp, _vp, _vp, _i, _vp]),
    ("ramses_amd_make_boundary_hydro", _i, [_PP, _PB, _vp, _i, _i, _vp, _i, _vp]),
    ("ramses_amd_mg_rhs", _i, [_vp, _vp, _i64, _d, _d, _vp]),
    ("ramses_amd_mg_restrict_ghost", _i, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    ("ramses_amd_mg_interp_correct_ghost", _i, [_vp, _i, _i, _i, _i, _vp, _i, _i, _vp, _vp]),
    ("ramses_amd_gradient_phi_ghost", _i, [_vp, _vp, _i, _i, _i, _i, _d, _vp]),
    ("ramses_amd_mg_coarse_solve_dense", _i, [_i, _vp, _vp, _vp, _i, _vp]),
    ("ramses_amd_mgamr_begin", _i, [_i, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp]),
    ("ramses_amd_mgamr_add_level", _i, [_i, _i, _vp, _vp, _vp]),
    ("ramses_amd_mgamr_level_begin", _i, [_i, _i]),
    ("ramses_amd_mgamr_level_block", _i, [_i, _i, _vp, _vp, _vp]),
    ("ramses_amd_mgamr_fine_active", _i, [_i]),
    ("ramses_amd_mgamr_force_sync", _i, [_i]),
    ("ramses_amd_mgamr_gauss_seidel", _i, [_i, _i, _i]),
    ("ramses_amd_mgamr_residual", _i, [_i]),
    ("ramses_amd_mgamr_norm2", _i, [_i, _vp]),
    ("ramses_amd_mgamr_restrict", _i, [_i]),
    ("ramses_amd_mgamr_interpolate", _i, [_i]),
    ("ramses_amd_mgamr_end", _i, []),
    ("ramses_amd_mgamr_comm_set", _i, [_i, _i, _i, _vp, _vp, _i, _vp]),
    ("ramses_amd_mgamr_halo_stage_out", _i, [_i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    ("ramses_amd_mgamr_halo_stage_in", _i, [_i, _i, _i]),
    ("ramses_amd_mgamr_halo_rccl", _i, [_i, _i, _i]),
    ("ramses_amd_mgamr_stats", _i, [_vp]),
    ("ramses_amd_poisamr_tree", _i, [_i, _i64, _i64, _vp, _vp, _vp]),
    ("ramses_amd_poisamr_multigrid", _i, [_i, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _d, _d, _d, _i, _d, _i, _i, _i, _vp, _vp, _vp]),
    ("ramses_amd_poisamr_levelmin_mg", _i, []),
    ("ramses_amd_poisamr_force", _i, [_i, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _d, _i, _i, _d, _vp]),
    ("ramses_amd_poisamr_force_mpi", _i, [_i, _i, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _d, _i, _d, _vp]),
    ("ramses_amd_poisamr_force_mpi_resident", _i, [_i, _i, _i, _vp, _i, _vp, _vp, _vp, _vp, _d, _i, _d, _vp]),
    ("ramses_amd_prof_add", _i, [C.c_char_p, _i, _d]),
    ("ramses_amd_warmup", _i, []),
    ("ramses_amd_host_register", _i, [_vp, _i64]),
    ("ramses_amd_resident_synchro_f90", _i, [_PP, _i, _i, _vp, _vp, _i64, _i64, _i, _vp, _vp, _d]),
    ("ramses_amd_resident_courant_grav_f90", _i, [_PP, _i, _i, _vp, _vp, _i64, _i64, _i, _vp, _vp, _d, _d, _vp]),
    ("ramses_amd_resident_godunov_grav_f90", _i, [_PP, _i, _i, _vp, _vp, _i64, _i64, _i, _vp, _vp, _d, _d]),
    ("ramses_amd_resident_set_uold_grav_f90", _i, [_PP, _i, _d]),
    ("ramses_amd_resident_sync_density_f90", _i, [_vp]),
    ("ramses_amd_force_fine_f90", _i, [_i, _i, _vp, _vp, _i64, _i64, _i, _vp, _vp, _vp, _vp, _i, _d, _vp]),
    ("ramses_amd_resident_rho_fine_f90", _i, [_PP, _i, _i, _vp, _vp, _i64, _i64, _i, _vp, _d, _i, _vp]),
    ("ramses_amd_resident_multigrid_f90", _i, [_i, _d, _d, _d, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_double)]),
    ("ramses_amd_resident_force_fine_f90", _i, [_i, _d, _vp]),
    ("ramses_amd_resident_sync_poisson_f90", _i, [_vp, _vp, _vp]),
    ("ramses_amd_cg_solve_host", _i, [_i, _i, _vp, _vp, _vp, _i64, _i64, _vp, _vp, _vp, _d, _d, _d, _d, _i, _i, _vp, _vp]),
    ("ramses_amd_ordered_sum_scratch", C.c_size_t, [_i64]),
    ("ramses_amd_ordered_sum_device", _i, [_vp, _i64, _vp, _vp, _vp]),
    ("ramses_amd_cgmpi_begin", _i, [_i, _i, _vp, _vp, _vp, _i64, _i64, _vp, _vp, _vp, _d, _d, _i, _vp]),
    ("ramses_amd_cgmpi_get", _i, [_i, _vp]),
    ("ramses_amd_cgmpi_set", _i, [_i, _d]),
    ("ramses_amd_cgmpi_step", _i, [_i, _i]),
    ("ramses_amd_cgmpi_p_cells", _i, [_i, _vp, _i]),
    ("ramses_amd_cgmpi_end", _i, [_vp, _vp]),
    ("ramses_amd_cgmpi_comm_set", _i, [_i, _vp, _vp, _vp, _vp]),
    ("ramses_amd_cgmpi_p_halo_stage_out", _i, [_i, _vp, _vp, _vp, _vp]),
    ("ramses_amd_cgmpi_p_halo_stage_in", _i, []),
    ("ramses_amd_cgmpi_p_halo_rccl", _i, []),
    ("ramses_amd_resident_courant_f90", _i, [_PP, _i, _i, _vp, _vp, _i64, _i64, _i, _vp, _d, _d, _vp]),
    ("ramses_amd_resident_godunov_f90", _i, [_PP, _i, _i, _vp, _vp, _i64, _i64, _i, _vp, _d, _d]),
    ("ramses_amd_resident_set_uold_f90", _i, [_i]),
    ("ramses_amd_resident_sync_host_f90", _i, [_vp]),
    ("ramses_amd_resident_invalidate", _i, []),
    # MPI: one rank per GPU
    ("ramses_amd_device_uid", _i, [_vp]),
    ("ramses_amd_rccl_probe", _i, []),
    ("ramses_amd_rccl_unique_id", _i, [_vp]),
    ("ramses_amd_rccl_init", _i, [_vp, _i, _i]),
    ("ramses_amd_rccl_ready", _i, []),
    ("ramses_amd_rccl_finalize", _i, []),
    ("ramses_amd_rccl_exchange", _i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    ("ramses_amd_rccl_sendrecv", _i, [_i, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp]),
    ("ramses_amd_rccl_allreduce", _i, [_vp, _i, _i, _vp]),
    ("ramses_amd_rccl_allgather", _i, [_vp, _i64, _vp, _vp]),
    ("ramses_amd_mgdist_create", _i, [_i, _vp, _i, _vp, _vp, _vp]),
    ("ramses_amd_mgdist_destroy", _i, [_vp]),
    ("ramses_amd_mgdist_info", _i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    ("ramses_amd_mgdist_set_safe_mode", _i, [_vp, _i]),
    ("ramses_amd_mgdist_set_order", _i, [_vp, _vp, _i64]),
    ("ramses_amd_mgdist_solve", _i, [_vp, _vp, _d, _d, _d, _vp, _vp, _vp]),
    ("ramses_amd_mgdist_get_phi", _i, [_vp, _vp, _vp]),
    ("ramses_amd_mgdist_set_phi", _i, [_vp, _vp, _vp]),
    ("ramses_amd_mgdist_force", _i, [_vp, _vp, _vp]),
    ("ramses_amd_mgdist_oct_box", _i, [_i, _i, _vp, _vp, _i64, _vp, _vp]),
    ("ramses_amd_mgdist_plan", _i, [_vp, _i, _vp, _vp, _i] + [_vp] * 11),
    ("ramses_amd_mgdist_force_f90", _i, [_vp, _i, _i, _vp, _vp, _i64, _i64, _vp, _vp, _vp, _vp, _i, _d, _vp]),
    ("ramses_amd_mgdist_force_resident_f90", _i, [_vp, _i, _i, _vp, _i64, _i64, _vp, _i, _d, _vp]),
    ("ramses_amd_mgdist_multigrid_resident_f90", _i, [_vp, _i, _i, _vp, _vp, _i64, _vp, _d, _d, _d, _vp, _vp, _vp]),
    ("ramses_amd_mgdist_fetch_phi_f90", _i, [_vp, _i, _vp, _i64, _i64, _vp]),
    ("ramses_amd_mgdist_traffic", _i, [_vp]),
    ("ramses_amd_mgdist_force_resident_dev_f90", _i, [_vp, _i, _i, _vp, _i64, _i64, _i, _d, _vp]),
    ("ramses_amd_mgdist_multigrid_f90", _i, [_vp, _i, _i, _vp, _vp, _i64, _i64, _vp, _vp, _vp, _d, _d, _d, _vp, _vp, _vp]),
    ("ramses_amd_halo_plan", _i, [_i, _i, _vp, _vp, _i64, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64]),
    ("ramses_amd_mpires_setup", _i, [_PP, _i, _i, _vp, _vp, _i64, _i64, _i, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp]),
    ("ramses_amd_mpires_active", _i, []),
    ("ramses_amd_mpires_which", _i, [_vp]),
    ("ramses_amd_mpires_courant", _i, [_PP, _d, _d, _vp]),
    ("ramses_amd_mpires_godunov", _i, [_PP, _d, _d]),
    ("ramses_amd_mpires_reverse_unew", _i, []),
    ("ramses_amd_mpires_set_uold", _i, []),
    ("ramses_amd_mpires_halo_forward", _i, []),
    ("ramses_amd_mpires_halo_stage_out", _i, [_vp, _vp, _vp, _vp]),
    ("ramses_amd_mpires_halo_stage_out_f90", _i, [_vp, _vp, _vp, _vp, _i]),
    ("ramses_amd_mpires_halo_stage_in", _i, []),
    ("ramses_amd_mpires_sync_host", _i, [_vp]),
    ("ramses_amd_mpires_invalidate", _i, []),
    # residency for AMR runs
    ("ramses_amd_godunov_fine_lowdim_f90", _i, [_PP, _i, _i, _vp, _i, _vp, _vp, _i64, _i64, _vp, _vp, _vp, _vp, _d, _d]),
    ("ramses_amd_lowdim_note_reference", _i, [_i]),
    ("ramses_amd_lowdim_device_sweeps", _i64, []),
    ("ramses_amd_lowdim_reference_sweeps", _i64, []),
    ("ramses_amd_amrres_active", _i, []),
    ("ramses_amd_amrres_load", _i, [_i, _i64, _i64, _vp, _vp, _vp, _vp]),
    ("ramses_amd_amrres_tree", _i, [_vp, _vp, _vp]),
    ("ramses_amd_amrres_first_changed", _i, []),
    ("ramses_amd_amrres_invalidate", _i, []),
    ("ramses_amd_amrres_sync_level", _i, [_i, _vp, _vp]),
    ("ramses_amd_amrres_load_level", _i, [_i, _vp, _vp]),
    ("ramses_amd_amrres_sync_all", _i, [_vp]),
    ("ramses_amd_amrres_set_unew", _i, [_i, _vp]),
    ("ramses_amd_amrres_set_uold", _i, [_PP, _i, _vp]),
    ("ramses_amd_amrres_upload_fine", _i, [_PP, _i, _vp, _i]),
    ("ramses_amd_amrres_courant", _i, [_PP, _i, _vp, _d, _d, _vp]),
    ("ramses_amd_amrres_load_f", _i, [_i, _vp, _vp]),
    ("ramses_amd_amrres_take_f_device", _i, [_i, _vp, _vp]),
    ("ramses_amd_amrres_sync_f", _i, [_i, _vp, _vp]),
    ("ramses_amd_amrres_f_traffic", _i, [_vp]),
    ("ramses_amd_amrres_compare_f", _i, [_i, _vp, _vp, _vp, _vp]),
    ("ramses_amd_amrres_rho_keep", _i, [_i]),
    ("ramses_amd_amrres_sync_rho", _i, [_i, _vp, _vp]),
    ("ramses_amd_amrres_rho_to_brick", _i, [_i, _vp, _vp, _vp]),
    ("ramses_amd_amrres_rho_absmax", _i, [_i, _vp, _vp]),
    ("ramses_amd_amrres_rho_traffic", _i64, []),
    ("ramses_amd_amrres_has_gravity", _i, []),
    ("ramses_amd_amrres_sync_density", _i, [_i, _vp, _vp]),
    ("ramses_amd_amrres_synchro", _i, [_PP, _i, _vp, _d]),
    ("ramses_amd_amrres_set_uold_grav", _i, [_PP, _i, _vp, _d]),
    ("ramses_amd_amrres_enable_pfix", _i, []),
    ("ramses_amd_amrres_set_unew_pfix", _i, [_PP, _i, _vp]),
    ("ramses_amd_amrres_set_uold_pfix", _i, [_PP, _i, _vp, _d, _d, _d, _d]),
    ("ramses_amd_amrres_xg", _i, [_vp]),
    ("ramses_amd_amrres_rho_fine", _i, [_PP, _i, _i, _i, _i, _vp, _vp, _d, _vp, _vp]),
    ("ramses_amd_amrres_covered_sweeps", _i64, []),
    ("ramses_amd_amrres_tile_sweeps", _i64, []),
    ("ramses_amd_amrres_tree_sweeps", _i64, []),
    ("ramses_amd_amrres_relayouts", _i64, []),
    ("ramses_amd_amrres_tiled_levels", _i, []),
    ("ramses_amd_amrres_boundary_hydro", _i, [_i, _vp, _vp, _vp, _i, _d, _i, _vp]),
    ("ramses_amd_mhd_workspace_bytes", _i64, [_i, _i, _i]),
    ("ramses_amd_mhd_godunov_brick", _i, [_vp, _i, _i, _i, _vp, _vp, _d, _d, _vp, _i64, _vp]),
    ("ramses_amd_mhd_godunov_brick_fast", _i, [_vp, _i, _i, _i, _vp, _vp, _d, _d, _vp, _i64, _vp]),
    ("ramses_amd_mhd_godunov_fine_f90", _i, [_vp, _i, _i, _vp, _vp, _i64, _i64, _i, _vp, _vp, _d, _d]),
    ("ramses_amd_mhd_resident_active", _i, []),
    ("ramses_amd_mhd_resident_courant_f90", _i, [_vp, _i, _i, _vp, _vp, _i64, _i64, _i, _vp, _d, _d, _d, _vp]),
    ("ramses_amd_mhd_resident_godunov_f90", _i, [_vp, _i, _i, _vp, _vp, _i64, _i64, _i, _vp, _d, _d]),
    ("ramses_amd_mhd_resident_set_uold_f90", _i, [_i]),
    ("ramses_amd_mhd_resident_sync_host_f90", _i, [_vp]),
    ("ramses_amd_mhd_resident_invalidate", _i, []),
    ("ramses_amd_mhd_godfine_amr_device", _i, [_vp, _i, _i, _vp, _vp, _vp, _vp, _i64, _i64, _vp, _vp, _vp, _d, _d, _i, _i, _i, _i, _i, _vp]),
    ("ramses_amd_mhd_godunov_fine_amr_f90", _i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp, _i64, _i64, _vp, _vp, _vp, _i, _d, _d, _i, _i, _i, _i]),
    ("ramses_amd_mhd_note_reference_sweep", _i, [_i]),
    ("ramses_amd_mhd_amr_sweeps", _i64, []),
    ("ramses_amd_mhd_amr_octs", _i64, []),
    ("ramses_amd_amrres_rho_mpi_multipole", _i, [_PP, _i, _i, _i, _vp, _d]),
    ("ramses_amd_amrres_rho_mpi_deposit", _i, [_i, _i, _d]),
    ("ramses_amd_amrres_rho_mpi_finish", _i, [_i, _i, _i, _vp, _vp, _vp]),
    ("ramses_amd_amrres_hydro_flag", _i, [_PP, _i, _vp, _d, _d, _d, _d, _d, _d, _vp, _vp]),
    ("ramses_amd_amrres_godunov", _i, [_PP, _i, _i, _vp, _d, _d, _i, _i, _i]),
    # AMR residency under MPI: the virtual-boundary exchanges on the resident cell vectors
    ("ramses_amd_which_column", _i, [_vp, _vp, _i64, _i]),
    ("ramses_amd_amrres_comm_epoch", _i, [_i]),
    ("ramses_amd_amrres_comm_set", _i, [_i, _i, _i, _vp, _vp, _vp, _vp]),
    ("ramses_amd_amrres_zero_unew_virtual", _i, [_i]),
    ("ramses_amd_amrres_halo_rccl", _i, [_i, _i, _i]),
    ("ramses_amd_amrres_halo_stage_out", _i, [_i, _i, _i, _vp, _vp, _vp, _vp]),
    ("ramses_amd_amrres_halo_stage_in", _i, [_i, _i]),
]


def lib():
    """Load libramses_amd.so; raise loudly when it is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RamsesAmdError(
                "%s is missing: build it with `python -m ramses_amd.build` "
                "(__graft_entry__.build()). There is no CPU fallback." % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        for name, res, args in SYMBOLS:
            f = getattr(L, name)
            f.restype = res
            f.argtypes = args
        if L.ramses_amd_abi_check(C.sizeof(HydroParams), C.sizeof(Brick)) != 0:
            raise RamsesAmdError(L.ramses_amd_last_error().decode())
        _lib = L
    return _lib


def check(rc):
    if rc != 0:
        raise RamsesAmdError("libramses_amd: error %d: %s" % (rc, lib().ramses_amd_last_error().decode()))


def make_params(ndim=3, nvar=None, gamma=1.4, smallr=1e-10, smallc=1e-10, slope_type=1,
                slope_theta=1.5, riemann="llf", scheme="muscl", niter_riemann=10,
                difmag=0.0, courant_factor=0.5, fast_math=False):
    """Defaults are the reference's (hydro/hydro_parameters.f90:75-89)."""
    return HydroParams(ndim, nvar if nvar else ndim + 2, gamma, smallr, smallc, slope_type,
                       RIEMANN[riemann] if isinstance(riemann, str) else riemann, slope_theta,
                       SCHEME[scheme] if isinstance(scheme, str) else scheme, niter_riemann,
                       difmag, courant_factor, 1 if fast_math else 0, 0)


def dense_brick(nx, ny, nz, ng):
    b = Brick()
    lib().ramses_amd_brick_dense(C.byref(b), nx, ny, nz, ng)
    return b
