!==============================================================================
! ramses_amd_mhd_iface.f90 -- ISO_C_BINDING interface of the MHD entry points of libramses_amd.so
! (include/ramses_amd.h: ramses_amd_mhd_params, ramses_amd_mhd_godunov_fine_f90) for a SOLVER=mhd build of RAMSES
! with PATCH=.../ramses_amd/patch_mhd.  RAMSES_AMD=0 in the environment keeps the reference's own routines.
!==============================================================================
module ramses_amd_mhd_iface
  use iso_c_binding
  implicit none
  type, bind(C) :: ramses_amd_mhd_params
     real(c_double) :: gamma, smallr, smallc, slope_theta
     integer(c_int32_t) :: slope_type, slope_mag_type, riemann, riemann2d
  end type ramses_amd_mhd_params
  interface
     function ramses_amd_mhd_godunov_fine_f90(p, ilevel, ngrid, igrid, xg, ngridmax, ncoarse, nx_loc, uold, unew, dx, dt) &
          & bind(C, name='ramses_amd_mhd_godunov_fine_f90') result(rc)
       import :: ramses_amd_mhd_params, c_int, c_int64_t, c_double
       type(ramses_amd_mhd_params), intent(in) :: p
       integer(c_int), value :: ilevel, ngrid, nx_loc
       integer(c_int) :: igrid(*)
       integer(c_int64_t), value :: ngridmax, ncoarse
       real(c_double) :: xg(*), uold(*), unew(*)
       real(c_double), value :: dx, dt
       integer(c_int) :: rc
     end function ramses_amd_mhd_godunov_fine_f90
     function ramses_amd_mhd_godunov_fine_amr_f90(p, ilevel, levelmin, ngrid, igrid, son, nbor, father, ngridmax, ncoarse, uold, unew, &
          & f, use_f, dx, dt, nvector, interpol_var, interpol_type, interpol_mag_type) &
          & bind(C, name='ramses_amd_mhd_godunov_fine_amr_f90') result(rc)
       import :: ramses_amd_mhd_params, c_int, c_int64_t, c_double
       type(ramses_amd_mhd_params), intent(in) :: p
       integer(c_int), value :: ilevel, levelmin, ngrid, use_f, nvector, interpol_var, interpol_type, interpol_mag_type
       integer(c_int) :: igrid(*), son(*), nbor(*), father(*)
       integer(c_int64_t), value :: ngridmax, ncoarse
       real(c_double) :: uold(*), unew(*), f(*)
       real(c_double), value :: dx, dt
       integer(c_int) :: rc
     end function ramses_amd_mhd_godunov_fine_amr_f90
     function ramses_amd_mhd_note_reference_sweep(ilevel) bind(C, name='ramses_amd_mhd_note_reference_sweep') result(rc)
       import :: c_int
       integer(c_int), value :: ilevel
       integer(c_int) :: rc
     end function ramses_amd_mhd_note_reference_sweep
     function ramses_amd_mhd_resident_active() bind(C, name='ramses_amd_mhd_resident_active') result(rc)
       import :: c_int
       integer(c_int) :: rc
     end function ramses_amd_mhd_resident_active
     function ramses_amd_mhd_resident_courant_f90(p, ilevel, ngrid, igrid, xg, ngridmax, ncoarse, nx_loc, uold, dx, dt_in, &
          & courant_factor, out5) bind(C, name='ramses_amd_mhd_resident_courant_f90') result(rc)
       import :: ramses_amd_mhd_params, c_int, c_int64_t, c_double
       type(ramses_amd_mhd_params), intent(in) :: p
       integer(c_int), value :: ilevel, ngrid, nx_loc
       integer(c_int) :: igrid(*)
       integer(c_int64_t), value :: ngridmax, ncoarse
       real(c_double) :: xg(*), uold(*), out5(5)
       real(c_double), value :: dx, dt_in, courant_factor
       integer(c_int) :: rc
     end function ramses_amd_mhd_resident_courant_f90
     function ramses_amd_mhd_resident_godunov_f90(p, ilevel, ngrid, igrid, xg, ngridmax, ncoarse, nx_loc, uold, dx, dt) &
          & bind(C, name='ramses_amd_mhd_resident_godunov_f90') result(rc)
       import :: ramses_amd_mhd_params, c_int, c_int64_t, c_double
       type(ramses_amd_mhd_params), intent(in) :: p
       integer(c_int), value :: ilevel, ngrid, nx_loc
       integer(c_int) :: igrid(*)
       integer(c_int64_t), value :: ngridmax, ncoarse
       real(c_double) :: xg(*), uold(*)
       real(c_double), value :: dx, dt
       integer(c_int) :: rc
     end function ramses_amd_mhd_resident_godunov_f90
     function ramses_amd_mhd_resident_set_uold_f90(ilevel) bind(C, name='ramses_amd_mhd_resident_set_uold_f90') result(rc)
       import :: c_int
       integer(c_int), value :: ilevel
       integer(c_int) :: rc
     end function ramses_amd_mhd_resident_set_uold_f90
     function ramses_amd_mhd_resident_sync_host_f90(uold) bind(C, name='ramses_amd_mhd_resident_sync_host_f90') result(rc)
       import :: c_int, c_double
       real(c_double) :: uold(*)
       integer(c_int) :: rc
     end function ramses_amd_mhd_resident_sync_host_f90
     function ramses_amd_last_error() bind(C, name='ramses_amd_last_error') result(msg)
       import :: c_ptr
       type(c_ptr) :: msg
     end function ramses_amd_last_error
  end interface
  logical, save :: ramses_amd_mhd_first = .true.
contains
  logical function ramses_amd_mhd_enabled()
    character(len=16) :: val
    integer :: stat
    logical, save :: first = .true., on = .true.
    if (first) then
       call get_environment_variable('RAMSES_AMD', val, status=stat)
       if (stat == 0) then
          if (trim(val) == '0') on = .false.
       end if
       first = .false.
    end if
    ramses_amd_mhd_enabled = on
  end function ramses_amd_mhd_enabled

  ! What the device sweep covers that does not depend on the level (the caller adds: the level is fully refined and has
  ! no finer level)
  logical function ramses_amd_mhd_device_config()
    use amr_commons
    use hydro_commons
    integer :: nx_loc
    nx_loc = icoarse_max - icoarse_min + 1
    ramses_amd_mhd_device_config = ramses_amd_mhd_enabled() .and. hydro .and. ndim == 3 .and. nvar == 8 .and. ncpu == 1 &
         & .and. nboundary == 0 .and. nx_loc == 1 .and. jcoarse_max == jcoarse_min .and. kcoarse_max == kcoarse_min &
         & .and. .not. poisson .and. .not. pressure_fix .and. ischeme == 0 .and. .not. allow_switch_solver &
         & .and. .not. allow_switch_solver2D &
         & .and. iriemann >= 0 .and. iriemann <= 5 .and. iriemann2d >= 0 .and. iriemann2d <= 5 &
         & .and. (slope_type == 0 .or. slope_type == 1 .or. slope_type == 2 .or. slope_type == 3 .or. slope_type == 7 &
         &        .or. slope_type == 8) &
         & .and. (slope_mag_type == 0 .or. slope_mag_type == 1 .or. slope_mag_type == 2 .or. slope_mag_type == 7 &
         &        .or. slope_mag_type == 8)
  end function ramses_amd_mhd_device_config

  ! godfine1 of ANY level of an AMR tree on the device (csrc/mhd_amr.hip, staged): what the brick sweep asks for, except that
  ! self-gravity is allowed (ctoprim's half kick from f; the source terms stay the reference's host routines) -- any number of
  ! ranks, periodic or walled box, no pressure_fix
  logical function ramses_amd_mhd_amr_config()
    use amr_commons
    use hydro_commons
    integer :: nx_loc
    character(len=16) :: val
    integer :: stat
    logical, save :: first = .true., on = .true.
    if (first) then
       first = .false.
       call get_environment_variable('RAMSES_AMD_MHD_AMR', val, status=stat)
       if (stat == 0) then
          if (trim(val) == '0') on = .false.
       end if
    end if
    nx_loc = icoarse_max - icoarse_min + 1
    ! (several ranks: every rank sweeps its own active octs; the virtual octs of its neighbours are ordinary octs of its tree, kept
    !  current by the reference's make_virtual_fine_dp, and what the sweep owes to coarse cells of other ranks travels home through
    !  the reference's make_virtual_reverse_dp on unew, as with the reference's own godfine1)
    ! (physical boundaries: the boundary octs are ordinary octs of the tree, filled by the reference's make_boundary_hydro
    !  (mhd/hydro_boundary.f90) before godunov_fine reads them -- as for the reference's own godfine1)
    ramses_amd_mhd_amr_config = on .and. ramses_amd_mhd_enabled() .and. hydro .and. ndim == 3 .and. nvar == 8 &
         & .and. .not. pressure_fix .and. ischeme == 0 .and. .not. allow_switch_solver &
         & .and. .not. allow_switch_solver2D .and. .not. MC_tracer &
         & .and. iriemann >= 0 .and. iriemann <= 5 .and. iriemann2d >= 0 .and. iriemann2d <= 5 &
         & .and. (slope_type == 0 .or. slope_type == 1 .or. slope_type == 2 .or. slope_type == 3 .or. slope_type == 7 &
         &        .or. slope_type == 8) &
         & .and. (slope_mag_type == 0 .or. slope_mag_type == 1 .or. slope_mag_type == 2 .or. slope_mag_type == 7 &
         &        .or. slope_mag_type == 8) &
         & .and. interpol_var >= 0 .and. interpol_var <= 1 .and. interpol_type >= 0 .and. interpol_type <= 3 &
         & .and. interpol_mag_type >= 0 .and. interpol_mag_type <= 3
  end function ramses_amd_mhd_amr_config

  ! The level stays on the device between courant_fine, godunov_fine and set_uold: one level (levelmin = nlevelmax), and
  ! nothing else in the time loop that reads or writes uold on the host (magnetic diffusion, cooling, particles, ...).
  ! RAMSES_AMD_MHD_RESIDENT=0 keeps the staged sweep.
  logical function ramses_amd_mhd_resident()
    use amr_commons
    use hydro_commons
#if USE_TURB==1
    use turb_commons, only: turb
#endif
    character(len=16) :: val
    integer :: stat
    logical, save :: first = .true., on = .false.
    if (first) then
       first = .false.
       on = ramses_amd_mhd_device_config() .and. levelmin == nlevelmax .and. levelmin >= 2 .and. levelmin <= 10
       if (eta_mag > 0.0d0) on = .false.
       if (pic .or. rt .or. cooling .or. star .or. sink .or. tracer .or. clumpfind .or. lightcone .or. movie) on = .false.
       if (static .or. cosmo) on = .false.
       ! every host routine of amr_step that reads or writes uold during the time loop sends the run to the staged path
       ! (the list of the hydro gate, ramses_amd_amr_config): cooling_fine (amr/amr_step.f90:472) runs for T2_star > 0,
       ! barotropic_eos, neq_chem too; the turbulent forcing of synchro_hydro_fine (:433-437) and courant_fine's gg
       if (T2_star > 0.0d0 .or. barotropic_eos .or. neq_chem .or. isothermal) on = .false.
       if (MC_tracer .or. momentum_feedback > 0 .or. strict_equilibrium > 0 .or. static_gas .or. metal .or. aton) on = .false.
#if USE_TURB==1
       if (turb) on = .false.
#endif
       call get_environment_variable('RAMSES_AMD_MHD_RESIDENT', val, status=stat)
       if (stat == 0) then
          if (trim(val) == '0') on = .false.
       end if
       if (on) write(*,*) 'ramses_amd: the MHD level stays resident on the GPU (courant_fine, godunov_fine, set_uold)'
    end if
    ramses_amd_mhd_resident = on
  end function ramses_amd_mhd_resident

  subroutine ramses_amd_mhd_fill_params(p)
    use hydro_commons
    type(ramses_amd_mhd_params), intent(out) :: p
    p%gamma = gamma; p%smallr = smallr; p%smallc = smallc; p%slope_theta = slope_theta
    p%slope_type = slope_type; p%slope_mag_type = slope_mag_type; p%riemann = iriemann; p%riemann2d = iriemann2d
  end subroutine ramses_amd_mhd_fill_params

  subroutine ramses_amd_mhd_fatal(where)
    character(len=*), intent(in) :: where
    type(c_ptr) :: msg
    character(kind=c_char), pointer :: s(:)
    integer :: n
    msg = ramses_amd_last_error()
    write(*,*) 'ramses_amd (MHD): ', where, ' failed'
    if (c_associated(msg)) then
       call c_f_pointer(msg, s, [512])
       n = 1
       do while (n < 512 .and. s(n) /= c_null_char)
          n = n + 1
       end do
       write(*,*) s(1:n-1)
    end if
    call clean_stop
  end subroutine ramses_amd_mhd_fatal
end module ramses_amd_mhd_iface
