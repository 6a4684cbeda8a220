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
