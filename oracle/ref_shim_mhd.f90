! ref_shim_mhd.f90 -- TEST INFRASTRUCTURE ONLY.
!
! bind(C) entry points around the UNMODIFIED MHD routines of the reference (mhd/umuscl.f90, mhd/godunov_utils.f90,
! compiled from /root/reference by oracle/build_ref.sh into oracle/_ref/libref_kernels3d_mhd.so with -DSOLVERmhd
! -DNDIM=3 -DNVAR=8): mag_unsplit on a batch of 6^3 stencils.  Nothing here restates reference arithmetic.
subroutine ref_mhd_get_dims(ndim_out, nvar_out, nvector_out) bind(C, name='ref_mhd_get_dims')
  use iso_c_binding
  use amr_parameters
  use hydro_parameters
  implicit none
  integer(c_int), intent(out) :: ndim_out, nvar_out, nvector_out
  ndim_out = ndim
  nvar_out = nvar
  nvector_out = nvector
end subroutine ref_mhd_get_dims

subroutine ref_mhd_set_params(gamma_in, smallr_in, smallc_in, slope_type_in, slope_mag_type_in, slope_theta_in, &
     & iriemann_in, iriemann2d_in) bind(C, name='ref_mhd_set_params')
  use iso_c_binding
  use amr_parameters
  use hydro_parameters
  implicit none
  real(c_double), value :: gamma_in, smallr_in, smallc_in, slope_theta_in
  integer(c_int), value :: slope_type_in, slope_mag_type_in, iriemann_in, iriemann2d_in
  gamma = gamma_in
  smallr = smallr_in
  smallc = smallc_in
  slope_type = slope_type_in
  slope_mag_type = slope_mag_type_in
  slope_theta = slope_theta_in
  iriemann = iriemann_in
  iriemann2d = iriemann2d_in
  ischeme = 0
  allow_switch_solver = .false.
  allow_switch_solver2D = .false.
end subroutine ref_mhd_set_params

! uin(nvector,-1:4,-1:4,-1:4,nvar+3), gravin(nvector,-1:4,-1:4,-1:4,3), flux(nvector,1:3,1:3,1:3,nvar,3),
! emfx/y/z(nvector,1:3,1:3,1:3), tmp(nvector,1:3,1:3,1:3,2,3)
subroutine ref_mag_unsplit(uin, gravin, flux, emfx, emfy, emfz, tmp, dx, dy, dz, dt, ngrid) bind(C, name='ref_mag_unsplit')
  use iso_c_binding
  use amr_parameters
  use hydro_parameters
  implicit none
  real(c_double) :: uin(*), gravin(*), flux(*), emfx(*), emfy(*), emfz(*), tmp(*)
  real(c_double), value :: dx, dy, dz, dt
  integer(c_int), value :: ngrid
  real(dp) :: dxl, dyl, dzl, dtl
  integer :: ng
  dxl = dx; dyl = dy; dzl = dz; dtl = dt; ng = ngrid
  call mag_unsplit(uin, gravin, flux, emfx, emfy, emfz, tmp, dxl, dyl, dzl, dtl, ng)
end subroutine ref_mag_unsplit

! cmpdt of the MHD solver (mhd/godunov_utils.f90:5-115) on ncell <= nvector cells without gravity: uu(nvector,nvar+3) is
! overwritten by the routine, dt returned
subroutine ref_mhd_cmpdt(uu, dx, courant_factor_in, ncell, dt) bind(C, name='ref_mhd_cmpdt')
  use iso_c_binding
  use amr_parameters
  use hydro_parameters
  implicit none
  real(c_double) :: uu(*)
  real(c_double), value :: dx, courant_factor_in
  integer(c_int), value :: ncell
  real(c_double), intent(out) :: dt
  real(dp), dimension(1:nvector, 1:ndim) :: gg
  real(dp) :: dxl, dtl
  integer :: nc
  gg = 0.0d0
  courant_factor = courant_factor_in
  dxl = dx; nc = ncell
  call cmpdt(uu, gg, dxl, dtl, nc)
  dt = dtl
end subroutine ref_mhd_cmpdt

! cmpflxm / cmp_mag_flx call clean_stop on an unknown solver code (amr/update_time.f90 in the full program)
subroutine clean_stop
  implicit none
  write(*,*) 'ref_shim_mhd: clean_stop called by the reference kernels'
  stop 1
end subroutine clean_stop
