! ref_shim.f90 -- TEST INFRASTRUCTURE ONLY.
!
! Thin bind(C) entry points around the UNMODIFIED reference routines
! (compiled from /root/reference by oracle/build_ref.sh into oracle/_ref/)
! so that tests can call the reference's own unsplit / riemann_* / cmpdt and
! compare them bit-for-bit with oracle/hydro_oracle.c and with the HIP path.
! Nothing here restates reference arithmetic: it only sets the solver knobs
! (module variables of hydro_parameters) and forwards the arrays.
!
! The reference fixes NDIM / NVAR / NVECTOR at compile time, so there is one
! shared object per NDIM: oracle/_ref/libref_kernels{1,2,3}d.so.

subroutine ref_get_dims(ndim_out, nvar_out, nvector_out) bind(C, name='ref_get_dims')
  use iso_c_binding
  use amr_parameters
  use hydro_parameters
  implicit none
  integer(c_int), intent(out) :: ndim_out, nvar_out, nvector_out
  ndim_out = ndim
  nvar_out = nvar
  nvector_out = nvector
end subroutine ref_get_dims

subroutine ref_set_hydro_params(gamma_in, smallr_in, smallc_in, slope_type_in, &
     & slope_theta_in, riemann_in, scheme_in, niter_in, difmag_in, courant_in) &
     & bind(C, name='ref_set_hydro_params')
  use iso_c_binding
  use amr_parameters
  use hydro_parameters
  implicit none
  real(c_double), value :: gamma_in, smallr_in, smallc_in, slope_theta_in, difmag_in, courant_in
  integer(c_int), value :: slope_type_in, riemann_in, scheme_in, niter_in
  gamma = gamma_in
  smallr = smallr_in
  smallc = smallc_in
  slope_type = slope_type_in
  slope_theta = slope_theta_in
  niter_riemann = niter_in
  difmag = difmag_in
  courant_factor = courant_in
  select case (riemann_in)
  case (0); riemann = 'llf'
  case (1); riemann = 'hllc'
  case (2); riemann = 'hll'
  case (3); riemann = 'acoustic'
  case (4); riemann = 'exact'
  end select
  if (scheme_in == 0) then
     scheme = 'muscl'
  else
     scheme = 'plmde'
  end if
end subroutine ref_set_hydro_params

subroutine ref_unsplit(uin, gravin, flux, tmp, dx, dy, dz, dt, ngrid) bind(C, name='ref_unsplit')
  use iso_c_binding
  use amr_parameters
  use hydro_parameters
  implicit none
  real(c_double) :: uin(*), gravin(*), flux(*), tmp(*)
  real(c_double), value :: dx, dy, dz, dt
  integer(c_int), value :: ngrid
  real(dp), dimension(1:nvector, iu1:iu2, ju1:ju2, ku1:ku2) :: pin
  real(dp) :: dxl, dyl, dzl, dtl
  integer :: ng
  pin = 0.0d0
  dxl = dx; dyl = dy; dzl = dz; dtl = dt; ng = ngrid
  call unsplit(uin, gravin, pin, flux, tmp, dxl, dyl, dzl, dtl, ng)
end subroutine ref_unsplit

subroutine ref_riemann(qleft, qright, fgdnv, ngrid) bind(C, name='ref_riemann')
  use iso_c_binding
  use amr_parameters
  use hydro_parameters
  implicit none
  real(c_double) :: qleft(*), qright(*), fgdnv(*)
  integer(c_int), value :: ngrid
  integer :: ng
  ng = ngrid
  if (riemann .eq. 'acoustic') then
     call riemann_acoustic(qleft, qright, fgdnv, ng)
  else if (riemann .eq. 'exact') then
     call riemann_approx(qleft, qright, fgdnv, ng)
  else if (riemann .eq. 'llf') then
     call riemann_llf(qleft, qright, fgdnv, ng)
  else if (riemann .eq. 'hllc') then
     call riemann_hllc(qleft, qright, fgdnv, ng)
  else if (riemann .eq. 'hll') then
     call riemann_hll(qleft, qright, fgdnv, ng)
  end if
end subroutine ref_riemann

! cmpdt overwrites uu; the caller passes a scratch copy (nvector,nvar).
subroutine ref_cmpdt(uu, gg, dx, dt, ncell) bind(C, name='ref_cmpdt')
  use iso_c_binding
  use amr_parameters
  use hydro_parameters
  implicit none
  real(c_double) :: uu(*), gg(*)
  real(c_double), value :: dx
  real(c_double), intent(out) :: dt
  integer(c_int), value :: ncell
  real(dp) :: dxl, dtl
  integer :: nc
  dxl = dx; nc = ncell
  call cmpdt(uu, gg, dxl, dtl, nc)
  dt = dtl
end subroutine ref_cmpdt

! ---------------------------------------------------------------------------
! coarse<->fine operators: interpol_hydro is a pure routine of its arguments
! and the interpolation knobs; upl reads the tree (son) and uold, which this
! shim allocates as a one-oct-per-parent toy tree: parent cell i = coarse-level
! cell slot i, its son oct = slot i, children at ncoarse+(ind-1)*ngridmax+i.
! ---------------------------------------------------------------------------
subroutine ref_interpol_hydro(u1, u2, nn, ivar_in, itype_in, smallr_in) bind(C, name='ref_interpol_hydro')
  use iso_c_binding
  use amr_parameters
  use hydro_parameters
  implicit none
  real(c_double) :: u1(*), u2(*)
  integer(c_int), value :: nn, ivar_in, itype_in
  real(c_double), value :: smallr_in
  integer :: n
  interpol_var = ivar_in
  interpol_type = itype_in
  smallr = smallr_in
  n = nn
  call interpol_hydro(u1, u2, n)
end subroutine ref_interpol_hydro

subroutine ref_upl(child, parent, nn, ivar_in, smallr_in) bind(C, name='ref_upl')
  use iso_c_binding
  use amr_commons
  use hydro_commons
  implicit none
  real(c_double) :: child(nvector, 8, nvar), parent(nvector, nvar)
  integer(c_int), value :: nn, ivar_in
  real(c_double), value :: smallr_in
  integer :: i, ind, iv, n, ncell
  integer, dimension(1:nvector) :: ind_cell
  interpol_var = ivar_in
  smallr = smallr_in
  n = nn
  ! toy tree: ngridmax = 2*nvector oct slots; parents live in octant 1 of octs
  ! nvector+1..2*nvector, their sons are octs 1..nvector
  ngridmax = 2 * nvector
  ncoarse = 1
  ncell = ncoarse + 8 * ngridmax
  if (.not. allocated(uold)) allocate(uold(1:ncell, 1:nvar))
  if (.not. allocated(son)) allocate(son(1:ncell))
  uold = 0.0d0
  son = 0
  do i = 1, n
     ind_cell(i) = ncoarse + (nvector + i)
     son(ind_cell(i)) = i
     do iv = 1, nvar
        uold(ind_cell(i), iv) = parent(i, iv)
        do ind = 1, 8
           uold(ncoarse + (ind - 1) * ngridmax + i, iv) = child(i, ind, iv)
        end do
     end do
  end do
  call upl(ind_cell, n)
  do i = 1, n
     do iv = 1, nvar
        parent(i, iv) = uold(ind_cell(i), iv)
     end do
  end do
end subroutine ref_upl

! interpol_hydro references clean_stop (amr/end.f90), which is not part of the
! kernel-level library: a stub with the same effect (the run ends).
subroutine clean_stop
  implicit none
  write(*,*) 'ref_shim: clean_stop called'
  stop 2
end subroutine clean_stop
