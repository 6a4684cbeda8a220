!==============================================================================
! courant_fine.f90 of the ramses_amd MHD patch directory.
!
! Shadows mhd/courant_fine.f90 (courant_fine -> courant_fine_reference by the preprocessor; velocity_fine stays the
! reference's).  While the level of a one-level run is device-resident (ramses_amd_mhd_iface: ramses_amd_mhd_resident)
! the CFL time step (cmpdt, mhd/godunov_utils.f90:5-115, per cell; the minimum is exact) and the four sums of the
! conservation diagnostics are computed on the device; the routine keeps the reference's name, argument and meaning
! (mhd/courant_fine.f90:1-160).  The first call of the time loop is what sends the level to the device.
!==============================================================================
#define courant_fine courant_fine_reference
#include "mhd/courant_fine.f90"
#undef courant_fine

subroutine courant_fine(ilevel)
  use amr_commons
  use hydro_commons
  use ramses_amd_mhd_iface
  implicit none
  integer::ilevel
  integer::rc,nx_loc,i
  real(dp)::dx,scale
  real(kind=8),dimension(5)::out5
  type(ramses_amd_mhd_params)::p
  integer,allocatable,dimension(:)::octs
  if(numbtot(1,ilevel)==0)return
  if(.not.ramses_amd_mhd_resident().or.ilevel/=levelmin.or.int(active(ilevel)%ngrid,8)*8_8/=(2_8**ilevel)**3)then
     call courant_fine_reference(ilevel)
     return
  end if
  if(verbose)write(*,111)ilevel
  call ramses_amd_mhd_fill_params(p)
  nx_loc=icoarse_max-icoarse_min+1
  scale=boxlen/dble(nx_loc)
  dx=0.5D0**ilevel*scale
  allocate(octs(active(ilevel)%ngrid))
  do i=1,active(ilevel)%ngrid
     octs(i)=active(ilevel)%igrid(i)
  end do
  rc=ramses_amd_mhd_resident_courant_f90(p,ilevel,active(ilevel)%ngrid,octs,xg,int(ngridmax,8),int(ncoarse,8),nx_loc, &
       & uold,dx,dtnew(ilevel),courant_factor,out5)
  deallocate(octs)
  if(rc/=0)call ramses_amd_mhd_fatal('courant_fine')
  ! :148-158
  mass_tot=mass_tot+out5(2)
  ekin_tot=ekin_tot+out5(3)
  eint_tot=eint_tot+out5(4)
  emag_tot=emag_tot+out5(5)
  dtnew(ilevel)=MIN(dtnew(ilevel),out5(1))
111 format('   Entering courant_fine (MHD, MI355X) for level ',I2)
end subroutine courant_fine
