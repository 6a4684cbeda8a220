!==============================================================================
! interpol_hydro.f90 of the ramses_amd patch directory.
!
! Shadows hydro/interpol_hydro.f90 (upload_fine -> upload_fine_reference by #define + #include;
! upl, interpol_hydro and the limiters stay the reference's).  While the hydro state of an AMR
! run is device-resident (ramses_amd_iface: ramses_amd_amr_resident) the restriction of a level's
! split cells (upl, hydro/interpol_hydro.f90:73-263) runs on the GPU on the reference's own cell
! vectors; otherwise the reference routine.
!==============================================================================
#define upload_fine upload_fine_reference
#include "hydro/interpol_hydro.f90"
#undef upload_fine

subroutine upload_fine(ilevel)
  use amr_commons
  use hydro_commons
  use ramses_amd_iface
  implicit none
  integer::ilevel
  integer::rc
  type(ramses_amd_hydro_params)::p
  if(ilevel==nlevelmax)return
  if(numbtot(1,ilevel)==0)return
  if(.not.ramses_amd_amr_resident())then
     call upload_fine_reference(ilevel)
     return
  end if
  if(verbose)write(*,111)ilevel
  call ramses_amd_amr_ensure()
  call ramses_amd_fill_hydro_params(p)
  rc=ramses_amd_amrres_upload_fine(p,active(ilevel)%ngrid,ramses_amd_octs(ilevel),interpol_var)
  if(rc/=0)call ramses_amd_fatal('upload_fine')
111 format('   Entering upload_fine (MI355X) for level',i2)
end subroutine upload_fine
