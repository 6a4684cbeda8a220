!==============================================================================
! multigrid_fine_coarse.f90 of the ramses_amd patch directory.
!
! Shadows poisson/multigrid_fine_coarse.f90: the coarse-level twins of the
! compute routines (see multigrid_fine_fine.f90 of this directory).
!==============================================================================
#define gauss_seidel_mg_coarse gauss_seidel_mg_coarse_reference
#define cmp_residual_mg_coarse cmp_residual_mg_coarse_reference
#define restrict_residual_coarse_reverse restrict_residual_coarse_reverse_reference
#define interpolate_and_correct_coarse interpolate_and_correct_coarse_reference
#include "poisson/multigrid_fine_coarse.f90"
#undef gauss_seidel_mg_coarse
#undef cmp_residual_mg_coarse
#undef restrict_residual_coarse_reverse
#undef interpolate_and_correct_coarse

subroutine gauss_seidel_mg_coarse(ilevel,safe,redstep)
  use ramses_amd_iface
  implicit none
  integer, intent(in) :: ilevel
  logical, intent(in) :: safe
  logical, intent(in) :: redstep
  integer :: rc, ired, isafe
  if(.not.ramses_amd_mg_on_device(32))then
     call gauss_seidel_mg_coarse_reference(ilevel,safe,redstep)
     return
  end if
  call ramses_amd_mg_ensure()
  ired=0; if(redstep)ired=1
  isafe=0; if(safe)isafe=1
  rc=ramses_amd_mgamr_gauss_seidel(ilevel,ired,isafe)
  if(rc/=0)call ramses_amd_fatal('gauss_seidel_mg_coarse')
end subroutine gauss_seidel_mg_coarse

subroutine cmp_residual_mg_coarse(ilevel)
  use ramses_amd_iface
  implicit none
  integer, intent(in) :: ilevel
  integer :: rc
  if(.not.ramses_amd_mg_on_device(64))then
     call cmp_residual_mg_coarse_reference(ilevel)
     return
  end if
  call ramses_amd_mg_ensure()
  rc=ramses_amd_mgamr_residual(ilevel)
  if(rc/=0)call ramses_amd_fatal('cmp_residual_mg_coarse')
end subroutine cmp_residual_mg_coarse

subroutine restrict_residual_coarse_reverse(ifinelevel)
  use ramses_amd_iface
  implicit none
  integer, intent(in) :: ifinelevel
  integer :: rc
  if(.not.ramses_amd_mg_on_device(128))then
     call restrict_residual_coarse_reverse_reference(ifinelevel)
     return
  end if
  call ramses_amd_mg_ensure()
  rc=ramses_amd_mgamr_restrict(ifinelevel)
  if(rc/=0)call ramses_amd_fatal('restrict_residual_coarse_reverse')
end subroutine restrict_residual_coarse_reverse

subroutine interpolate_and_correct_coarse(ifinelevel)
  use ramses_amd_iface
  implicit none
  integer, intent(in) :: ifinelevel
  integer :: rc
  if(.not.ramses_amd_mg_on_device(256))then
     call interpolate_and_correct_coarse_reference(ifinelevel)
     return
  end if
  call ramses_amd_mg_ensure()
  rc=ramses_amd_mgamr_interpolate(ifinelevel)
  if(rc/=0)call ramses_amd_fatal('interpolate_and_correct_coarse')
end subroutine interpolate_and_correct_coarse
