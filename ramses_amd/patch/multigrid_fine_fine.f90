!==============================================================================
! multigrid_fine_fine.f90 of the ramses_amd patch directory.
!
! Shadows poisson/multigrid_fine_fine.f90.  The compute routines of the AMR
! multigrid are pulled in under *_reference names; the routines of the same
! name below run on the MI355X while the multigrid_fine shim has a device
! solve open (ramses_amd_mg_active), and are the reference's otherwise.  The
! mask restriction and the scan flags (per-solve setup) stay the reference's.
!==============================================================================
#define gauss_seidel_mg_fine gauss_seidel_mg_fine_reference
#define cmp_residual_mg_fine cmp_residual_mg_fine_reference
#define cmp_residual_norm2_fine cmp_residual_norm2_fine_reference
#define restrict_residual_fine_reverse restrict_residual_fine_reverse_reference
#define interpolate_and_correct_fine interpolate_and_correct_fine_reference
#include "poisson/multigrid_fine_fine.f90"
#undef gauss_seidel_mg_fine
#undef cmp_residual_mg_fine
#undef cmp_residual_norm2_fine
#undef restrict_residual_fine_reverse
#undef interpolate_and_correct_fine

subroutine gauss_seidel_mg_fine(ilevel,redstep)
  use amr_commons
  use poisson_commons
  use ramses_amd_iface
  implicit none
  integer, intent(in) :: ilevel
  logical, intent(in) :: redstep
  integer :: rc, ired, isafe
  if(.not.ramses_amd_mg_on_device(1))then
     call gauss_seidel_mg_fine_reference(ilevel,redstep)
     return
  end if
  call ramses_amd_mg_ensure()
  ired=0; if(redstep)ired=1
  isafe=0; if(safe_mode(ilevel))isafe=1
  rc=ramses_amd_mgamr_gauss_seidel(ilevel,ired,isafe)
  if(rc/=0)call ramses_amd_fatal('gauss_seidel_mg_fine')
end subroutine gauss_seidel_mg_fine

subroutine cmp_residual_mg_fine(ilevel)
  use ramses_amd_iface
  implicit none
  integer, intent(in) :: ilevel
  integer :: rc
  if(.not.ramses_amd_mg_on_device(2))then
     call cmp_residual_mg_fine_reference(ilevel)
     return
  end if
  call ramses_amd_mg_ensure()
  rc=ramses_amd_mgamr_residual(ilevel)
  if(rc/=0)call ramses_amd_fatal('cmp_residual_mg_fine')
end subroutine cmp_residual_mg_fine

subroutine cmp_residual_norm2_fine(ilevel, norm2)
  use ramses_amd_iface
  implicit none
  integer,  intent(in)  :: ilevel
  real(kind=8), intent(out) :: norm2
  integer :: rc
  if(.not.ramses_amd_mg_on_device(4))then
     call cmp_residual_norm2_fine_reference(ilevel, norm2)
     return
  end if
  call ramses_amd_mg_ensure()
  rc=ramses_amd_mgamr_norm2(ilevel,norm2)
  if(rc/=0)call ramses_amd_fatal('cmp_residual_norm2_fine')
end subroutine cmp_residual_norm2_fine

subroutine restrict_residual_fine_reverse(ifinelevel)
  use ramses_amd_iface
  implicit none
  integer, intent(in) :: ifinelevel
  integer :: rc
  if(.not.ramses_amd_mg_on_device(8))then
     call restrict_residual_fine_reverse_reference(ifinelevel)
     return
  end if
  call ramses_amd_mg_ensure()
  rc=ramses_amd_mgamr_restrict(ifinelevel)
  if(rc/=0)call ramses_amd_fatal('restrict_residual_fine_reverse')
end subroutine restrict_residual_fine_reverse

subroutine interpolate_and_correct_fine(ifinelevel)
  use ramses_amd_iface
  implicit none
  integer, intent(in) :: ifinelevel
  integer :: rc
  if(.not.ramses_amd_mg_on_device(16))then
     call interpolate_and_correct_fine_reference(ifinelevel)
     return
  end if
  call ramses_amd_mg_ensure()
  rc=ramses_amd_mgamr_interpolate(ifinelevel)
  if(rc/=0)call ramses_amd_fatal('interpolate_and_correct_fine')
end subroutine interpolate_and_correct_fine
