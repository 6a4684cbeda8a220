!==============================================================================
! refine_utils.f90 of the ramses_amd patch directory.
!
! Shadows amr/refine_utils.f90 (refine_fine -> refine_fine_reference by #define + #include;
! refine_coarse, make_grid_fine, kill_grid, ... stay the reference's).  The mesh is the
! reference's host code; while the hydro state of an AMR run is device-resident the new
! refine_fine(ilevel) first brings back the levels the reference is about to read (interpol_hydro
! of new octs: level ilevel and, through getnborfather's fallback, ilevel-1) and notes which
! levels it rebuilds, so that they -- and the tree -- are sent again before the next device routine.
!==============================================================================
#define refine_fine refine_fine_reference
#include "amr/refine_utils.f90"
#undef refine_fine

subroutine refine_fine(ilevel)
  use amr_commons
  use ramses_amd_iface
  implicit none
  integer::ilevel
  call ramses_amd_amr_refine_hook(ilevel)
  call refine_fine_reference(ilevel)
end subroutine refine_fine
