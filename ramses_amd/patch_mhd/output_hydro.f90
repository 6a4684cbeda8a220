!==============================================================================
! output_hydro.f90 of the ramses_amd MHD patch directory.
!
! Shadows mhd/output_hydro.f90.  backup_hydro is the one host routine that reads uold while the level is device-resident:
! the new backup_hydro refreshes the host array from the GPU (a no-op when it is current) and then runs the untouched
! reference routine, so snapshots keep the reference's format and content.
!==============================================================================
#define backup_hydro backup_hydro_reference
#include "mhd/output_hydro.f90"
#undef backup_hydro

subroutine backup_hydro(filename, filename_desc)
  use amr_commons
  use hydro_commons
  use ramses_amd_mhd_iface
  implicit none
  character(len=80), intent(in) :: filename, filename_desc
  integer::rc
  if(ramses_amd_mhd_enabled())then
     rc=ramses_amd_mhd_resident_sync_host_f90(uold)
     if(rc/=0)call ramses_amd_mhd_fatal('backup_hydro (sync of the resident level)')
  end if
  call backup_hydro_reference(filename, filename_desc)
end subroutine backup_hydro
