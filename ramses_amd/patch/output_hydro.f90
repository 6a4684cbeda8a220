!==============================================================================
! output_hydro.f90 of the ramses_amd patch directory.
!
! Shadows hydro/output_hydro.f90.  backup_hydro is the one host routine that
! reads uold while a level is device-resident: the new backup_hydro refreshes
! the host array from the GPU (a no-op when it is current) and then runs the
! untouched reference routine, so snapshots keep the reference's format and
! content.
!==============================================================================
#define backup_hydro backup_hydro_reference
#include "hydro/output_hydro.f90"
#undef backup_hydro

subroutine backup_hydro(filename, filename_desc)
  use amr_commons
  use hydro_commons
  use ramses_amd_iface
  implicit none
  character(len=80), intent(in) :: filename, filename_desc
  integer::rc
  if(ramses_amd_enabled())then
     rc=ramses_amd_resident_sync_host_f90(uold)
     if(rc/=0)call ramses_amd_fatal('backup_hydro (sync of the resident level)')
     if(ramses_amd_amr_resident())then
        if(ramses_amd_amrres_active()/=0)then
           call ramses_amd_amr_ensure()          ! (levels rebuilt on the host since the last device routine go first)
           rc=ramses_amd_amrres_sync_all(uold)
           if(rc/=0)call ramses_amd_fatal('backup_hydro (sync of the resident AMR state)')
        end if
     end if
     if(ramses_amd_mpi_on)then
        rc=ramses_amd_mpires_sync_host(uold)
        if(rc/=0)call ramses_amd_fatal('backup_hydro (sync of the resident bricks)')
     end if
  end if
  call backup_hydro_reference(filename, filename_desc)
end subroutine backup_hydro
