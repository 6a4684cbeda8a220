!==============================================================================
! output_poisson.f90 of the ramses_amd patch directory.
!
! Shadows poisson/output_poisson.f90.  backup_poisson is the one host routine that reads
! phi and f (and rho with -DOUTPUT_PARTICLE_DENSITY) while the Poisson fields of the
! device-resident level live on the GPU: the new backup_poisson refreshes the host arrays
! (a no-op when they are current) and then runs the untouched reference routine, so the
! grav_*.out files keep the reference's format and content.
!==============================================================================
#define backup_poisson backup_poisson_reference
#include "poisson/output_poisson.f90"
#undef backup_poisson

subroutine backup_poisson(filename)
  use amr_commons
  use poisson_commons
  use ramses_amd_iface
  implicit none
  character(LEN=80)::filename
  integer::rc
  if(ramses_amd_enabled())then
     rc=ramses_amd_resident_sync_poisson_f90(phi,f,rho)
     if(rc/=0)call ramses_amd_fatal('backup_poisson (sync of the resident level)')
     ! the acceleration of the levels force_fine left on the device only (several ranks: patch/force_fine.f90)
     if(ramses_amd_amr_resident())then
        call ramses_amd_amr_sync_f()
        ! rho and phi of a one-level run under the distributed dense multigrid (ramses_amd_iface: ramses_amd_pois_mpi_dev)
        call ramses_amd_pois_mpi_sync_host(.false.)
     end if
  end if
  call backup_poisson_reference(filename)
end subroutine backup_poisson
