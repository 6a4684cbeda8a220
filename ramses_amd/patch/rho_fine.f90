!==============================================================================
! rho_fine.f90 of the ramses_amd patch directory.
!
! Shadows pm/rho_fine.f90 (rho_fine -> rho_fine_reference by #define + #include;
! multipole_fine, cic_from_multipole, ... stay the reference's).  rho_fine's hydro
! deposit reads the density uold(:,1) (multipole_fine, pm/rho_fine.f90:666-820):
! while the level is device-resident the new rho_fine first brings that one
! variable back to the host array, then runs the untouched reference routine.
!==============================================================================
#define rho_fine rho_fine_reference
#include "pm/rho_fine.f90"
#undef rho_fine

subroutine rho_fine(ilevel,icount)
  use amr_commons
  use hydro_commons
  use ramses_amd_iface
  implicit none
  integer::ilevel,icount
  integer::rc
  if(poisson.and.hydro)then
     if(ramses_amd_resident())then
        rc=ramses_amd_resident_sync_density_f90(uold)
        if(rc/=0)call ramses_amd_fatal('rho_fine (density of the resident level)')
     end if
  end if
  call rho_fine_reference(ilevel,icount)
end subroutine rho_fine
