!==============================================================================
! synchro_hydro_fine.f90 of the ramses_amd patch directory.
!
! Shadows hydro/synchro_hydro_fine.f90 (synchro_hydro_fine -> *_reference by
! #define + #include; synchydrofine1 stays the reference's).  While the level is
! device-resident (ramses_amd_iface: ramses_amd_resident) the gravity kick of the
! momenta and the energy runs on the resident brick; otherwise the reference.
!==============================================================================
#define synchro_hydro_fine synchro_hydro_fine_reference
#include "hydro/synchro_hydro_fine.f90"
#undef synchro_hydro_fine

subroutine synchro_hydro_fine_amd(ilevel,dteff,which_force)
  use amr_commons
  use hydro_commons
  use poisson_commons
#if USE_TURB==1
  use turb_commons
#endif
  use ramses_amd_iface
  implicit none
  integer::ilevel
  real(dp)::dteff
  integer::which_force !gravity=1, turbulence=2
  !--------------------------------------------------------------------------
  ! Same contract as the reference (hydro/synchro_hydro_fine.f90:5-40): uold of
  ! the level's cells receives rho*f*dteff on the momenta, the total energy
  ! follows (internal energy unchanged).
  !--------------------------------------------------------------------------
  type(ramses_amd_hydro_params)::p
  integer::rc,nx_loc

  ! the reference's own early returns (hydro/synchro_hydro_fine.f90:21-26): a USE_TURB=1 build
  ! also enters for the turbulent forcing (which_force=2) without self-gravity
#if USE_TURB==1
  if(.not.(poisson.or.turb))return
#else
  if(.not.poisson)return
  if(numbtot(1,ilevel)==0)return
#endif
  if(which_force==1.and.poisson.and.ramses_amd_amr_resident())then
     ! AMR run with the state on the device: the kick on the resident arrays (the acceleration was mirrored by force_fine)
     call ramses_amd_amr_ensure()
     call ramses_amd_fill_hydro_params(p)
     rc=ramses_amd_amrres_synchro(p,active(ilevel)%ngrid,ramses_amd_octs(ilevel),dteff)
     if(rc/=0)call ramses_amd_fatal('synchro_hydro_fine')
     return
  end if
  if(which_force/=1.or..not.poisson.or..not.ramses_amd_resident())then
     call synchro_hydro_fine_reference(ilevel,dteff,which_force)
     return
  end if
  if(verbose)write(*,111)ilevel

  call ramses_amd_fill_hydro_params(p)
  nx_loc=icoarse_max-icoarse_min+1
  rc=ramses_amd_resident_synchro_f90(p,ilevel,active(ilevel)%ngrid,ramses_amd_octs(ilevel),xg, &
       & int(ngridmax,8),int(ncoarse,8),nx_loc,uold,f,dteff)
  if(rc/=0)call ramses_amd_fatal('synchro_hydro_fine')

111 format('   Entering synchro_hydro_fine (MI355X) for level',i2)

end subroutine synchro_hydro_fine_amd

subroutine synchro_hydro_fine(ilevel,dteff,which_force)
  use amr_parameters, only: dp
  use ramses_amd_iface
  implicit none
  integer::ilevel,which_force
  real(dp)::dteff
  integer(8)::t0
  call ramses_amd_tic(t0)
  call synchro_hydro_fine_amd(ilevel,dteff,which_force)
  call ramses_amd_toc('synchro_hydro_fine',ilevel,t0)
end subroutine synchro_hydro_fine
