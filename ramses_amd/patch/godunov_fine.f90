!==============================================================================
! godunov_fine.f90 of the ramses_amd patch directory (make PATCH=.../ramses_amd/patch).
!
! RAMSES shadows whole files (bin/Makefile:153 VPATH), so this file must define
! every external symbol of hydro/godunov_fine.f90: godunov_fine, set_unew,
! set_uold, add_gravity_source_terms, add_pdv_source_terms, godfine1.  The
! untouched reference file is pulled in by the preprocessor with its
! godunov_fine renamed to godunov_fine_reference (no reference source is
! copied); the new godunov_fine below keeps the reference's name, argument and
! meaning and hands the level to the MI355X sweep through the C ABI.
!
! Needs -I<ramses root> on the compile line (the patch Makefile adds -I..).
!==============================================================================
#define godunov_fine godunov_fine_reference
#include "hydro/godunov_fine.f90"
#undef godunov_fine

subroutine godunov_fine(ilevel)
  use amr_commons
  use hydro_commons
  use poisson_commons
  use ramses_amd_iface
  implicit none
  integer::ilevel
  !--------------------------------------------------------------------------
  ! Same contract as the reference (hydro/godunov_fine.f90:5-35): on entry
  ! hydro variables are in uold and unew was prepared by set_unew; on exit
  ! unew of the level's active cells has been updated by the second-order
  ! Godunov fluxes.  The sweep itself runs on the GPU.
  !--------------------------------------------------------------------------
  type(ramses_amd_hydro_params)::p
  integer::rc,nx_loc,has_f
  real(dp)::scale,dx

  if(numbtot(1,ilevel)==0)return
  if(static)return

  if(.not.ramses_amd_enabled())then
     call godunov_fine_reference(ilevel)
     return
  end if
  if(verbose)write(*,111)ilevel

  ! What the device path does not implement stops the run (no silent fallback)
  if(ncpu>1)then
     write(*,*)'ramses_amd: godunov_fine on the device needs one rank per level brick; ncpu=',ncpu
     call ramses_amd_fatal('godunov_fine (ncpu>1)')
  end if
  if(pressure_fix.or.MC_tracer.or.momentum_feedback>0.or.strict_equilibrium>0)then
     write(*,*)'ramses_amd: pressure_fix/MC_tracer/momentum_feedback/strict_equilibrium are not on the device'
     call ramses_amd_fatal('godunov_fine (unsupported option)')
  end if

  call ramses_amd_fill_hydro_params(p)
  nx_loc=icoarse_max-icoarse_min+1
  scale=boxlen/dble(nx_loc)
  dx=0.5d0**ilevel*scale

  if(poisson)then
     has_f=1
     rc=ramses_amd_godunov_fine_f90(p,ilevel,active(ilevel)%ngrid,active(ilevel)%igrid,xg, &
          & int(ngridmax,8),int(ncoarse,8),nx_loc,uold,unew,f,has_f,dx,dtnew(ilevel))
  else
     has_f=0
     rc=ramses_amd_godunov_fine_f90(p,ilevel,active(ilevel)%ngrid,active(ilevel)%igrid,xg, &
          & int(ngridmax,8),int(ncoarse,8),nx_loc,uold,unew,uold,has_f,dx,dtnew(ilevel))
  end if
  if(rc/=0)call ramses_amd_fatal('godunov_fine')

111 format('   Entering godunov_fine (MI355X) for level ',i2)

end subroutine godunov_fine
