!==============================================================================
! godunov_fine.f90 of the ramses_amd MHD patch directory (make SOLVER=mhd PATCH=.../ramses_amd/patch_mhd).
!
! Shadows mhd/godunov_fine.f90: the untouched reference file is pulled in by the preprocessor with godunov_fine, set_unew
! and set_uold renamed to *_reference (the source terms and godfine1 stay the reference's); the routines of the same names
! below keep the reference's names, arguments and meaning (mhd/godunov_fine.f90:5-35, 40-109, 185-281).
!   godunov_fine(ilevel) hands a fully refined periodic level of a single-rank run to the MI355X sweep through the C ABI.
!     Run with ONE level (levelmin = nlevelmax): the level stays on the device between courant_fine, godunov_fine and
!     set_uold (ramses_amd_mhd_iface: ramses_amd_mhd_resident; set_unew is implied by the sweep, set_uold is a swap of two
!     bricks, backup_hydro fetches the state back: output_hydro.f90 of this directory).  Otherwise staged: uold of the
!     level's cells goes up, unew comes back (SURVEY.md 8 row f4).
!     Any other level of an AMR tree (partly refined, or with finer levels inside): godfine1 on the device, oct by oct with
!     the 6^3 stencil, the divergence-free interpolation of missing neighbour octs and the flux / EMF corrections of the
!     coarser level in the reference's order (ramses_amd_mhd_godunov_amr below, csrc/mhd_amr.hip; staged).
! Everything the device does not cover -- several ranks, physical boundaries, self-gravity, pressure_fix,
! NENER>0, passive scalars, the solvers and slope types outside ramses_amd_mhd_params -- takes the reference's routines.
!==============================================================================
#define godunov_fine godunov_fine_reference
#define set_unew set_unew_reference
#define set_uold set_uold_reference
#include "mhd/godunov_fine.f90"
#undef godunov_fine
#undef set_unew
#undef set_uold

subroutine set_unew(ilevel)
  use amr_commons
  use ramses_amd_mhd_iface
  implicit none
  integer::ilevel
  ! resident level: the sweep writes uold + the updates of godfine1 into the second brick
  if(ramses_amd_mhd_resident())then
     if(int(active(ilevel)%ngrid,8)*8_8==(2_8**ilevel)**3.and.ilevel==levelmin)return
  end if
  call set_unew_reference(ilevel)
end subroutine set_unew

subroutine set_uold(ilevel)
  use amr_commons
  use ramses_amd_mhd_iface
  implicit none
  integer::ilevel,rc
  if(ramses_amd_mhd_resident())then
     if(int(active(ilevel)%ngrid,8)*8_8==(2_8**ilevel)**3.and.ilevel==levelmin)then
        rc=ramses_amd_mhd_resident_set_uold_f90(ilevel)
        if(rc/=0)call ramses_amd_mhd_fatal('set_uold')
        return
     end if
  end if
  call set_uold_reference(ilevel)
end subroutine set_uold

subroutine godunov_fine(ilevel)
  use amr_commons
  use hydro_commons
  use ramses_amd_mhd_iface
  implicit none
  integer::ilevel
  integer::rc,nx_loc,i
  real(dp)::dx,scale
  type(ramses_amd_mhd_params)::p
  integer,allocatable,dimension(:)::octs
  logical::dev
  if(numbtot(1,ilevel)==0)return
  if(static)return
  nx_loc=icoarse_max-icoarse_min+1
  dev=ramses_amd_mhd_device_config().and.ilevel>=2.and.int(active(ilevel)%ngrid,8)*8_8==(2_8**ilevel)**3
  if(dev)then
     ! a finer level exists: its coarse-fine corrections are in unew already and godfine1 masks refined cells (:760-903)
     if(ilevel<nlevelmax)then
        if(numbtot(1,ilevel+1)>0)dev=.false.
     end if
  end if
  if(.not.dev)then
     ! any other level of the tree: godfine1 as the reference writes it, on the device (csrc/mhd_amr.hip; staged)
     if(ramses_amd_mhd_amr_config().and.ilevel>=3)then
        call ramses_amd_mhd_godunov_amr(ilevel)
        return
     end if
     ! nothing silent: the library counts the levels that take the reference's host routine and prints them at exit
     if(ramses_amd_mhd_enabled())rc=ramses_amd_mhd_note_reference_sweep(ilevel)
     call godunov_fine_reference(ilevel)
     return
  end if
  if(verbose)write(*,111)ilevel
  call ramses_amd_mhd_fill_params(p)
  scale=boxlen/dble(nx_loc)
  dx=0.5D0**ilevel*scale
  allocate(octs(active(ilevel)%ngrid))
  do i=1,active(ilevel)%ngrid
     octs(i)=active(ilevel)%igrid(i)
  end do
  if(ramses_amd_mhd_resident().and.ilevel==levelmin)then
     rc=ramses_amd_mhd_resident_godunov_f90(p,ilevel,active(ilevel)%ngrid,octs,xg,int(ngridmax,8),int(ncoarse,8),nx_loc, &
          & uold,dx,dtnew(ilevel))
  else
     ! (set_unew has just made unew = uold on the level; the sweep returns uold + the updates of godfine1)
     rc=ramses_amd_mhd_godunov_fine_f90(p,ilevel,active(ilevel)%ngrid,octs,xg,int(ngridmax,8),int(ncoarse,8),nx_loc, &
          & uold,unew,dx,dtnew(ilevel))
     if(ramses_amd_mhd_first)then
        write(*,*)'ramses_amd: MHD godunov_fine of fully refined levels on the MI355X (staged)'
        ramses_amd_mhd_first=.false.
     end if
  end if
  deallocate(octs)
  if(rc/=0)call ramses_amd_mhd_fatal('godunov_fine')
111 format('   Entering godunov_fine (MHD, MI355X) for level ',i2)
end subroutine godunov_fine

!------------------------------------------------------------------------------
! godunov_fine(ilevel) of a level of an AMR tree: the whole list of active octs in one call (the library follows the
! reference's batches of nvector octs when it adds the corrections of the coarser level)
!------------------------------------------------------------------------------
subroutine ramses_amd_mhd_godunov_amr(ilevel)
  use amr_commons
  use hydro_commons
  use poisson_commons
  use ramses_amd_mhd_iface
  implicit none
  integer::ilevel
  integer::rc,nx_loc,i,usef
  real(dp)::dx,scale
  type(ramses_amd_mhd_params)::p
  integer,allocatable,dimension(:)::octs
  if(verbose)write(*,112)ilevel
  call ramses_amd_mhd_fill_params(p)
  nx_loc=icoarse_max-icoarse_min+1
  scale=boxlen/dble(nx_loc)
  dx=0.5D0**ilevel*scale
  allocate(octs(active(ilevel)%ngrid))
  do i=1,active(ilevel)%ngrid
     octs(i)=active(ilevel)%igrid(i)
  end do
  usef=0
  if(poisson)usef=1
  if(poisson)then
     rc=ramses_amd_mhd_godunov_fine_amr_f90(p,ilevel,levelmin,active(ilevel)%ngrid,octs,son,nbor,father,int(ngridmax,8), &
          & int(ncoarse,8),uold,unew,f,usef,dx,dtnew(ilevel),nvector,interpol_var,interpol_type,interpol_mag_type)
  else
     rc=ramses_amd_mhd_godunov_fine_amr_f90(p,ilevel,levelmin,active(ilevel)%ngrid,octs,son,nbor,father,int(ngridmax,8), &
          & int(ncoarse,8),uold,unew,uold,usef,dx,dtnew(ilevel),nvector,interpol_var,interpol_type,interpol_mag_type)
  end if
  deallocate(octs)
  if(rc/=0)call ramses_amd_mhd_fatal('godunov_fine (AMR level)')
  if(ramses_amd_mhd_first)then
     write(*,*)'ramses_amd: MHD godunov_fine of AMR levels on the MI355X (staged)'
     ramses_amd_mhd_first=.false.
  end if
112 format('   Entering godunov_fine (MHD, AMR level, MI355X) for level ',i2)
end subroutine ramses_amd_mhd_godunov_amr
