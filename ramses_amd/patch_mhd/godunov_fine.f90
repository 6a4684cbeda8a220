!==============================================================================
! godunov_fine.f90 of the ramses_amd MHD patch directory (make SOLVER=mhd PATCH=.../ramses_amd/patch_mhd).
!
! Shadows mhd/godunov_fine.f90: the untouched reference file is pulled in by the preprocessor with godunov_fine renamed
! to godunov_fine_reference (set_unew, set_uold, the source terms and godfine1 stay the reference's); the new
! godunov_fine(ilevel) keeps the reference's name, argument and meaning (mhd/godunov_fine.f90:5-35) and hands a fully
! refined periodic level of a single-rank run to the MI355X sweep through the C ABI: uold(1:ncell,1:nvar+3) of the
! level's cells goes up, unew comes back (staged; SURVEY.md 8 row f4).  Everything the device does not cover -- AMR
! levels, several ranks, physical boundaries, self-gravity, pressure_fix, NENER>0, passive scalars, the solvers and
! slope types outside ramses_amd_mhd_params -- takes the reference's routine.
!==============================================================================
#define godunov_fine godunov_fine_reference
#include "mhd/godunov_fine.f90"
#undef godunov_fine

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
  dev=ramses_amd_mhd_enabled().and.hydro.and.ndim==3.and.nvar==8.and.ncpu==1.and.nboundary==0.and.nx_loc==1 &
       & .and.jcoarse_max==jcoarse_min.and.kcoarse_max==kcoarse_min.and..not.poisson.and..not.pressure_fix &
       & .and.ischeme==0.and..not.allow_switch_solver.and..not.allow_switch_solver2D.and.ilevel>=2 &
       & .and.int(active(ilevel)%ngrid,8)*8_8==(2_8**ilevel)**3
  if(dev)then
     ! a finer level exists: its coarse-fine corrections are in unew already and godfine1 masks refined cells (:760-903)
     if(ilevel<nlevelmax)then
        if(numbtot(1,ilevel+1)>0)dev=.false.
     end if
  end if
  if(dev)dev=(iriemann==0.or.iriemann==2.or.iriemann==3.or.iriemann==4).and.(iriemann2d==0.or.iriemann2d==3.or.iriemann2d==5) &
       & .and.(slope_type==0.or.slope_type==1.or.slope_type==2.or.slope_type==7.or.slope_type==8) &
       & .and.(slope_mag_type==0.or.slope_mag_type==1.or.slope_mag_type==2.or.slope_mag_type==7.or.slope_mag_type==8)
  if(.not.dev)then
     call godunov_fine_reference(ilevel)
     return
  end if
  if(verbose)write(*,111)ilevel
  p%gamma=gamma; p%smallr=smallr; p%smallc=smallc; p%slope_theta=slope_theta
  p%slope_type=slope_type; p%slope_mag_type=slope_mag_type; p%riemann=iriemann; p%riemann2d=iriemann2d
  scale=boxlen/dble(nx_loc)
  dx=0.5D0**ilevel*scale
  allocate(octs(active(ilevel)%ngrid))
  do i=1,active(ilevel)%ngrid
     octs(i)=active(ilevel)%igrid(i)
  end do
  ! (set_unew has just made unew = uold on the level; the sweep returns uold + the updates of godfine1)
  rc=ramses_amd_mhd_godunov_fine_f90(p,ilevel,active(ilevel)%ngrid,octs,xg,int(ngridmax,8),int(ncoarse,8),nx_loc, &
       & uold,unew,dx,dtnew(ilevel))
  deallocate(octs)
  if(rc/=0)call ramses_amd_mhd_fatal('godunov_fine')
  if(ramses_amd_mhd_first)then
     write(*,*)'ramses_amd: MHD godunov_fine of fully refined levels on the MI355X (staged)'
     ramses_amd_mhd_first=.false.
  end if
111 format('   Entering godunov_fine (MHD, MI355X) for level ',i2)
end subroutine godunov_fine
