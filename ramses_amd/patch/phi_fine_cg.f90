!==============================================================================
! phi_fine_cg.f90 of the ramses_amd patch directory.
!
! Shadows poisson/phi_fine_cg.f90 (bin/Makefile:153 VPATH).  The untouched
! reference file is pulled in by the preprocessor with phi_fine_cg renamed to
! phi_fine_cg_reference, so cmp_residual_cg, cmp_Ap_cg, make_initial_phi and
! make_multipole_phi stay the reference's; the new phi_fine_cg(ilevel,icount)
! keeps the reference's name, arguments and meaning and runs the iteration
! loop of the conjugate-gradient solver on the MI355X through the C ABI.
!==============================================================================
#define phi_fine_cg phi_fine_cg_reference
#include "poisson/phi_fine_cg.f90"
#undef phi_fine_cg

subroutine phi_fine_cg(ilevel,icount)
  use amr_commons
  use pm_commons
  use poisson_commons
  use constants, only: twopi
  use ramses_amd_iface
  implicit none
  integer::ilevel,icount
  !--------------------------------------------------------------------------
  ! Same contract as the reference (poisson/phi_fine_cg.f90:5-206): on entry rho
  ! and rho_tot hold the source, on exit phi the potential of the level and
  ! f(:,1:3) the solver's r, p, Ap.  The per-solve preparation (:52-85: initial
  ! guess interpolated from the coarser level, boundaries, first residual with
  ! interpol_phi along the level's edge) is the reference's own host code, the
  ! loop (:88-187) runs on the device.
  !--------------------------------------------------------------------------
  integer::rc,nx_loc,iter,itermax
  real(dp)::dx2,fourpi,scale,oneoversix,fact
  real(kind=8),dimension(1:3)::err

  if(gravity_type>0)return
  if(numbtot(1,ilevel)==0)return

  if(.not.ramses_amd_enabled())then
     call phi_fine_cg_reference(ilevel,icount)
     return
  end if
  if(verbose)write(*,111)ilevel
  call ramses_amd_need_ndim3('phi_fine_cg')

  ! What the device path does not implement stops the run (no silent fallback)
  if(ncpu>1)then
     write(*,*)'ramses_amd: the device conjugate-gradient solver handles single-rank runs; got ncpu=',ncpu
     call ramses_amd_fatal('phi_fine_cg (several ranks)')
  end if

  dx2=(0.5D0**ilevel)**2
  nx_loc=icoarse_max-icoarse_min+1
  scale=boxlen/dble(nx_loc)
  fourpi=2*twopi*scale
  if(cosmo)fourpi=1.5D0*omega_m*aexp*scale
  oneoversix=1.0D0/dble(twondim)
  fact=oneoversix*fourpi*dx2

  if(ilevel>levelmin)then
     call make_initial_phi(ilevel,icount)
  else
     call make_multipole_phi(ilevel)
  endif
  call make_virtual_fine_dp(phi(1),ilevel)
  call make_boundary_phi(ilevel)
  call cmp_residual_cg(ilevel,icount)

  itermax=10000
  rc=ramses_amd_cg_solve_host(ilevel,active(ilevel)%ngrid,active(ilevel)%igrid,son,nbor,int(ngridmax,8),int(ncoarse,8), &
       & phi,f,rho,rho_tot,fact,dble(twotondim)*dble(numbtot(1,ilevel)),epsilon,itermax,-1,iter,err)
  if(rc/=0)call ramses_amd_fatal('phi_fine_cg')

  if(myid==1)write(*,115)ilevel,iter,err(1)/err(3),err(1)/err(2)
  if(iter >= itermax)then
     if(myid==1)write(*,*)'Poisson failed to converge...'
  end if

  call make_virtual_fine_dp(phi(1),ilevel)

111 format('   Entering phi_fine_cg (MI355X) for level ',I2)
115 format('   ==> Level=',i5,' Step=',i5,' Error=',2(1pe10.3,1x))

end subroutine phi_fine_cg
