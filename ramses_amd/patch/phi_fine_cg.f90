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
#ifndef WITHOUTMPI
  if(ncpu>1)then
     call phi_fine_cg_mpi(ilevel,fact,itermax,iter,err)
     if(myid==1)write(*,115)ilevel,iter,err(1)/err(3),err(1)/err(2)
     if(iter >= itermax)then
        if(myid==1)write(*,*)'Poisson failed to converge...'
     end if
     call make_virtual_fine_dp(phi(1),ilevel)
     return
  end if
#endif
  rc=ramses_amd_cg_solve_host(ilevel,active(ilevel)%ngrid,ramses_amd_octs(ilevel),son,nbor,int(ngridmax,8),int(ncoarse,8), &
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

#ifndef WITHOUTMPI
!------------------------------------------------------------------------------
! The iteration loop of phi_fine_cg (poisson/phi_fine_cg.f90:88-187) with several MPI ranks: the loop and its two
! MPI_ALLREDUCEs per iteration (:108,154) here; every loop body -- the recurrence on p, cmp_Ap_cg with the local p.Ap,
! the recurrences on x and r with the local r.r -- on the rank's GPU (ramses_amd_cgmpi_*), and the halo exchange of p
! (:134) on the device vector (ramses_amd_cg_p_halo: RCCL, or MPI on pinned buffers).  alpha and beta are formed on the
! device from the device scalars this routine keeps global.  The local sums run in the reference's order (default) and
! the same MPI library reduces them: the run equals the MPI reference bit for bit.
!------------------------------------------------------------------------------
subroutine phi_fine_cg_mpi(ilevel,fact,itermax,iter,err)
  use amr_commons
  use poisson_commons
  use mpi_mod
  use ramses_amd_iface
  implicit none
  integer::ilevel,itermax,iter
  real(dp)::fact
  real(kind=8),dimension(1:3)::err
  integer::rc,info,nem,nrc
  integer,allocatable,dimension(:)::em_n,em_ig,rc_n,rc_ig
  real(kind=8)::error,error_ini,rhs_norm,r2,pAp,x_all
  real(kind=8),dimension(2)::out2,out2_all

  call ramses_amd_comm_lists(ilevel,em_n,em_ig,rc_n,rc_ig)
  nem=sum(em_n); nrc=sum(rc_n)
  rc=ramses_amd_cgmpi_begin(ilevel,active(ilevel)%ngrid,ramses_amd_octs(ilevel),son,nbor,int(ngridmax,8),int(ncoarse,8), &
       & phi,f,rho,rho_tot,fact,-1,out2)
  if(rc/=0)call ramses_amd_fatal('phi_fine_cg (begin)')
  call MPI_ALLREDUCE(out2,out2_all,2,MPI_DOUBLE_PRECISION,MPI_SUM,MPI_COMM_WORLD,info)
  rhs_norm=DSQRT(out2_all(1)/dble(twotondim*numbtot(1,ilevel)))
  r2=out2_all(2)
  rc=ramses_amd_cgmpi_set(0,r2)
  if(rc/=0)call ramses_amd_fatal('phi_fine_cg (r2)')
  ! the halo of p (:134) runs on the device vector: the level's communicators go there once per solve
  rc=ramses_amd_cgmpi_comm_set(ncpu,em_n,em_ig,rc_n,rc_ig)
  if(rc/=0)call ramses_amd_fatal('phi_fine_cg (communicators)')

  iter=0
  error=1.0D0; error_ini=1.0D0
  do while(error>epsilon*error_ini.and.iter<itermax)
     iter=iter+1
     ! recurrence on p (beta = r2/r2_old on the device), then its virtual cells
     rc=ramses_amd_cgmpi_step(0,iter)
     if(rc/=0)call ramses_amd_fatal('phi_fine_cg (recurrence on p)')
     call ramses_amd_cg_p_halo()
     rc=0
     ! z = A p and p.Ap
     if(rc==0)rc=ramses_amd_cgmpi_step(1,iter)
     if(rc==0)rc=ramses_amd_cgmpi_get(2,pAp)
     if(rc/=0)call ramses_amd_fatal('phi_fine_cg (cmp_Ap_cg)')
     call MPI_ALLREDUCE(pAp,x_all,1,MPI_DOUBLE_PRECISION,MPI_SUM,MPI_COMM_WORLD,info)
     rc=ramses_amd_cgmpi_set(2,x_all)
     ! recurrences on x and r (alpha = r2/pAp on the device); the error of THIS iteration is that of the r2 it started from
     error=DSQRT(r2/dble(twotondim*numbtot(1,ilevel)))
     if(iter==1)error_ini=error
     if(rc==0)rc=ramses_amd_cgmpi_step(2,iter)
     if(rc==0)rc=ramses_amd_cgmpi_get(0,r2)
     if(rc/=0)call ramses_amd_fatal('phi_fine_cg (recurrences on x and r)')
     call MPI_ALLREDUCE(r2,x_all,1,MPI_DOUBLE_PRECISION,MPI_SUM,MPI_COMM_WORLD,info)
     r2=x_all
     rc=ramses_amd_cgmpi_set(0,r2)
     if(rc/=0)call ramses_amd_fatal('phi_fine_cg (r2)')
  end do
  rc=ramses_amd_cgmpi_end(phi,f)
  if(rc/=0)call ramses_amd_fatal('phi_fine_cg (end)')
  err(1)=error; err(2)=error_ini; err(3)=rhs_norm
end subroutine phi_fine_cg_mpi
#endif
