!==============================================================================
! multigrid_fine_commons.f90 of the ramses_amd patch directory.
!
! Shadows poisson/multigrid_fine_commons.f90 (bin/Makefile:153 VPATH).  The
! untouched reference file is pulled in by the preprocessor with
! multigrid_fine renamed to multigrid_fine_reference, so every other symbol of
! that file (recursive_multigrid_coarse, build_parent_comms_mg, make_fine_mask,
! make_fine_bc_rhs, make_virtual_mg_int, ...) stays the reference's; make_virtual_mg_dp
! and make_reverse_mg_dp are shadowed the same way (below); the new
! multigrid_fine(ilevel,icount) keeps the reference's name, arguments and
! meaning and runs the V-cycles on the MI355X through the C ABI.
!==============================================================================
! (multigrid_fine_commons_ref.f90: the reference's file with the DEFINITIONS of make_virtual_mg_dp and make_reverse_mg_dp
!  renamed to *_reference, generated into the build directory by prepare.sh -- their callers live in the same file, where a
!  preprocessor rename would catch the calls too)
#define multigrid_fine multigrid_fine_reference
#include "multigrid_fine_commons_ref.f90"
#undef multigrid_fine

!------------------------------------------------------------------------------
! Virtual boundaries of the multigrid levels (poisson/multigrid_fine_commons.f90:1172-1290, 1378-1475), same names,
! arguments and meaning.  While a solve runs on the device with several ranks and its levels are resident there
! (ramses_amd_mg_mpi_resident), active_mg(:,ilevel)%u(:,ivar) lives on the device: the exchange gathers the emission cells
! there, moves one message per peer (RCCL, or this program's MPI on pinned buffers when ranks share a GPU) and drops /
! adds what arrives on the device -- the reverse exchange peer by peer in icpu order like the reference.  Before the first
! device routine of a solve (the masks, ivar=4) and otherwise: the reference's routines.
!------------------------------------------------------------------------------
subroutine make_virtual_mg_dp(ivar,ilevel)
  use amr_commons
  use ramses_amd_iface
  implicit none
  integer::ilevel,ivar
#ifndef WITHOUTMPI
  if(ramses_amd_mg_active.and.ramses_amd_mg_started.and.ramses_amd_mg_mpi_resident.and.ncpu>1)then
     call ramses_amd_mg_halo(ilevel,ivar,0)
     return
  end if
#endif
  call make_virtual_mg_dp_reference(ivar,ilevel)
end subroutine make_virtual_mg_dp

subroutine make_reverse_mg_dp(ivar,ilevel)
  use amr_commons
  use ramses_amd_iface
  implicit none
  integer::ilevel,ivar
#ifndef WITHOUTMPI
  if(ramses_amd_mg_active.and.ramses_amd_mg_started.and.ramses_amd_mg_mpi_resident.and.ncpu>1)then
     call ramses_amd_mg_halo(ilevel,ivar,1)
     return
  end if
#endif
  call make_reverse_mg_dp_reference(ivar,ilevel)
end subroutine make_reverse_mg_dp

subroutine multigrid_fine_amd(ilevel,icount)
  use amr_commons
  use poisson_commons
  use poisson_parameters
  use constants, only: twopi
  use ramses_amd_iface
  implicit none
  integer, intent(in) :: ilevel,icount
  !--------------------------------------------------------------------------
  ! Same contract as the reference (poisson/multigrid_fine_commons.f90:25-296):
  ! on entry rho and rho_tot hold the source; on exit phi holds the potential
  ! of the level.  f(:,1:3) is scratch in the reference (force_fine overwrites
  ! it next) and is left untouched here.
  !--------------------------------------------------------------------------
  integer::rc,nx_loc,isafe,iters,interp
  real(dp)::scale,fourpi,tfrac
  real(kind=8)::err
  logical::dist

  if(gravity_type>0)return
  if(numbtot(1,ilevel)==0)return

  if(.not.ramses_amd_enabled())then
     call multigrid_fine_reference(ilevel,icount)
     return
  end if
  if(verbose) print '(A,I2)','Entering fine multigrid (MI355X) at level ',ilevel
  call ramses_amd_need_ndim3('multigrid_fine')

  ! An AMR level (it does not cover the box, or it is not levelmin): the reference's own
  ! driver and per-solve setup run on the host, the compute routines it calls (shadowed by
  ! multigrid_fine_fine.f90 / multigrid_fine_coarse.f90 of this directory) on the device
  ! (a box with physical boundaries: the Dirichlet values enter through the masks and the
  !  right-hand side the reference prepares, so every level takes this path)
  nx_loc=icoarse_max-icoarse_min+1
  ! several MPI ranks, levelmin fully refined and periodic, every rank's domain a power-of-two box: the dense V-cycles of
  ! the single-rank path, distributed (one brick per rank, one deep-halo exchange per smoother launch, coarse levels
  ! replicated: csrc/mg_dist.hip)
  if(ncpu>1.and.ilevel==levelmin.and.nboundary==0.and.nx_loc==1.and.jcoarse_max==jcoarse_min.and.kcoarse_max==kcoarse_min)then
     call ramses_amd_mgdist_multigrid(ilevel,dist,iters,err)
     if(dist)then
        ! what the reference's per-solve setup leaves behind for an unmasked periodic box: the coarsest multigrid level is 1
        ! (poisson/multigrid_fine_commons.f90:134-180)
        levelmin_mg=1
        if(myid==1) print '(A,I5,A,I5,A,1pE10.3)','   ==> Level=',ilevel, ' Step=', &
             iters,' Error=',err
        if(myid==1 .and. iters==ramses_amd_mg_maxiter) print *,'WARN: Fine multigrid Poisson failed to converge...'
        return
     end if
  end if
  ! (several MPI ranks otherwise: every level takes this path -- the reference's driver with its halo exchanges,
  !  each compute routine on the rank's GPU over its own octs and the reception octs of its neighbours)
  if(ilevel>levelmin.or.nboundary>0.or.ncpu>1.or.nx_loc/=1.or.jcoarse_max/=jcoarse_min.or.kcoarse_max/=kcoarse_min &
       & .or.int(active(ilevel)%ngrid,8)*8_8/=(2_8**ilevel)**3*int(nx_loc,8)**3)then
     ! periodic box of one coarse cell: driver and per-solve setup on the device too (csrc/pois_amr.hip); only
     ! rho of the level and phi, phi_old of the level above travel in, phi of the level out
     ! (RAMSES_AMD_MG_DRIVER=host: the reference's driver and setup with the device operators, as with walls)
     if(nboundary==0.and.ncpu==1.and.ncoarse==1.and.ilevel>1.and.ramses_amd_mg_device_driver())then
        if(nremap>0)ramses_amd_tree_epoch=ramses_amd_tree_epoch+1     ! (defrag may renumber the octs)
        rc=ramses_amd_poisamr_tree(ramses_amd_tree_epoch,int(ngridmax,8),int(ncoarse,8),son,nbor,father)
        if(rc/=0)call ramses_amd_fatal('multigrid_fine (AMR level, tree)')
        scale=boxlen/dble(nx_loc)
        fourpi=2*twopi*scale
        if(cosmo)fourpi=1.5D0*omega_m*aexp*scale
        interp=0
        tfrac=0.0d0
        if(ilevel>levelmin)then
           interp=1
           if(icount/=1.and.icount/=2)then
              write(*,*)'icount has bad value'
              call clean_stop
           end if
           if(dtold(ilevel-1)>0)tfrac=1d0*dtnew(ilevel)/dtold(ilevel-1)*(icount-1)
        end if
        isafe=0
        if(safe_mode(ilevel))isafe=1
        rc=ramses_amd_poisamr_multigrid(ilevel,active(ilevel)%ngrid,ramses_amd_octs(ilevel), &
             & active(ilevel-1)%ngrid,ramses_amd_octs(ilevel-1),phi,phi_old,rho,flag2(1),rho_tot,fourpi,tfrac,interp, &
             & epsilon,ngs_fine,ngs_coarse,ncycles_coarse_safe,isafe,iters,err)
        if(rc/=0)call ramses_amd_fatal('multigrid_fine (AMR level)')
        safe_mode(ilevel)=(isafe/=0)
        ramses_amd_pois_amr_level=ilevel
        if(myid==1) print '(A,I5,A,I5,A,1pE10.3)','   ==> Level=',ilevel, ' Step=', &
             iters,' Error=',err
        if(myid==1 .and. iters==ramses_amd_mg_maxiter) print *,'WARN: Fine multigrid Poisson failed to converge...'
        return
     end if
     ramses_amd_mg_active=.true.
     ramses_amd_mg_started=.false.
     ramses_amd_mg_level=ilevel
     call multigrid_fine_reference(ilevel,icount)
     if(ramses_amd_mg_started)then
        rc=ramses_amd_mgamr_end()
        if(rc/=0)call ramses_amd_fatal('multigrid_fine (AMR level, end)')
     end if
     ramses_amd_mg_active=.false.
     return
  end if

  ! What the device path does not implement stops the run (no silent fallback)
  if(ncpu>1.or.nboundary>0)then
     write(*,*)'ramses_amd: device multigrid_fine handles periodic single-rank runs;'
     write(*,*)'            got ncpu=',ncpu,' nboundary=',nboundary
     call ramses_amd_fatal('multigrid_fine (several ranks / physical boundaries)')
  end if

  nx_loc=icoarse_max-icoarse_min+1
  scale=boxlen/dble(nx_loc)
  fourpi=2*twopi*scale
  if(cosmo)fourpi=1.5D0*omega_m*aexp*scale

  isafe=0
  if(safe_mode(ilevel))isafe=1
  if(ramses_amd_pois_dev)then
     ! the source is already on the device (rho_fine shim), phi stays there for force_fine
     rc=ramses_amd_resident_multigrid_f90(ilevel,rho_tot,fourpi,epsilon,isafe,iters,err)
  else
     rc=ramses_amd_multigrid_fine_f90(ilevel,active(ilevel)%ngrid,ramses_amd_octs(ilevel),xg, &
          & int(ngridmax,8),int(ncoarse,8),nx_loc,rho,phi,rho_tot,fourpi,epsilon,isafe,iters,err)
  end if
  if(rc/=0)call ramses_amd_fatal('multigrid_fine')
  safe_mode(ilevel)=(isafe/=0)

  if(myid==1) print '(A,I5,A,I5,A,1pE10.3)','   ==> Level=',ilevel, ' Step=', &
       iters,' Error=',err
  if(myid==1 .and. iters==ramses_amd_mg_maxiter) print *,'WARN: Fine multigrid Poisson failed to converge...'

end subroutine multigrid_fine_amd


subroutine multigrid_fine(ilevel,icount)
  use ramses_amd_iface
  implicit none
  integer::ilevel,icount
  integer(8)::t0
  call ramses_amd_tic(t0)
  call multigrid_fine_amd(ilevel,icount)
  call ramses_amd_toc('multigrid_fine',ilevel,t0)
end subroutine multigrid_fine
