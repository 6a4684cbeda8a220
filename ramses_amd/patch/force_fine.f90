!==============================================================================
! force_fine.f90 of the ramses_amd patch directory.
!
! Shadows poisson/force_fine.f90 (bin/Makefile:153 VPATH).  The untouched
! reference file is pulled in by the preprocessor with force_fine renamed to
! force_fine_reference, so gradient_phi stays the reference's; the new
! force_fine(ilevel,icount) keeps the reference's name, arguments and meaning
! and computes the acceleration of a fully refined periodic level on the MI355X.
!==============================================================================
#define force_fine force_fine_reference
#include "poisson/force_fine.f90"
#undef force_fine

subroutine force_fine_amd(ilevel,icount)
  use amr_commons
  use pm_commons
  use poisson_commons
  use constants, only : twopi
  use ramses_amd_iface
  implicit none
  integer::ilevel,icount
  !--------------------------------------------------------------------------
  ! Same contract as the reference (poisson/force_fine.f90:5-194): on entry phi
  ! holds the potential of the level, on exit f(:,1:ndim) its acceleration,
  ! epot_tot has received the level's potential energy and rho_max(ilevel) the
  ! level's maximum density.  Device path: gravity_type = 0, one rank, periodic
  ! box, level fully refined (every neighbour exists: no interpol_phi), no sink
  ! particles; anything else is the reference's routine.
  !--------------------------------------------------------------------------
  integer::rc,nx_loc,has_son,fresh
  integer(8)::tm
  real(dp)::dx,dx_loc,scale,fact,fourpi,tfrac
  real(kind=8),dimension(2)::diag

  if(numbtot(1,ilevel)==0)return
  nx_loc=(icoarse_max-icoarse_min+1)
  fresh=0
  if(ramses_amd_pois_amr_level==ilevel)fresh=1
  ramses_amd_pois_amr_level=0
  ! an AMR level (not the whole box) of a periodic single-rank run: gradient_phi with interpol_phi at the level's
  ! edge on the device (csrc/pois_amr.hip), reusing what the multigrid driver left there
  if(ramses_amd_enabled().and.gravity_type==0.and.ncpu==1.and.nboundary==0.and..not.sink.and.ndim==3.and.ilevel>levelmin &
       & .and.ncoarse==1.and.ramses_amd_mg_device_driver() &
       & .and.int(active(ilevel)%ngrid,8)*8_8/=(2_8**ilevel)**3*int(nx_loc,8)**3)then
     if(verbose)write(*,111)ilevel
     if(nremap>0)ramses_amd_tree_epoch=ramses_amd_tree_epoch+1
     rc=ramses_amd_poisamr_tree(ramses_amd_tree_epoch,int(ngridmax,8),int(ncoarse,8),son,nbor,father)
     if(rc/=0)call ramses_amd_fatal('force_fine (AMR level, tree)')
     dx=0.5D0**ilevel
     scale=boxlen/dble(nx_loc)
     dx_loc=dx*scale
     fourpi=2*twopi
     if(cosmo)fourpi=1.5D0*omega_m*aexp
     fact=-dx_loc**ndim/fourpi/2.0D0
     if(icount/=1.and.icount/=2)then
        write(*,*)'icount has bad value'
        call clean_stop
     end if
     tfrac=0.0d0
     if(dtold(ilevel-1)>0)tfrac=1d0*dtnew(ilevel)/dtold(ilevel-1)*(icount-1)
     rc=ramses_amd_poisamr_force(ilevel,active(ilevel)%ngrid,ramses_amd_octs(ilevel),active(ilevel-1)%ngrid, &
          & ramses_amd_octs(ilevel-1),phi,phi_old,rho,f,tfrac,1,fresh,fact,diag)
     if(rc/=0)call ramses_amd_fatal('force_fine (AMR level)')
     epot_tot=epot_tot+diag(1)
     rho_max(ilevel)=diag(2)
     return
  end if
#ifndef WITHOUTMPI
  ! several ranks, AMR run resident on the GPUs, any level the distributed dense multigrid has not just solved: gradient_phi
  ! (with interpol_phi at the level's edge) of the rank's own octs on its GPU -- phi of the own and the virtual octs of the
  ! level and of the level above goes up from the host vectors, f of the own cells comes back; the halo of f and the two
  ! all-reduces of the diagnostics stay the reference's (poisson/force_fine.f90:137-139,181-186).  RAMSES_AMD_FORCE_MPI=0:
  ! the reference's host loops.
  if(ramses_amd_enabled().and.ncpu>1.and.gravity_type==0.and.nboundary==0.and..not.sink.and.ndim==3.and.ilevel>=levelmin &
       & .and.ilevel>=2.and.ncoarse==1.and.ramses_amd_mgdist_phi_level/=ilevel.and.ramses_amd_force_mpi_on())then
     if(ramses_amd_amr_resident())then
        if(verbose)write(*,111)ilevel
        ramses_amd_mgdist_phi_level=0
        call ramses_amd_tic(tm)
        call ramses_amd_force_fine_mpi(ilevel,icount)
        call ramses_amd_toc('force_fine (device, MPI)',ilevel,tm)
        return
     end if
  end if
#endif
  ! several ranks, right after the distributed dense multigrid of this level (patch/multigrid_fine_commons.f90): the
  ! potential still sits on the rank's brick on the device, gradient_phi runs there
  if(ramses_amd_enabled().and.ncpu>1.and.ramses_amd_mgdist_phi_level==ilevel.and.gravity_type==0.and.nboundary==0 &
       & .and..not.sink.and.ndim==3)then
     ramses_amd_mgdist_phi_level=0
     if(verbose)write(*,111)ilevel
     call ramses_amd_mgdist_force_fine(ilevel)
     return
  end if
  ramses_amd_mgdist_phi_level=0
  if(.not.ramses_amd_enabled().or.gravity_type>0.or.ncpu>1.or.nboundary>0.or.sink.or.ndim/=3.or.ilevel<2 &
       & .or.nx_loc/=1.or.int(active(ilevel)%ngrid,8)*8_8/=(2_8**ilevel)**3*int(nx_loc,8)**3)then
     call force_fine_reference(ilevel,icount)
     return
  end if
  if(verbose)write(*,111)ilevel

  ! the factor of the potential-energy diagnostic (:164-168)
  dx=0.5D0**ilevel
  scale=boxlen/dble(nx_loc)
  dx_loc=dx*scale
  fourpi=2*twopi
  if(cosmo)fourpi=1.5D0*omega_m*aexp
  fact=-dx_loc**ndim/fourpi/2.0D0

  ! (make_boundary_phi and the halos of f are no-ops in this configuration)
  if(ramses_amd_pois_dev)then
     ! phi and rho are on the device (multigrid_fine / rho_fine shims); f stays there for the hydro routines
     rc=ramses_amd_resident_force_fine_f90(ilevel,fact,diag)
  else
     has_son=0
     if(ilevel<nlevelmax)then
        if(numbtot(1,ilevel+1)>0)has_son=1
     end if
     rc=ramses_amd_force_fine_f90(ilevel,active(ilevel)%ngrid,ramses_amd_octs(ilevel),xg, &
          & int(ngridmax,8),int(ncoarse,8),nx_loc,phi,f,rho,son,has_son,fact,diag)
  end if
  if(rc/=0)call ramses_amd_fatal('force_fine')

  ! Diagnostics (:158-190), reduced on the device: potential energy of the leaf cells, maximum density
  epot_tot=epot_tot+diag(1)
  rho_max(ilevel)=diag(2)

111 format('   Entering force_fine (MI355X) for level ',I2)

end subroutine force_fine_amd

#ifndef WITHOUTMPI
subroutine ramses_amd_force_fine_mpi(ilevel,icount)
  use amr_commons
  use pm_commons
  use poisson_commons
  use constants, only : twopi
  use ramses_amd_iface
  use mpi_mod
  implicit none
  integer,intent(in)::ilevel,icount
  integer::rc,nx_loc,nl,nlc,idim,info
  integer,allocatable,dimension(:)::list,listc
  real(dp)::dx,dx_loc,scale,fact,fourpi,tfrac
  real(kind=8),dimension(2)::diag
  real(kind=8)::epot_loc,epot_all,rho_loc,rho_all,dmax
  real(kind=8),dimension(2)::diag2
  integer(8)::nbad
  character(len=16)::val
  integer::stat
  nx_loc=(icoarse_max-icoarse_min+1)
  ! a regrid or a load balance since the tree went to the device: ramses_amd_tree_epoch has moved
  rc=ramses_amd_poisamr_tree(ramses_amd_tree_epoch,int(ngridmax,8),int(ncoarse,8),son,nbor,father)
  if(rc/=0)call ramses_amd_fatal('force_fine (MPI, tree)')
  dx=0.5D0**ilevel
  scale=boxlen/dble(nx_loc)
  dx_loc=dx*scale
  fourpi=2*twopi
  if(cosmo)fourpi=1.5D0*omega_m*aexp
  fact=-dx_loc**ndim/fourpi/2.0D0
  if(icount/=1.and.icount/=2)then
     write(*,*)'icount has bad value'
     call clean_stop
  end if
  tfrac=0.0d0
  if(dtold(ilevel-1)>0)tfrac=1d0*dtnew(ilevel)/dtold(ilevel-1)*(icount-1)
  call ramses_amd_amr_level_octs(ilevel,nl,list)
  call ramses_amd_amr_level_octs(ilevel-1,nlc,listc)
  diag=0d0
  if(ramses_amd_amrres_active()/=0.and.ramses_amd_f_resident_on())then
     ! the hydro state is resident: f of the own cells goes from the kernel's buffer into the resident acceleration, the
     ! virtual octs follow with ONE exchange of the device arrays (the reference's three, poisson/force_fine.f90:137-139);
     ! the host array f is not written (backup_poisson and load_balance fetch it: ramses_amd_amr_sync_f)
     call ramses_amd_amr_ensure()      ! (a level the host rebuilt since the last device routine goes up first, its stale f with it)
     rc=ramses_amd_poisamr_force_mpi_resident(ilevel,active(ilevel)%ngrid,nl,list,nlc,listc,phi,phi_old,rho,tfrac,1,fact,diag)
     if(rc/=0)call ramses_amd_fatal('force_fine (MPI, AMR level, resident f)')
     call ramses_amd_amr_halo(ilevel,7)
     ramses_amd_f_on_device(ilevel)=.true.
     call get_environment_variable('RAMSES_AMD_F_CHECK',val,status=stat)
     if(stat==0.and.trim(val)=='1')then
        ! diagnostic: the path through the host array beside it, cell by cell (own and virtual octs)
        rc=ramses_amd_poisamr_force_mpi(ilevel,active(ilevel)%ngrid,nl,list,nlc,listc,phi,phi_old,rho,f,tfrac,1,fact,diag2)
        if(rc/=0)call ramses_amd_fatal('force_fine (MPI, AMR level, check)')
        do idim=1,ndim
           call make_virtual_fine_dp(f(1,idim),ilevel)
        end do
        rc=ramses_amd_amrres_compare_f(nl,list,f,dmax,nbad)
        if(rc/=0)call ramses_amd_fatal('force_fine (MPI, AMR level, compare)')
        write(*,'(A,I3,A,I3,A,I10,A,ES12.4,A,I8,A,I8)')' ramses_amd: f check, level ',ilevel,' rank ',myid,': ',nbad, &
             & ' cells differ, max ',dmax,' own octs ',active(ilevel)%ngrid,' all ',nl
     end if
     deallocate(list,listc)
  else
     rc=ramses_amd_poisamr_force_mpi(ilevel,active(ilevel)%ngrid,nl,list,nlc,listc,phi,phi_old,rho,f,tfrac,1,fact,diag)
     if(rc/=0)call ramses_amd_fatal('force_fine (MPI, AMR level)')
     deallocate(list,listc)
     do idim=1,ndim
        call make_virtual_fine_dp(f(1,idim),ilevel)
     end do
  end if
  epot_loc=diag(1); rho_loc=diag(2)
  call MPI_ALLREDUCE(epot_loc,epot_all,1,MPI_DOUBLE_PRECISION,MPI_SUM,MPI_COMM_WORLD,info)
  call MPI_ALLREDUCE(rho_loc ,rho_all ,1,MPI_DOUBLE_PRECISION,MPI_MAX,MPI_COMM_WORLD,info)
  epot_tot=epot_tot+epot_all
  rho_max(ilevel)=rho_all
end subroutine ramses_amd_force_fine_mpi
#endif

subroutine force_fine(ilevel,icount)
  use amr_commons, only: numbtot
  use ramses_amd_iface
  implicit none
  integer::ilevel,icount
  integer(8)::t0
  call ramses_amd_tic(t0)
  if(ilevel>=1.and.ilevel<=64)ramses_amd_f_on_device(ilevel)=.false.
  call force_fine_amd(ilevel,icount)
  ! AMR run with the hydro state on the device: its copy of the acceleration follows (unless force_fine has just left f
  ! there and nowhere else: ramses_amd_force_fine_mpi)
  if(ramses_amd_amr_resident())then
     if(ramses_amd_amrres_active()/=0.and.numbtot(1,ilevel)>0)then
        if(.not.ramses_amd_f_on_device(min(max(ilevel,1),64)))call ramses_amd_amr_load_f(ilevel)
     end if
  end if
  call ramses_amd_toc('force_fine',ilevel,t0)
end subroutine force_fine
