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

subroutine rho_fine_amd(ilevel,icount)
  use amr_commons
  use hydro_commons
  use poisson_commons
  use ramses_amd_iface
  implicit none
  integer::ilevel,icount
  !--------------------------------------------------------------------------
  ! Same contract as the reference (pm/rho_fine.f90:5-226).  On the device-resident level of a
  ! hydro + self-gravity run without particles the source of the Poisson equation is the hydro
  ! deposit alone: it is computed on the GPU from the resident density (every cell's mass CIC-
  ! deposited at its centre of mass, contributions added in the reference's order; multipole(1:4)
  ! by strictly sequential sums) and stays there for multigrid_fine / force_fine.  The other duties
  ! of rho_fine in that configuration are no-ops (no particles, no reception / boundary cells,
  ! m_refine < 0, cic_levelmax = 0); its reset of phi is subsumed by multigrid_fine's first guess.
  ! RAMSES_AMD_RESIDENT_RHO=0 keeps the reference's host loops (the density is synced back first).
  !--------------------------------------------------------------------------
  integer::rc,nx_loc,stat
  real(dp)::scale
  real(kind=8),dimension(4)::mp4
  type(ramses_amd_hydro_params)::p
  character(len=16)::val
  logical,save::first=.true.,rho_dev=.true.
  ramses_amd_pois_dev=.false.
  ramses_amd_pois_amr_level=0
  if(poisson.and.hydro)then
     if(ramses_amd_resident())then
        if(first)then
           call get_environment_variable('RAMSES_AMD_RESIDENT_RHO',val,status=stat)
           if(stat==0)then
              if(trim(val)=='0')rho_dev=.false.
           end if
           first=.false.
        end if
        if(rho_dev.and.numbtot(1,ilevel)>0.and.ilevel==levelmin.and.m_refine(ilevel)<0.0d0.and.cic_levelmax==0 &
             & .and.nboundary==0)then
           if(verbose)write(*,111)ilevel
           call ramses_amd_fill_hydro_params(p)
           nx_loc=icoarse_max-icoarse_min+1
           scale=boxlen/dble(nx_loc)
           rc=ramses_amd_resident_rho_fine_f90(p,ilevel,active(ilevel)%ngrid,ramses_amd_octs(ilevel),xg, &
                & int(ngridmax,8),int(ncoarse,8),nx_loc,uold,boxlen,nvector,mp4)
           if(rc/=0)call ramses_amd_fatal('rho_fine')
           multipole(1:ndim+1)=mp4(1:ndim+1)
           rho_tot=multipole(1)/scale**ndim
           ramses_amd_pois_dev=.true.
           return
        end if
        rc=ramses_amd_resident_sync_density_f90(uold)
        if(rc/=0)call ramses_amd_fatal('rho_fine (density of the resident level)')
     end if
  end if
  call rho_fine_reference(ilevel,icount)
111 format('   Entering rho_fine (MI355X) for level ',I2)
end subroutine rho_fine_amd

!------------------------------------------------------------------------------
! rho_fine(ilevel,icount) of an AMR run whose hydro state lives on the device (single rank, periodic box, no
! particles): the deposit loop of the reference (pm/rho_fine.f90:45-60: multipole_fine(l) and cic_from_multipole(l)
! for l = nlevelmax .. ilevel) runs on the GPU on the resident density -- multipoles of leaf and split cells, the
! order-tagged CIC gather through the tree, the four sequential multipole sums at levelmin (csrc/rho_fine.hip) -- and
! only rho of the visited levels comes back.  What else the reference's routine does in this configuration is the
! reset of phi on the level and rho_tot (:66-70,176-183); everything tied to particles, cic_levelmax, m_refine,
! physical boundaries or several ranks keeps the reference's routine (ramses_amd_rho_amr_device says no).
!------------------------------------------------------------------------------
logical function ramses_amd_rho_amr_device(ilevel)
  use amr_commons
  use pm_commons
  use hydro_commons
  use poisson_commons
  use ramses_amd_iface
  implicit none
  integer,intent(in)::ilevel
  character(len=16)::val
  integer::stat,l
  logical,save::first=.true.,enabled=.true.,enabled_mpi=.true.
  if(first)then
     call get_environment_variable('RAMSES_AMD_RESIDENT_RHO',val,status=stat)
     if(stat==0)then
        if(trim(val)=='0')enabled=.false.
     end if
     call get_environment_variable('RAMSES_AMD_RESIDENT_RHO_MPI',val,status=stat)
     if(stat==0)then
        if(trim(val)=='0')enabled_mpi=.false.
     end if
     first=.false.
  end if
  ramses_amd_rho_amr_device=.false.
  if(.not.enabled)return
#ifdef TSC
  return
#endif
  if(.not.(poisson.and.hydro))return
  if(ramses_amd_amrres_active()==0)return
  if(pic.or.nboundary>0.or.cic_levelmax>0.or.ilevel<2.or.ndim/=3)return
#ifdef WITHOUTMPI
  if(ncpu>1)return
#endif
  ! several ranks: the deposit and its three exchanges per level run on every rank's GPU (ramses_amd_rho_fine_mpi below);
  ! RAMSES_AMD_RESIDENT_RHO_MPI=0 keeps the reference's host routine there (the density comes back first)
  if(ncpu>1.and..not.enabled_mpi)return
  if(icoarse_max/=icoarse_min)return
  do l=ilevel,nlevelmax
     if(m_refine(l)>-1.0d0)return
  end do
  ramses_amd_rho_amr_device=.true.
end function ramses_amd_rho_amr_device

subroutine rho_fine(ilevel,icount)
  use amr_commons
  use hydro_commons, only: uold, smallr
  use poisson_commons
  use ramses_amd_iface
  implicit none
  integer::ilevel,icount,l,rc,n,k,ind,i
  integer(8)::t0
  integer,allocatable,dimension(:)::first,lists
  real(kind=8),dimension(4)::mp4
  type(ramses_amd_hydro_params)::p
  logical,external::ramses_amd_rho_amr_device
  integer,save::xg_epoch=-1
  call ramses_amd_tic(t0)
  if(ramses_amd_amr_resident().and.poisson.and.numbtot(1,ilevel)>0)then
     if(ramses_amd_rho_amr_device(ilevel))then
        call ramses_amd_amr_ensure()
        if(xg_epoch/=ramses_amd_tree_epoch)then
           rc=ramses_amd_amrres_xg(xg)
           if(rc/=0)call ramses_amd_fatal('rho_fine (oct centres)')
           xg_epoch=ramses_amd_tree_epoch
        end if
        if(ilevel==levelmin)multipole=0d0
        if(ncpu>1)then
           call ramses_amd_rho_fine_mpi(ilevel,icount)
           call ramses_amd_toc('rho_fine (device, MPI)',ilevel,t0)
           call ramses_amd_toc('rho_fine',ilevel,t0)
           return
        end if
        if(ilevel==levelmin.or.icount>1)then
           allocate(first(0:nlevelmax-ilevel+1))
           first(0)=0
           do l=ilevel,nlevelmax
              n=0
              if(numbtot(1,l)>0)n=active(l)%ngrid
              first(l-ilevel+1)=first(l-ilevel)+n
           end do
           allocate(lists(1:max(first(nlevelmax-ilevel+1),1)))
           do l=ilevel,nlevelmax
              k=first(l-ilevel)
              n=first(l-ilevel+1)-k
              if(n>0)lists(k+1:k+n)=active(l)%igrid(1:n)
           end do
           call ramses_amd_fill_hydro_params(p)
           mp4=0d0
           rc=ramses_amd_amrres_rho_fine(p,ilevel,nlevelmax,levelmin,nvector,first,lists,boxlen,rho,mp4)
           if(rc/=0)call ramses_amd_fatal('rho_fine (AMR deposit)')
           if(ilevel==levelmin)multipole(1:ndim+1)=mp4(1:ndim+1)
           deallocate(first,lists)
        end if
        ! the level's first guess starts from zero (:66-70); rho_tot (:176-183)
        do ind=1,twotondim
           k=ncoarse+(ind-1)*ngridmax
           do i=1,active(ilevel)%ngrid
              phi(k+active(ilevel)%igrid(i))=0.0d0
           end do
        end do
        rho_tot=multipole(1)/boxlen**ndim
        ramses_amd_pois_dev=.false.
        ramses_amd_pois_amr_level=0
        call ramses_amd_toc('rho_fine',ilevel,t0)
        return
     end if
  end if
  ! AMR run with the hydro state on the device: multipole_fine (pm/rho_fine.f90:666-770) reads the density of the
  ! levels it visits (:45-47) from the host array -- bring it back (nothing else of uold is read)
  if(ramses_amd_amr_resident())then
     if(ramses_amd_amrres_active()/=0.and.(ilevel==levelmin.or.icount>1))then
        do l=nlevelmax,ilevel,-1
           if(numbtot(1,l)>0.and.l<ramses_amd_amr_host_from)then
              rc=ramses_amd_amrres_sync_density(active(l)%ngrid,ramses_amd_octs(l),uold)
              if(rc/=0)call ramses_amd_fatal('rho_fine (density back to the host)')
           end if
        end do
     end if
  end if
  call rho_fine_amd(ilevel,icount)
  call ramses_amd_toc('rho_fine',ilevel,t0)
end subroutine rho_fine

!------------------------------------------------------------------------------
! The same with several ranks (one per GPU; hydro state, tree and communicators of the AMR levels resident,
! ramses_amd_iface: AMR residency under MPI).  The reference's loop (pm/rho_fine.f90:45-60), level by level from
! nlevelmax down: multipole_fine(l) on the rank's own octs, the exchange of the four multipoles (:814-817; a split
! cell's son oct may belong to another rank), cic_from_multipole(l) into the own AND the reception cells (cic_cell loops
! over the own octs only, :858-866), make_virtual_reverse_dp(rho,l) added peer by peer, make_virtual_fine_dp(rho,l) --
! all five on the device vectors; rho of the level's cells then goes to the host vector, where multigrid_fine /
! phi_fine_cg / force_fine of the MPI path read it.  What rho_fine does after the loop (:66-183): phi = 0 on the
! level's own and reception cells, the reset of rho in the virtual boundaries followed by the two exchanges that
! restore it (no particles: + 0 on the owners' side, then the same values back), the MPI_ALLREDUCE of the multipole.
!------------------------------------------------------------------------------
subroutine ramses_amd_rho_fine_mpi(ilevel,icount)
  use amr_commons
  use hydro_commons, only: uold
  use poisson_commons
  use ramses_amd_iface
  use mpi_mod
  implicit none
  integer,intent(in)::ilevel,icount
#ifndef WITHOUTMPI
  integer::l,rc,nl,ind,i,k,icpu,info
  integer,allocatable,dimension(:)::list
  real(kind=8),dimension(4)::mp4
  real(kind=8),dimension(1:ndim+1)::multipole_in,multipole_out
  type(ramses_amd_hydro_params)::p
  if(ilevel==levelmin.or.icount>1)then
     call ramses_amd_fill_hydro_params(p)
     ! (the steady state of a one-level run under the distributed dense multigrid: the deposit stays on the device, where the
     !  solve and force_fine read it -- ramses_amd_iface: ramses_amd_pois_mpi_dev)
     k=0
     if(ramses_amd_pois_mpi_dev)k=1
     rc=ramses_amd_amrres_rho_keep(k)
     do l=nlevelmax,ilevel,-1
        if(numbtot(1,l)==0)cycle
        call ramses_amd_amr_level_octs(l,nl,list)
        rc=ramses_amd_amrres_rho_mpi_multipole(p,l,active(l)%ngrid,nl,list,boxlen)
        if(rc/=0)call ramses_amd_fatal('rho_fine (multipole_fine under MPI)')
        call ramses_amd_amr_halo(l,6)
        rc=ramses_amd_amrres_rho_mpi_deposit(l,nvector,boxlen)
        if(rc/=0)call ramses_amd_fatal('rho_fine (cic_from_multipole under MPI)')
        call ramses_amd_amr_halo(l,4)
        call ramses_amd_amr_halo(l,5)
        mp4=0d0
        rc=ramses_amd_amrres_rho_mpi_finish(l,levelmin,nvector,list,rho,mp4)
        if(rc/=0)call ramses_amd_fatal('rho_fine (deposit back to the host vector)')
        if(l==levelmin)multipole(1:ndim+1)=mp4(1:ndim+1)
        deallocate(list)
     end do
  end if
  do ind=1,twotondim
     k=ncoarse+(ind-1)*ngridmax
     do i=1,active(ilevel)%ngrid
        phi(k+active(ilevel)%igrid(i))=0.0d0
     end do
     do icpu=1,ncpu
        do i=1,reception(icpu,ilevel)%ngrid
           phi(k+reception(icpu,ilevel)%igrid(i))=0.0d0
        end do
     end do
  end do
  if(ilevel==levelmin)then
     multipole_in=multipole(1:ndim+1)
     call MPI_ALLREDUCE(multipole_in,multipole_out,ndim+1,MPI_DOUBLE_PRECISION,MPI_SUM,MPI_COMM_WORLD,info)
     multipole(1:ndim+1)=multipole_out
  endif
  rho_tot=multipole(1)/boxlen**ndim
  ramses_amd_pois_dev=.false.
  ramses_amd_pois_amr_level=0
#endif
end subroutine ramses_amd_rho_fine_mpi
