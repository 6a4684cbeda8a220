!==============================================================================
! virtual_boundaries.f90 of the ramses_amd patch directory.
!
! Shadows amr/virtual_boundaries.f90 (bin/Makefile:153 VPATH).  The untouched
! reference file is pulled in by the preprocessor with make_virtual_fine_dp,
! make_virtual_reverse_dp and build_comm renamed to *_reference (nothing is
! copied; authorize_*, make_virtual_*_int, make_virtual_coarse_int stay the
! reference's).  The routines of the same name below keep the reference's
! names, arguments and meaning.
!
! While a level is device-resident under MPI (ramses_amd_iface:
! ramses_amd_mpi_resident) the two halo calls of amr_step on the hydro state,
!     make_virtual_reverse_dp(unew(1,ivar),ilevel)   amr/amr_step.f90:388-399
!     make_virtual_fine_dp   (uold(1,ivar),ilevel)   amr/amr_step.f90:497-508
! run on the GPU: the anonymous array xx is recognised by its address (a column
! of uold or unew), the call for ivar=1 moves ALL nvar fields in one exchange
! (pack on the device, RCCL neighbour send/recv or the host-MPI fallback,
! unpack), the calls for ivar>1 have nothing left to do.  Every other array
! (flag1, cpu_map, phi, rho, f, ... and uold/unew of other levels) takes the
! reference's host MPI path.
!
! AMR runs with several ranks whose hydro state is device-resident
! (ramses_amd_iface: ramses_amd_amr_resident): the same two calls, on ANY level,
! and the forward call after a regrid (amr/amr_step.f90:49-62), run on the
! reference's own cell vectors on the device, addressed by the reference's own
! communicators (emission / reception oct lists; ramses_amd_amr_halo).  The
! reverse exchange accumulates peer by peer in icpu order like the reference
! (:857-867).  A level the host has just rebuilt (refine_fine; it is re-sent to
! the device before the next device routine) takes the reference's path; a level
! current on both sides takes both, so that both stay current.
!
! While a multigrid solve of a level runs on the device with several ranks
! (poisson/multigrid_fine_commons.f90:194-257), make_virtual_fine_dp on phi and on the
! residual f(:,1) of that level exchanges the device arrays (ramses_amd_mg_halo).
!==============================================================================
#define make_virtual_fine_dp make_virtual_fine_dp_reference
#define make_virtual_reverse_dp make_virtual_reverse_dp_reference
#define build_comm build_comm_reference
#include "amr/virtual_boundaries.f90"
#undef make_virtual_fine_dp
#undef make_virtual_reverse_dp
#undef build_comm

subroutine make_virtual_fine_dp(xx,ilevel)
  use amr_commons
  use hydro_commons
  use poisson_commons, only: phi, f
  use ramses_amd_iface
  implicit none
  integer::ilevel
  real(dp),dimension(1:ncoarse+ngridmax*twotondim)::xx
#ifndef WITHOUTMPI
  integer::k
  ! the level a device multigrid solve is running on (several ranks, levels resident): phi and the residual f(:,1) live on
  ! the device between the routines of the solve (poisson/multigrid_fine_commons.f90:194-257); their virtual boundaries are
  ! exchanged from there
  if(ramses_amd_mg_active.and.ramses_amd_mg_started.and.ramses_amd_mg_mpi_resident.and.ncpu>1)then
     if(ilevel==ramses_amd_mg_level)then
        if(ramses_amd_which_column(xx,phi,int(ncoarse,8)+int(twotondim,8)*int(ngridmax,8),1)==1)then
           call ramses_amd_mg_halo(ilevel,1,0)
           return
        end if
        k=ramses_amd_which_column(xx,f,int(ncoarse,8)+int(twotondim,8)*int(ngridmax,8),3)
        if(k==1)then
           call ramses_amd_mg_halo(ilevel,3,0)
           return
        end if
        if(k/=0)then
           write(*,*)'ramses_amd: make_virtual_fine_dp on f(:,',k,') while the level is solved on the device'
           call ramses_amd_fatal('make_virtual_fine_dp (multigrid level)')
        end if
     end if
  end if
  if(ramses_amd_mpi_on)then
     if(ramses_amd_mpires_active()/=0)then
        k=ramses_amd_mpires_which(xx)
        if(k/=0.and.ilevel==levelmin)then
           if(k<0)then
              write(*,*)'ramses_amd: make_virtual_fine_dp on unew of the device-resident level'
              call ramses_amd_fatal('make_virtual_fine_dp (unew)')
           end if
           if(k==1)call ramses_amd_halo_forward()
           return
        end if
     end if
  end if
  if(ncpu>1.and.numbtot(1,ilevel)>0)then
     if(ramses_amd_amr_resident())then
        if(ramses_amd_amrres_active()/=0)then
           ! (a column of the HOST unew is rho_fine's multipole scratch, pm/rho_fine.f90:815: the reference's path)
           k=ramses_amd_which_column(xx,uold,int(ncoarse,8)+int(twotondim,8)*int(ngridmax,8),nvar)
           if(k/=0.and.ilevel<ramses_amd_amr_reload_from)then
              ! the device holds the level: all nvar variables in one exchange, with the call for ivar=1
              if(k==1)call ramses_amd_amr_halo(ilevel,0)
              ! the host copy is current as well (synced for refine_fine): keep it so with the reference's exchange
              if(ilevel<ramses_amd_amr_host_from)return
           end if
        end if
     end if
  end if
#endif
  call make_virtual_fine_dp_reference(xx,ilevel)
end subroutine make_virtual_fine_dp

subroutine make_virtual_reverse_dp(xx,ilevel)
  use amr_commons
  use hydro_commons
  use ramses_amd_iface
  implicit none
  integer::ilevel
  real(dp),dimension(1:ncoarse+ngridmax*twotondim)::xx
#ifndef WITHOUTMPI
  integer::k,rc
  if(ramses_amd_mpi_on)then
     if(ramses_amd_mpires_active()/=0)then
        k=ramses_amd_mpires_which(xx)
        if(k/=0.and.ilevel==levelmin)then
           if(k>0)then
              write(*,*)'ramses_amd: make_virtual_reverse_dp on uold of the device-resident level'
              call ramses_amd_fatal('make_virtual_reverse_dp (uold)')
           end if
           if(k==-1)then
              ! the reception cells of unew are zero on a fully refined level: + 0.0 on the emission cells
              rc=ramses_amd_mpires_reverse_unew()
              if(rc/=0)call ramses_amd_fatal('make_virtual_reverse_dp')
           end if
           return
        end if
     end if
  end if
  if(ncpu>1.and.numbtot(1,ilevel)>0)then
     if(ramses_amd_amr_resident())then
        if(ramses_amd_amrres_active()/=0)then
           if(ramses_amd_which_column(xx,uold,int(ncoarse,8)+int(twotondim,8)*int(ngridmax,8),nvar)/=0)then
              write(*,*)'ramses_amd: make_virtual_reverse_dp on uold of a device-resident AMR level'
              call ramses_amd_fatal('make_virtual_reverse_dp (uold)')
           end if
           k=ramses_amd_which_column(xx,unew,int(ncoarse,8)+int(twotondim,8)*int(ngridmax,8),nvar)
           if(k/=0)then
              ! unew exists on the device only (set_unew and the sweep ran there): the corrections collected in the
              ! virtual octs go home, all nvar variables with the call for ivar=1
              if(ilevel>=ramses_amd_amr_reload_from)call ramses_amd_fatal('make_virtual_reverse_dp (level awaiting its reload)')
              if(k==1)call ramses_amd_amr_halo(ilevel,1)
              return
           end if
           if(pressure_fix)then
              ! enew / divu are device vectors there (scratch of one hydro step, amr/amr_step.f90:417-418)
              if(ramses_amd_which_column(xx,enew,int(ncoarse,8)+int(twotondim,8)*int(ngridmax,8),1)==1)then
                 call ramses_amd_amr_halo(ilevel,2)
                 return
              end if
              if(ramses_amd_which_column(xx,divu,int(ncoarse,8)+int(twotondim,8)*int(ngridmax,8),1)==1)then
                 call ramses_amd_amr_halo(ilevel,3)
                 return
              end if
           end if
        end if
     end if
  end if
#endif
  call make_virtual_reverse_dp_reference(xx,ilevel)
end subroutine make_virtual_reverse_dp

subroutine build_comm(ilevel)
  use amr_commons
  use hydro_commons
  use ramses_amd_iface
  implicit none
  integer::ilevel
  !--------------------------------------------------------------------------
  ! The communicators of a level are about to be rebuilt (after refine_fine /
  ! load_balance): the device image of the resident level is dropped first (the
  ! host array is brought up to date if the device holds the only current copy)
  ! and rebuilt from the new lists the next time the level is stepped.
  !--------------------------------------------------------------------------
#ifndef WITHOUTMPI
  integer::rc
  ! AMR residency: the device copy of this level's communicators is stale from here on
  if(ilevel>=1.and.ilevel<=64)ramses_amd_comm_epoch(ilevel)=ramses_amd_comm_epoch(ilevel)+1
  if(ramses_amd_mpi_on.and.ilevel==levelmin)then
     if(ramses_amd_mpires_active()/=0)then
        rc=ramses_amd_mpires_sync_host(uold)
        if(rc==0)rc=ramses_amd_mpires_invalidate()
        if(rc/=0)call ramses_amd_fatal('build_comm (device-resident level)')
     end if
  end if
#endif
  call build_comm_reference(ilevel)
end subroutine build_comm
