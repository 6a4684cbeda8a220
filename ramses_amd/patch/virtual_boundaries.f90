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
  use ramses_amd_iface
  implicit none
  integer::ilevel
  real(dp),dimension(1:ncoarse+ngridmax*twotondim)::xx
#ifndef WITHOUTMPI
  integer::k
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
#endif
  call make_virtual_fine_dp_reference(xx,ilevel)
end subroutine make_virtual_fine_dp

subroutine make_virtual_reverse_dp(xx,ilevel)
  use amr_commons
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
