!==============================================================================
! courant_fine.f90 of the ramses_amd patch directory.
!
! Shadows hydro/courant_fine.f90.  The untouched reference routine is pulled
! in under the name courant_fine_reference; the new courant_fine(ilevel) keeps
! the reference's name, argument and meaning.  When the level is
! device-resident the CFL reduction (cmpdt, hydro/godunov_utils.f90:5-120) and
! the mass/energy sums run on the MI355X over the resident brick; otherwise
! the reference routine runs on the host arrays.
!==============================================================================
#define courant_fine courant_fine_reference
#include "hydro/courant_fine.f90"
#undef courant_fine

subroutine courant_fine(ilevel)
  use amr_commons
  use hydro_commons
  use poisson_commons
  use mpi_mod
  use ramses_amd_iface
  implicit none
  integer::ilevel
  type(ramses_amd_hydro_params)::p
  integer::rc,nx_loc
  real(dp)::scale,dx
  real(kind=8),dimension(4)::out4
#ifndef WITHOUTMPI
  integer::info
  real(kind=8),dimension(3)::comm_buffin,comm_buffout
  real(kind=8)::dt_all
#endif

  if(numbtot(1,ilevel)==0)return
#ifndef WITHOUTMPI
  if(ramses_amd_mpi_resident())then
     ! one brick per rank on its GPU: local reduction on the device, then the reference's own
     ! reductions over the ranks (hydro/courant_fine.f90:133-140)
     if(verbose)write(*,111)ilevel
     call ramses_amd_mpires_ensure()
     call ramses_amd_fill_hydro_params(p)
     nx_loc=icoarse_max-icoarse_min+1
     scale=boxlen/dble(nx_loc)
     dx=0.5D0**ilevel*scale
     rc=ramses_amd_mpires_courant(p,dx,dtnew(ilevel),out4)
     if(rc/=0)call ramses_amd_fatal('courant_fine')
     comm_buffin(1:3)=out4(2:4)
     call MPI_ALLREDUCE(comm_buffin,comm_buffout,3,MPI_DOUBLE_PRECISION,MPI_SUM,MPI_COMM_WORLD,info)
     call MPI_ALLREDUCE(out4(1),dt_all,1,MPI_DOUBLE_PRECISION,MPI_MIN,MPI_COMM_WORLD,info)
     mass_tot=mass_tot+comm_buffout(1)
     ekin_tot=ekin_tot+comm_buffout(2)
     eint_tot=eint_tot+comm_buffout(3)
     dtnew(ilevel)=MIN(dtnew(ilevel),dt_all)
     return
  end if
#endif
  if(ramses_amd_amr_config())then
     ! AMR run: the first courant_fine of the time loop arms the residency (the mesh construction before it ran
     ! on the host arrays); from here on uold lives on the device
     ramses_amd_amr_armed=.true.
     if(verbose)write(*,111)ilevel
     call ramses_amd_amr_ensure()
     call ramses_amd_fill_hydro_params(p)
     nx_loc=icoarse_max-icoarse_min+1
     scale=boxlen/dble(nx_loc)
     dx=0.5D0**ilevel*scale
     rc=ramses_amd_amrres_courant(p,active(ilevel)%ngrid,ramses_amd_octs(ilevel),dx,dtnew(ilevel),out4)
     if(rc/=0)call ramses_amd_fatal('courant_fine')
#ifndef WITHOUTMPI
     if(ncpu>1)then
        ! several ranks: the reference's own reductions (hydro/courant_fine.f90:133-140); the first call also
        ! chooses the transport of the virtual-boundary exchanges (every rank is here together)
        if(.not.ramses_amd_amr_halo_ready)then
           call ramses_amd_halo_init()
           ramses_amd_amr_halo_ready=.true.
        end if
        comm_buffin(1:3)=out4(2:4)
        call MPI_ALLREDUCE(comm_buffin,comm_buffout,3,MPI_DOUBLE_PRECISION,MPI_SUM,MPI_COMM_WORLD,info)
        call MPI_ALLREDUCE(out4(1),dt_all,1,MPI_DOUBLE_PRECISION,MPI_MIN,MPI_COMM_WORLD,info)
        out4(2:4)=comm_buffout(1:3)
        out4(1)=dt_all
     end if
#endif
     mass_tot=mass_tot+out4(2)
     ekin_tot=ekin_tot+out4(3)
     eint_tot=eint_tot+out4(4)
     dtnew(ilevel)=MIN(dtnew(ilevel),out4(1))
     return
  end if
  if(.not.ramses_amd_resident())then
     call courant_fine_reference(ilevel)
     return
  end if
  if(verbose)write(*,111)ilevel

  call ramses_amd_fill_hydro_params(p)
  nx_loc=icoarse_max-icoarse_min+1
  scale=boxlen/dble(nx_loc)
  dx=0.5D0**ilevel*scale

  if(poisson)then
     ! cmpdt with the gravity term (hydro/courant_fine.f90:77-85)
     rc=ramses_amd_resident_courant_grav_f90(p,ilevel,active(ilevel)%ngrid,ramses_amd_octs(ilevel),xg, &
          & int(ngridmax,8),int(ncoarse,8),nx_loc,uold,f,dx,dtnew(ilevel),out4)
  else
     rc=ramses_amd_resident_courant_f90(p,ilevel,active(ilevel)%ngrid,ramses_amd_octs(ilevel),xg, &
          & int(ngridmax,8),int(ncoarse,8),nx_loc,uold,dx,dtnew(ilevel),out4)
  end if
  if(rc/=0)call ramses_amd_fatal('courant_fine')

  ! same bookkeeping as hydro/courant_fine.f90:150-156 (single rank)
  mass_tot=mass_tot+out4(2)
  ekin_tot=ekin_tot+out4(3)
  eint_tot=eint_tot+out4(4)
  dtnew(ilevel)=MIN(dtnew(ilevel),out4(1))

111 format('   Entering courant_fine (MI355X) for level ',I2)

end subroutine courant_fine
