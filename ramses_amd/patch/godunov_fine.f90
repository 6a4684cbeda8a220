!==============================================================================
! godunov_fine.f90 of the ramses_amd patch directory (make PATCH=.../ramses_amd/patch).
!
! RAMSES shadows whole files (bin/Makefile:153 VPATH), so this file must define
! every external symbol of hydro/godunov_fine.f90: godunov_fine, set_unew,
! set_uold, add_gravity_source_terms, add_pdv_source_terms, godfine1.  The
! untouched reference file is pulled in by the preprocessor with its
! godunov_fine, set_unew and set_uold renamed to *_reference (no reference
! source is copied); the new godunov_fine below keeps the reference's name, argument and
! meaning and hands the level to the MI355X sweep through the C ABI.
!
! Needs -I<ramses root> on the compile line (the patch Makefile adds -I..).
!==============================================================================
#define godunov_fine godunov_fine_reference
#define set_unew set_unew_reference
#define set_uold set_uold_reference
#include "hydro/godunov_fine.f90"
#undef godunov_fine
#undef set_unew
#undef set_uold

!------------------------------------------------------------------------------
! set_unew / set_uold (hydro/godunov_fine.f90:40-130,135-232).  When the level
! is device-resident (ramses_amd_iface: ramses_amd_resident) the sweep kernel
! writes uold + flux differences into a second brick, so set_unew has nothing
! to do and set_uold is a buffer swap; otherwise the reference routines run.
!------------------------------------------------------------------------------
subroutine set_unew(ilevel)
  use amr_commons
  use ramses_amd_iface
  implicit none
  integer::ilevel
  integer::rc
  type(ramses_amd_hydro_params)::p
  if(numbtot(1,ilevel)==0)return
  if(ramses_amd_resident())return
  if(ramses_amd_mpi_resident())return
  if(ramses_amd_amr_resident())then
     call ramses_amd_amr_ensure()
     if(pressure_fix)then
        call ramses_amd_fill_hydro_params(p)
        rc=ramses_amd_amrres_set_unew_pfix(p,active(ilevel)%ngrid,ramses_amd_octs(ilevel))
     else
        rc=ramses_amd_amrres_set_unew(active(ilevel)%ngrid,ramses_amd_octs(ilevel))
     end if
     if(rc/=0)call ramses_amd_fatal('set_unew')
#ifndef WITHOUTMPI
     if(ncpu>1)then
        ! unew = 0 on the virtual octs (hydro/godunov_fine.f90:92-122): they collect the corrections the finer
        ! level owes to cells of other ranks, which make_virtual_reverse_dp then sends home
        call ramses_amd_amr_comm_ensure(ilevel)
        rc=ramses_amd_amrres_zero_unew_virtual(ilevel)
        if(rc/=0)call ramses_amd_fatal('set_unew (virtual octs)')
     end if
#endif
     return
  end if
  call set_unew_reference(ilevel)
end subroutine set_unew

subroutine set_uold(ilevel)
  use amr_commons
  use ramses_amd_iface
  implicit none
  integer::ilevel
  integer::rc
  type(ramses_amd_hydro_params)::p
  if(numbtot(1,ilevel)==0)return
  if(ramses_amd_resident())then
     if(poisson)then
        ! add_gravity_source_terms (:160-162,237-289) on the new state, then the swap
        call ramses_amd_fill_hydro_params(p)
        rc=ramses_amd_resident_set_uold_grav_f90(p,ilevel,dtnew(ilevel))
     else
        rc=ramses_amd_resident_set_uold_f90(ilevel)
     end if
     if(rc/=0)call ramses_amd_fatal('set_uold')
     return
  end if
  if(ramses_amd_mpi_resident())then
     rc=ramses_amd_mpires_set_uold()
     if(rc/=0)call ramses_amd_fatal('set_uold')
     return
  end if
  if(ramses_amd_amr_resident())then
     call ramses_amd_amr_ensure()
     call ramses_amd_fill_hydro_params(p)
     if(pressure_fix)then
        ! (+ add_gravity_source_terms with poisson), add_pdv_source_terms, uold = unew, the energy switch
        rc=ramses_amd_amrres_set_uold_pfix(p,active(ilevel)%ngrid,ramses_amd_octs(ilevel),dtnew(ilevel), &
             & 0.5d0**ilevel*boxlen/dble(icoarse_max-icoarse_min+1),beta_fix,hexp)
     else if(poisson)then
        rc=ramses_amd_amrres_set_uold_grav(p,active(ilevel)%ngrid,ramses_amd_octs(ilevel),dtnew(ilevel))
     else
        rc=ramses_amd_amrres_set_uold(p,active(ilevel)%ngrid,ramses_amd_octs(ilevel))
     end if
     if(rc/=0)call ramses_amd_fatal('set_uold')
     return
  end if
  call set_uold_reference(ilevel)
end subroutine set_uold

subroutine godunov_fine(ilevel)
  use amr_commons
  use hydro_commons
  use poisson_commons
  use ramses_amd_iface
  implicit none
  integer::ilevel
  !--------------------------------------------------------------------------
  ! Same contract as the reference (hydro/godunov_fine.f90:5-35): on entry
  ! hydro variables are in uold and unew was prepared by set_unew; on exit
  ! unew of the level's active cells has been updated by the second-order
  ! Godunov fluxes.  The sweep itself runs on the GPU.
  !--------------------------------------------------------------------------
  type(ramses_amd_hydro_params)::p
  integer::rc,nx_loc,has_f
  real(dp)::scale,dx
  logical::amr_level

  if(numbtot(1,ilevel)==0)return
  if(static)return

  if(.not.ramses_amd_enabled())then
     call godunov_fine_reference(ilevel)
     return
  end if
  if(verbose)write(*,111)ilevel

  ! What the device path does not implement stops the run (no silent fallback)
  if(MC_tracer.or.momentum_feedback>0.or.strict_equilibrium>0)then
     write(*,*)'ramses_amd: MC_tracer/momentum_feedback/strict_equilibrium are not on the device'
     call ramses_amd_fatal('godunov_fine (unsupported option)')
  end if

  call ramses_amd_fill_hydro_params(p)
  nx_loc=icoarse_max-icoarse_min+1
  scale=boxlen/dble(nx_loc)
  dx=0.5d0**ilevel*scale

#if NDIM<3
  ! NDIM = 1, 2 (BASELINE config C1: sedov1d.nml on one uniform level): a fully refined level without finer octs goes to
  ! the dense sweep, embedded in a 3-D brick with the boundary octs as its ghost cells (csrc/capi_host.hip
  ! ramses_amd_godunov_fine_lowdim_f90); AMR levels, several ranks, self-gravity, difmag, pressure_fix and passive scalars
  ! of such builds stay the reference's routine (said once).
  call ramses_amd_godunov_lowdim(ilevel,p,dx)
  return
#else

#ifndef WITHOUTMPI
  ! MPI, one rank per GPU: the dense sweep on the rank's resident brick (ghost layer kept current
  ! by the device halo exchange, virtual_boundaries.f90 of this directory)
  if(ramses_amd_mpi_resident())then
     call ramses_amd_mpires_ensure()
     rc=ramses_amd_mpires_godunov(p,dx,dtnew(ilevel))
     if(rc/=0)call ramses_amd_fatal('godunov_fine')
     return
  end if
#endif

  ! AMR run with the state and the tree resident on the device: the tree-walking sweep in place
  if(ramses_amd_amr_resident())then
     call ramses_amd_amr_ensure()
     rc=ramses_amd_amrres_godunov(p,ilevel,active(ilevel)%ngrid,ramses_amd_octs(ilevel),dx,dtnew(ilevel), &
          & nvector,interpol_var,interpol_type)
     if(rc/=0)call ramses_amd_fatal('godunov_fine')
     return
  end if

  ! A level with refined cells, or one that does not cover the box, takes the AMR
  ! sweep (one wavefront per oct on the tree arrays); a fully refined level without
  ! finer octs takes the dense brick sweep.
  amr_level=.false.
  if(ilevel<nlevelmax)then
     if(numbtot(1,ilevel+1)>0)amr_level=.true.
  end if
  if(int(active(ilevel)%ngrid,8)*8_8/=(2_8**ilevel)**3*int(nx_loc,8)**3)amr_level=.true.
  ! several MPI ranks (each sweeps its own active octs; ghost octs of the other ranks are
  ! ordinary neighbours in the tree) and physical boundary octs: the tree-walking sweep
  if(ncpu>1.or.nboundary>0)amr_level=.true.
  ! the dense brick entry points cover a box with nx=ny=nz=1; other coarse grids walk the tree
  if(nx_loc/=1.or.jcoarse_max/=jcoarse_min.or.kcoarse_max/=kcoarse_min)amr_level=.true.
  ! artificial diffusion (cmpdivu + consup) is implemented in the tree-walking sweep only
  if(difmag>0.0d0)amr_level=.true.
  ! so are the divu/enew updates of pressure_fix
  if(pressure_fix)amr_level=.true.

  if(amr_level)then
     ! f, divu, enew exist only with poisson resp. pressure_fix: uold stands in (never read)
     if(poisson.and.pressure_fix)then
        rc=ramses_amd_godunov_fine_amr_f90(p,ilevel,active(ilevel)%ngrid,ramses_amd_octs(ilevel),son,nbor,father, &
             & int(ngridmax,8),int(ncoarse,8),uold,unew,f,1,divu,enew,1,dx,dtnew(ilevel),nvector,interpol_var,interpol_type)
     else if(poisson)then
        rc=ramses_amd_godunov_fine_amr_f90(p,ilevel,active(ilevel)%ngrid,ramses_amd_octs(ilevel),son,nbor,father, &
             & int(ngridmax,8),int(ncoarse,8),uold,unew,f,1,uold,uold,0,dx,dtnew(ilevel),nvector,interpol_var,interpol_type)
     else if(pressure_fix)then
        rc=ramses_amd_godunov_fine_amr_f90(p,ilevel,active(ilevel)%ngrid,ramses_amd_octs(ilevel),son,nbor,father, &
             & int(ngridmax,8),int(ncoarse,8),uold,unew,uold,0,divu,enew,1,dx,dtnew(ilevel),nvector,interpol_var,interpol_type)
     else
        rc=ramses_amd_godunov_fine_amr_f90(p,ilevel,active(ilevel)%ngrid,ramses_amd_octs(ilevel),son,nbor,father, &
             & int(ngridmax,8),int(ncoarse,8),uold,unew,uold,0,uold,uold,0,dx,dtnew(ilevel),nvector,interpol_var,interpol_type)
     end if
  else if(ramses_amd_resident())then
     ! state already on the device (loaded by courant_fine or here); unew stays there
     if(poisson)then
        rc=ramses_amd_resident_godunov_grav_f90(p,ilevel,active(ilevel)%ngrid,ramses_amd_octs(ilevel),xg, &
             & int(ngridmax,8),int(ncoarse,8),nx_loc,uold,f,dx,dtnew(ilevel))
     else
        rc=ramses_amd_resident_godunov_f90(p,ilevel,active(ilevel)%ngrid,ramses_amd_octs(ilevel),xg, &
             & int(ngridmax,8),int(ncoarse,8),nx_loc,uold,dx,dtnew(ilevel))
     end if
  else if(poisson)then
     has_f=1
     rc=ramses_amd_godunov_fine_f90(p,ilevel,active(ilevel)%ngrid,ramses_amd_octs(ilevel),xg, &
          & int(ngridmax,8),int(ncoarse,8),nx_loc,uold,unew,f,has_f,dx,dtnew(ilevel))
  else
     has_f=0
     rc=ramses_amd_godunov_fine_f90(p,ilevel,active(ilevel)%ngrid,ramses_amd_octs(ilevel),xg, &
          & int(ngridmax,8),int(ncoarse,8),nx_loc,uold,unew,uold,has_f,dx,dtnew(ilevel))
  end if
  if(rc/=0)call ramses_amd_fatal('godunov_fine')
#endif

111 format('   Entering godunov_fine (MI355X) for level ',i2)

end subroutine godunov_fine


#if NDIM<3
!------------------------------------------------------------------------------
! godunov_fine(ilevel) of an NDIM = 1 / 2 build (see godunov_fine above)
!------------------------------------------------------------------------------
subroutine ramses_amd_godunov_lowdim(ilevel,p,dx)
  use amr_commons
  use hydro_commons
  use ramses_amd_iface
  implicit none
  integer::ilevel
  type(ramses_amd_hydro_params)::p
  real(dp)::dx
  integer::rc,ib,i,nb,skip(2),nloc(2)
  integer(kind=8)::ncells_int
  integer,allocatable::blist(:)
  logical::ok
  logical,save::said=.false.
  skip(1)=icoarse_min; nloc(1)=icoarse_max-icoarse_min+1
  skip(2)=jcoarse_min; nloc(2)=jcoarse_max-jcoarse_min+1
  ncells_int=int(nloc(1),8)*2_8**ilevel
  if(ndim>1)ncells_int=ncells_int*int(nloc(2),8)*2_8**ilevel
  ok=ncpu==1.and..not.poisson.and.difmag<=0.0d0.and..not.pressure_fix.and.nvar==ndim+2
  if(ilevel<nlevelmax)then
     if(numbtot(1,ilevel+1)>0)ok=.false.
  end if
  if(int(active(ilevel)%ngrid,8)*int(twotondim,8)/=ncells_int)ok=.false.
  if(.not.ok)then
     if(.not.said.and.myid==1)write(*,*)'ramses_amd: NDIM<3 build: levels that are not uniform (or runs with gravity, difmag, ', &
          & 'pressure_fix, passive scalars, several ranks) keep the reference godunov_fine'
     said=.true.
     ! counted per level by the library and printed in its exit line (nothing silent: a run that regrids away from the
     ! uniform level is a CPU run from then on)
     rc=ramses_amd_lowdim_note_reference(ilevel)
     call godunov_fine_reference(ilevel)
     return
  end if
  nb=0
  do ib=1,nboundary
     nb=nb+boundary(ib,ilevel)%ngrid
  end do
  allocate(blist(max(nb,1)))
  nb=0
  do ib=1,nboundary
     do i=1,boundary(ib,ilevel)%ngrid
        blist(nb+i)=boundary(ib,ilevel)%igrid(i)
     end do
     nb=nb+boundary(ib,ilevel)%ngrid
  end do
  rc=ramses_amd_godunov_fine_lowdim_f90(p,ilevel,active(ilevel)%ngrid,ramses_amd_octs(ilevel),nb,blist,xg, &
       & int(ngridmax,8),int(ncoarse,8),skip,nloc,uold,unew,dx,dtnew(ilevel))
  deallocate(blist)
  if(rc==-2)then
     ! RAMSES_AMD_EUNSUPPORTED: a limit of the embedded brick (level too large, boundary regions that do not cover the
     ! ghost cells, a solver the brick sweep does not have): the reference's routine, counted like the cases above
     if(myid==1)write(*,*)'ramses_amd: NDIM<3 build: level ',ilevel,' is not covered by the device sweep; reference godunov_fine'
     rc=ramses_amd_lowdim_note_reference(ilevel)
     call godunov_fine_reference(ilevel)
     return
  end if
  if(rc/=0)call ramses_amd_fatal('godunov_fine (NDIM<3)')
end subroutine ramses_amd_godunov_lowdim
#endif
