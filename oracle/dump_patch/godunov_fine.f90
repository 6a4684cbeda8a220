!==============================================================================
! oracle/dump_patch/godunov_fine.f90 -- TEST INFRASTRUCTURE ONLY.
!
! A RAMSES patch directory (same mechanism as ramses_amd/patch) that wraps the
! UNMODIFIED reference godunov_fine and dumps its inputs and outputs, to pin
! the AMR sweep of the device path against the reference itself:
! tests/golden/make_golden_amr.py runs oracle/_ref/ramses3d_dump_patch on an
! AMR namelist and turns the dumps into tests/golden/amr_godunov_ref.npz.
! Nothing of the reference is restated here.
!
! RAMSES_DUMP_CALLS = comma separated 1-based call numbers of godunov_fine to dump
!==============================================================================
#define godunov_fine godunov_fine_reference
#include "hydro/godunov_fine.f90"
#undef godunov_fine

subroutine godunov_fine(ilevel)
  use amr_commons
  use hydro_commons
  implicit none
  integer::ilevel
  integer,save::ncall=0
  logical::dump
  character(len=256)::val
  character(len=16)::tag
  integer::stat,k,n0
  if(numbtot(1,ilevel)==0)return
  if(static)return
  ncall=ncall+1
  dump=.false.
  call get_environment_variable('RAMSES_DUMP_CALLS',val,status=stat)
  if(stat==0)then
     write(tag,'(I0)')ncall
     val=','//trim(adjustl(val))//','
     dump=index(val,','//trim(tag)//',')>0
  end if
  if(dump)call dump_godunov('in',ilevel,ncall)
  call godunov_fine_reference(ilevel)
  if(dump)call dump_godunov('out',ilevel,ncall)
end subroutine godunov_fine

subroutine dump_godunov(what,ilevel,ncall)
  use amr_commons
  use hydro_commons
  use poisson_commons
  implicit none
  character(len=*)::what
  integer::ilevel,ncall,nx_loc,ipoisson,ipfix
  character(len=64)::fname
  real(dp)::dx
  write(fname,'(A,I4.4,A,A,A)')'godunov_',ncall,'_',trim(what),'.bin'
  open(unit=77,file=trim(fname),form='unformatted',access='stream',status='replace')
  nx_loc=icoarse_max-icoarse_min+1
  dx=0.5d0**ilevel*boxlen/dble(nx_loc)
  if(what=='in')then
     ipoisson=0
     if(poisson)ipoisson=1
     ipfix=0
     if(pressure_fix)ipfix=1
     write(77)ilevel,active(ilevel)%ngrid,ngridmax,ncoarse,nvar,nvector,nlevelmax,interpol_var,interpol_type,ipoisson,ipfix
     write(77)dx,dtnew(ilevel),gamma,smallr,smallc
     write(77)active(ilevel)%igrid(1:active(ilevel)%ngrid)
     write(77)son
     write(77)nbor
     write(77)father
     write(77)uold
     write(77)unew
     if(poisson)write(77)f
     if(pressure_fix)then
        write(77)divu
        write(77)enew
     end if
  else
     write(77)unew
     if(pressure_fix)then
        write(77)divu
        write(77)enew
     end if
  end if
  close(77)
end subroutine dump_godunov
