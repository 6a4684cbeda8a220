!==============================================================================
! oracle/dump_patch/multigrid_fine_fine.f90 -- TEST INFRASTRUCTURE ONLY.
!
! Wraps the UNMODIFIED gauss_seidel_mg_fine and cmp_residual_mg_fine of the
! reference and dumps their inputs/outputs for chosen calls on partially
! refined levels (kernel-level goldens of the AMR multigrid).
! RAMSES_DUMP_MG = comma separated 1-based call numbers (counted over both routines)
!==============================================================================
#define gauss_seidel_mg_fine gauss_seidel_mg_fine_reference
#define cmp_residual_mg_fine cmp_residual_mg_fine_reference
#include "poisson/multigrid_fine_fine.f90"
#undef gauss_seidel_mg_fine
#undef cmp_residual_mg_fine

logical function dump_mg_wanted(ncall)
  implicit none
  integer::ncall,stat
  character(len=256)::val
  character(len=16)::tag
  dump_mg_wanted=.false.
  call get_environment_variable('RAMSES_DUMP_MG',val,status=stat)
  if(stat==0)then
     write(tag,'(I0)')ncall
     val=','//trim(adjustl(val))//','
     dump_mg_wanted=index(val,','//trim(tag)//',')>0
  end if
end function dump_mg_wanted

subroutine dump_mg_state(what,ikind,ilevel,ncall,iflag)
  use amr_commons
  use poisson_commons
  implicit none
  character(len=*)::what
  integer::ikind,ilevel,ncall,iflag
  character(len=64)::fname
  write(fname,'(A,I4.4,A,A,A)')'mgfine_',ncall,'_',trim(what),'.bin'
  open(unit=78,file=trim(fname),form='unformatted',access='stream',status='replace')
  if(what=='in')then
     write(78)ikind,ilevel,active(ilevel)%ngrid,ngridmax,ncoarse,iflag
     write(78)active(ilevel)%igrid(1:active(ilevel)%ngrid)
     write(78)son
     write(78)nbor
     write(78)flag2
     write(78)phi
     write(78)f
  else
     write(78)phi
     write(78)f
  end if
  close(78)
end subroutine dump_mg_state

subroutine gauss_seidel_mg_fine(ilevel,redstep)
  use amr_commons
  use poisson_commons
  implicit none
  integer, intent(in) :: ilevel
  logical, intent(in) :: redstep
  integer,save::ncall=0
  logical::dump_mg_wanted,d
  integer::iflag
  ncall=ncall+1
  d=dump_mg_wanted(ncall)
  iflag=0; if(redstep)iflag=1; if(safe_mode(ilevel))iflag=iflag+2
  if(d)call dump_mg_state('in',1,ilevel,ncall,iflag)
  call gauss_seidel_mg_fine_reference(ilevel,redstep)
  if(d)call dump_mg_state('out',1,ilevel,ncall,iflag)
end subroutine gauss_seidel_mg_fine

subroutine cmp_residual_mg_fine(ilevel)
  use amr_commons
  implicit none
  integer, intent(in) :: ilevel
  integer,save::ncall=1000
  logical::dump_mg_wanted,d
  ncall=ncall+1
  d=dump_mg_wanted(ncall)
  if(d)call dump_mg_state('in',2,ilevel,ncall,0)
  call cmp_residual_mg_fine_reference(ilevel)
  if(d)call dump_mg_state('out',2,ilevel,ncall,0)
end subroutine cmp_residual_mg_fine
