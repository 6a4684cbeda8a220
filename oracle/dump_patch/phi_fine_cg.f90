!==============================================================================
! oracle/dump_patch/phi_fine_cg.f90 -- TEST INFRASTRUCTURE ONLY.
!
! Wraps the UNMODIFIED phi_fine_cg of the reference
! (poisson/phi_fine_cg.f90:5-206,211-339) and dumps, for chosen solves, the
! state the iteration loop starts from (after cmp_residual_cg: tree, phi, rho,
! f(:,1:2)) and what it ends with (phi, f) -- kernel-level goldens of the
! conjugate-gradient solver on a partially refined level.
! RAMSES_DUMP_CG = comma separated 1-based solve numbers
!==============================================================================
#define phi_fine_cg phi_fine_cg_reference
#define cmp_residual_cg cmp_residual_cg_reference
#include "poisson/phi_fine_cg.f90"
#undef phi_fine_cg
#undef cmp_residual_cg

subroutine phi_fine_cg(ilevel,icount)
  use amr_commons
  use poisson_commons
  implicit none
  integer::ilevel,icount,stat
  integer,save::nsolve=0
  logical::dumping
  character(len=256)::val
  character(len=16)::tag
  character(len=64)::fname
  if(gravity_type>0)return
  if(numbtot(1,ilevel)==0)return
  nsolve=nsolve+1
  dumping=.false.
  call get_environment_variable('RAMSES_DUMP_CG',val,status=stat)
  if(stat==0)then
     write(tag,'(I0)')nsolve
     val=','//trim(adjustl(val))//','
     dumping=index(val,','//trim(tag)//',')>0
  end if
  if(dumping)then
     ! the state the iteration loop starts from: the reference's own pre-loop steps
     ! (:52-58,85), which the reference repeats identically below (they only read the
     ! coarser level, rho and the level's own boundary cells)
     if(ilevel>levelmin)then
        call make_initial_phi(ilevel,icount)
     else
        call make_multipole_phi(ilevel)
     endif
     call make_virtual_fine_dp(phi(1),ilevel)
     call make_boundary_phi(ilevel)
     call cmp_residual_cg_reference(ilevel,icount)
     write(fname,'(A,I4.4,A)')'cg_',nsolve,'_in.bin'
     open(unit=78,file=trim(fname),form='unformatted',access='stream',status='replace')
     write(78)ilevel,active(ilevel)%ngrid,ngridmax,ncoarse
     write(78)epsilon,rho_tot,boxlen
     write(78)active(ilevel)%igrid(1:active(ilevel)%ngrid)
     write(78)son
     write(78)nbor
     write(78)phi
     write(78)rho
     write(78)f
     close(78)
  end if
  call phi_fine_cg_reference(ilevel,icount)
  if(dumping)then
     write(fname,'(A,I4.4,A)')'cg_',nsolve,'_out.bin'
     open(unit=78,file=trim(fname),form='unformatted',access='stream',status='replace')
     write(78)phi
     write(78)f
     close(78)
  end if
end subroutine phi_fine_cg
