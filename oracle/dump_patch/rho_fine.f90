!==============================================================================
! oracle/dump_patch/rho_fine.f90 -- TEST INFRASTRUCTURE ONLY.
!
! Wraps the UNMODIFIED rho_fine of the reference (pm/rho_fine.f90:5-226) and
! dumps, for chosen calls, the tree, the oct list, the hydro density it reads and
! the rho / multipole / rho_tot it leaves (kernel-level goldens of the hydro
! deposit: multipole_fine + cic_from_multipole / cic_cell).
! RAMSES_DUMP_RHO = comma separated 1-based call numbers
!==============================================================================
#define rho_fine rho_fine_reference
#include "pm/rho_fine.f90"
#undef rho_fine

subroutine rho_fine(ilevel,icount)
  use amr_commons
  use hydro_commons
  use poisson_commons
  implicit none
  integer::ilevel,icount,stat,l
  integer,save::ncall=0
  logical::dumping
  character(len=256)::val
  character(len=16)::tag
  character(len=64)::fname
  if(.not.poisson)return
  if(numbtot(1,ilevel)==0)return
  ncall=ncall+1
  dumping=.false.
  call get_environment_variable('RAMSES_DUMP_RHO',val,status=stat)
  if(stat==0)then
     write(tag,'(I0)')ncall
     val=','//trim(adjustl(val))//','
     dumping=index(val,','//trim(tag)//',')>0
  end if
  if(dumping)then
     write(fname,'(A,I4.4,A)')'rho_',ncall,'_in.bin'
     open(unit=78,file=trim(fname),form='unformatted',access='stream',status='replace')
     write(78)ilevel,icount,active(ilevel)%ngrid,ngridmax,ncoarse,levelmin,nvector
     write(78)boxlen,smallr
     write(78)active(ilevel)%igrid(1:active(ilevel)%ngrid)
     write(78)xg
     write(78)son
     write(78)nbor
     write(78)father
     write(78)uold(:,1)
     close(78)
     ! the oct lists of every level the call visits (pm/rho_fine.f90:45-60: nlevelmax down to ilevel), in list order:
     ! the deposit of a level adds its contributions in that order
     write(fname,'(A,I4.4,A)')'rho_',ncall,'_lists.bin'
     open(unit=78,file=trim(fname),form='unformatted',access='stream',status='replace')
     write(78)nlevelmax
     do l=ilevel,nlevelmax
        write(78)active(l)%ngrid
        if(active(l)%ngrid>0)write(78)active(l)%igrid(1:active(l)%ngrid)
     end do
     close(78)
  end if
  call rho_fine_reference(ilevel,icount)
  if(dumping)then
     write(fname,'(A,I4.4,A)')'rho_',ncall,'_out.bin'
     open(unit=78,file=trim(fname),form='unformatted',access='stream',status='replace')
     write(78)rho
     write(78)multipole,rho_tot
     close(78)
  end if
end subroutine rho_fine
