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

subroutine force_fine(ilevel,icount)
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
  integer::rc,nx_loc,i,ind,idim,iskip,igrid,ngrid,ncache
  real(dp)::dx,dx_loc,scale,fact,fourpi
  real(kind=8)::rho_loc,epot_loc
  integer,dimension(1:nvector),save::ind_grid,ind_cell

  if(numbtot(1,ilevel)==0)return
  nx_loc=(icoarse_max-icoarse_min+1)
  if(.not.ramses_amd_enabled().or.gravity_type>0.or.ncpu>1.or.nboundary>0.or.sink.or.ndim/=3.or.ilevel<2 &
       & .or.int(active(ilevel)%ngrid,8)*8_8/=(2_8**ilevel)**3*int(nx_loc,8)**3)then
     call force_fine_reference(ilevel,icount)
     return
  end if
  if(verbose)write(*,111)ilevel

  ! (make_boundary_phi and the halos of f are no-ops in this configuration)
  rc=ramses_amd_force_fine_f90(ilevel,active(ilevel)%ngrid,active(ilevel)%igrid,xg, &
       & int(ngridmax,8),int(ncoarse,8),nx_loc,phi,f)
  if(rc/=0)call ramses_amd_fatal('force_fine')

  ! Diagnostics (:158-190), in the reference's order of summation: potential energy of the
  ! leaf cells and maximum density
  dx=0.5D0**ilevel
  scale=boxlen/dble(nx_loc)
  dx_loc=dx*scale
  rho_loc=0
  epot_loc=0
  fourpi=2*twopi
  if(cosmo)fourpi=1.5D0*omega_m*aexp
  fact=-dx_loc**ndim/fourpi/2.0D0
  ncache=active(ilevel)%ngrid
  do igrid=1,ncache,nvector
     ngrid=MIN(nvector,ncache-igrid+1)
     do i=1,ngrid
        ind_grid(i)=active(ilevel)%igrid(igrid+i-1)
     end do
     do ind=1,twotondim
        iskip=ncoarse+(ind-1)*ngridmax
        do i=1,ngrid
           ind_cell(i)=iskip+ind_grid(i)
        end do
        do idim=1,ndim
           do i=1,ngrid
              if(son(ind_cell(i))==0)epot_loc=epot_loc+fact*f(ind_cell(i),idim)**2
           end do
        end do
        do i=1,ngrid
           rho_loc=MAX(rho_loc,dble(abs(rho(ind_cell(i)))))
        end do
     end do
  end do
  epot_tot=epot_tot+epot_loc
  rho_max(ilevel)=rho_loc

111 format('   Entering force_fine (MI355X) for level ',I2)

end subroutine force_fine
