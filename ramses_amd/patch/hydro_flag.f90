!==============================================================================
! hydro_flag.f90 of the ramses_amd patch directory.
!
! Shadows hydro/hydro_flag.f90 (hydro_flag -> hydro_flag_reference by #define + #include;
! jeans_length_refine stays the reference's).  While the hydro state of an AMR run is
! device-resident the gradient criteria (hydro_refine, hydro/godunov_utils.f90:125-263:
! err_grad_d / err_grad_p / err_grad_u on a cell and its two neighbours per direction, a missing
! neighbour replaced by the neighbouring father cell) are evaluated on the GPU; the host receives
! one flag per cell and does the reference's bookkeeping (flag1, nflag).  Geometry-based
! refinement (r_refine) needs no hydro data and is applied on the host as in the reference.
!==============================================================================
#define hydro_flag hydro_flag_reference
#include "hydro/hydro_flag.f90"
#undef hydro_flag

subroutine hydro_flag(ilevel)
  use amr_commons
  use hydro_commons
  use ramses_amd_iface
  implicit none
  integer::ilevel
  integer::i,ind,idim,iskip,igrid,ngrid,ncache,nok,rc,nx_loc,ix,iy,iz
  integer,allocatable,dimension(:)::okdev
  integer,dimension(1:nvector),save::ind_grid,ind_cell
  logical,dimension(1:nvector),save::ok
  real(dp),dimension(1:nvector,1:ndim),save::xx
  real(dp),dimension(1:twotondim,1:3)::xc
  real(dp),dimension(1:3)::skip_loc
  real(dp)::dx,scale
  type(ramses_amd_hydro_params)::p

  if(.not.ramses_amd_amr_resident())then
     call hydro_flag_reference(ilevel)
     return
  end if
  ! the reference's early returns (hydro/hydro_flag.f90:32-33,56-66)
  if(ilevel==nlevelmax)return
  if(numbtot(1,ilevel)==0)return
  if(err_grad_d==-1.0.and.err_grad_p==-1.0.and.err_grad_u==-1.0.and.jeans_refine(ilevel)==-1.0)return

  call ramses_amd_amr_ensure()
  call ramses_amd_fill_hydro_params(p)
  ncache=active(ilevel)%ngrid
  allocate(okdev(1:twotondim*ncache))
  rc=ramses_amd_amrres_hydro_flag(p,ncache,ramses_amd_octs(ilevel),dble(err_grad_d),dble(err_grad_p),dble(err_grad_u), &
       & dble(floor_d),dble(floor_p),dble(floor_u),okdev)
  if(rc/=0)call ramses_amd_fatal('hydro_flag')

  ! cell centres for the geometry criteria (:36-53)
  dx=0.5d0**ilevel
  nx_loc=(icoarse_max-icoarse_min+1)
  skip_loc=(/0.0d0,0.0d0,0.0d0/)
  skip_loc(1)=dble(icoarse_min); skip_loc(2)=dble(jcoarse_min); skip_loc(3)=dble(kcoarse_min)
  scale=boxlen/dble(nx_loc)
  do ind=1,twotondim
     iz=(ind-1)/4
     iy=(ind-1-4*iz)/2
     ix=(ind-1-2*iy-4*iz)
     xc(ind,1)=(dble(ix)-0.5D0)*dx
     xc(ind,2)=(dble(iy)-0.5D0)*dx
     xc(ind,3)=(dble(iz)-0.5D0)*dx
  end do

  ! same loop nest and bookkeeping as the reference (:84-171)
  do igrid=1,ncache,nvector
     ngrid=MIN(nvector,ncache-igrid+1)
     do i=1,ngrid
        ind_grid(i)=active(ilevel)%igrid(igrid+i-1)
     end do
     do ind=1,twotondim
        iskip=ncoarse+(ind-1)*ngridmax
        do i=1,ngrid
           ind_cell(i)=iskip+ind_grid(i)
           ok(i)=okdev((ind-1)*ncache+igrid+i-1)/=0
        end do
        if(r_refine(ilevel)>-1.0)then
           do idim=1,ndim
              do i=1,ngrid
                 xx(i,idim)=xg(ind_grid(i),idim)+xc(ind,idim)
              end do
           end do
           do idim=1,ndim
              do i=1,ngrid
                 xx(i,idim)=(xx(i,idim)-skip_loc(idim))*scale
              end do
           end do
           call geometry_refine(xx,ok,ngrid,ilevel)
        end if
        nok=0
        do i=1,ngrid
           if(flag1(ind_cell(i))==0.and.ok(i))then
              nok=nok+1
           end if
        end do
        do i=1,ngrid
           if(ok(i))flag1(ind_cell(i))=1
        end do
        nflag=nflag+nok
     end do
  end do
  deallocate(okdev)
end subroutine hydro_flag
