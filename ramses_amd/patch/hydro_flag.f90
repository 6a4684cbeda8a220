!==============================================================================
! hydro_flag.f90 of the ramses_amd patch directory.
!
! Shadows hydro/hydro_flag.f90 (hydro_flag -> hydro_flag_reference by #define + #include;
! jeans_length_refine stays the reference's).  While the hydro state of an AMR run is
! device-resident the gradient criteria (hydro_refine, hydro/godunov_utils.f90:125-263:
! err_grad_d / err_grad_p / err_grad_u on a cell and its two neighbours per direction, a missing
! neighbour replaced by the neighbouring father cell) are evaluated on the GPU, which also compacts
! the answer: the host receives the LIST of cells that ask for refinement (a few thousand ints at
! most, not one flag per cell of the level) and marks them in flag1.  Geometry-based refinement
! (r_refine: a mask on the cell position, no hydro data) filters that list through the reference's
! own geometry_refine, a batch of nvector candidates at a time.
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
  integer::ncache,ncand,rc,k,n,j,d,icell,ind,ig
  integer,allocatable,dimension(:)::cand
  logical,allocatable,dimension(:)::keep
  logical,dimension(1:nvector)::inside
  real(dp),dimension(1:nvector,1:ndim)::pos
  real(dp)::half,scale
  real(dp),dimension(1:3)::corner
  type(ramses_amd_hydro_params)::p
  integer(8)::t0

  if(.not.ramses_amd_amr_resident())then
     call hydro_flag_reference(ilevel)
     return
  end if
  ! the reference's early returns (hydro/hydro_flag.f90:32-33,56-66)
  if(ilevel==nlevelmax)return
  if(numbtot(1,ilevel)==0)return
  if(err_grad_d==-1.0.and.err_grad_p==-1.0.and.err_grad_u==-1.0.and.jeans_refine(ilevel)==-1.0)return

  call ramses_amd_amr_ensure()
  call ramses_amd_tic(t0)
  call ramses_amd_fill_hydro_params(p)
  ncache=active(ilevel)%ngrid
  allocate(cand(1:twotondim*max(ncache,1)))
  rc=ramses_amd_amrres_hydro_flag(p,ncache,ramses_amd_octs(ilevel),dble(err_grad_d),dble(err_grad_p),dble(err_grad_u), &
       & dble(floor_d),dble(floor_p),dble(floor_u),cand,ncand)
  if(rc/=0)call ramses_amd_fatal('hydro_flag')

  allocate(keep(1:max(ncand,1)))
  keep=.true.
  if(r_refine(ilevel)>-1.0)then
     ! candidate cell -> (octant, oct) -> centre in user units, then the reference's region test
     half=0.5d0**(ilevel+1)
     scale=boxlen/dble(icoarse_max-icoarse_min+1)
     corner=(/dble(icoarse_min),dble(jcoarse_min),dble(kcoarse_min)/)
     do k=1,ncand,nvector
        n=min(nvector,ncand-k+1)
        do j=1,n
           icell=cand(k+j-1)
           ind=(icell-ncoarse-1)/ngridmax
           ig=icell-ncoarse-ind*ngridmax
           do d=1,ndim
              pos(j,d)=(xg(ig,d)+merge(half,-half,btest(ind,d-1))-corner(d))*scale
           end do
           inside(j)=.true.
        end do
        call geometry_refine(pos,inside,n,ilevel)
        keep(k:k+n-1)=inside(1:n)
     end do
  end if

  ! mark the survivors; nflag counts the cells that were not marked before
  do k=1,ncand
     if(keep(k))then
        icell=cand(k)
        if(flag1(icell)==0)nflag=nflag+1
        flag1(icell)=1
     end if
  end do
  deallocate(cand,keep)
  call ramses_amd_toc('hydro_flag (shim: kernel + list + flag1)',ilevel,t0)
end subroutine hydro_flag
