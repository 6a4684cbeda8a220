!==============================================================================
! hydro_boundary.f90 of the ramses_amd patch directory.
!
! Shadows hydro/hydro_boundary.f90 (bin/Makefile:153 VPATH).  The untouched
! reference file is pulled in by the preprocessor with make_boundary_hydro
! renamed to make_boundary_hydro_reference; the routine of the same name below
! keeps the reference's name, argument and meaning.
!
! While the hydro state of an AMR run is device-resident (ramses_amd_iface:
! ramses_amd_amr_resident) the boundary octs of the level live in the device's
! cell vectors like every other oct, and the three calls of amr_step
!     make_boundary_hydro(i)        amr/amr_step.f90:70    after a regrid
!     make_boundary_hydro(ilevel)   amr/amr_step.f90:293   after synchro_hydro_fine
!     make_boundary_hydro(ilevel)   amr/amr_step.f90:514   after set_uold / upload_fine
! fill them there (csrc/capi_amr.hip: ramses_amd_amrres_boundary_hydro --
! reflexive and free boundaries; imposed ones too: the shim evaluates the
! reference's boundana for the cells of the region and the device stores them).
! A level the host has just rebuilt (refine_fine; it is re-sent to the device,
! boundary octs included, before the next device routine) takes the reference's
! routine; a level current on both sides takes both, so that both stay current.
!==============================================================================
#define make_boundary_hydro make_boundary_hydro_reference
#include "hydro/hydro_boundary.f90"
#undef make_boundary_hydro

subroutine make_boundary_hydro(ilevel)
  use amr_commons
  use hydro_commons
  use ramses_amd_iface
  implicit none
  integer::ilevel
  integer(8)::tm
  if(.not.simple_boundary)return
  if(ramses_amd_amr_resident())then
     if(ramses_amd_amrres_active()/=0.and.ilevel<ramses_amd_amr_reload_from)then
        ! the device holds the level
        call ramses_amd_tic(tm)
        call ramses_amd_amr_boundary(ilevel)
        call ramses_amd_toc('make_boundary_hydro (device)',ilevel,tm)
        ! the host copy is stale (it is refreshed, boundary octs included, when the host needs it)
        if(ilevel<ramses_amd_amr_host_from)return
     end if
  end if
  call make_boundary_hydro_reference(ilevel)
end subroutine make_boundary_hydro
