!==============================================================================
! ramses_amd_iface -- what the shims of the RAMSES patch directory ramses_amd/patch (make PATCH=...) share: the switches
! (RAMSES_AMD, residency modes), the state of the device-resident paths, the helpers that translate the reference's
! module data into calls of the C ABI.  The C ABI itself (bind(C) types and interfaces) is module ramses_amd_cabi.
!==============================================================================
module ramses_amd_iface
  use iso_c_binding
  use ramses_amd_cabi
  implicit none

  ! stands in for active(l)%igrid where a rank holds no oct of a level (see ramses_amd_octs)
  integer, target, save :: ramses_amd_no_octs(1) = 0
  logical, save :: ramses_amd_checked = .false.
  logical, save :: ramses_amd_arith_said = .false.
  logical, save :: ramses_amd_on = .true.
  ! AMR multigrid: the reference driver is running with the device routines (set by the
  ! multigrid_fine shim); the level arrays have been handed to the device (first routine call)
  logical, save :: ramses_amd_mg_active = .false.
  logical, save :: ramses_amd_mg_started = .false.
  integer, save :: ramses_amd_mg_level = 0
  logical, save :: ramses_amd_res_checked = .false.
  logical, save :: ramses_amd_res_on = .false.
  ! the Poisson fields of the resident level are on the device (rho_fine's deposit ran there this step)
  logical, save :: ramses_amd_pois_dev = .false.
  ! MPI: the rank's share of the level stays on its GPU (ramses_amd_mpi_resident); the halo exchange
  ! goes over RCCL (ramses_amd_halo_rccl) or, failing that, through the program's own MPI on pinned buffers
  logical, save :: ramses_amd_mpi_checked = .false.
  logical, save :: ramses_amd_mpi_on = .false.
  logical, save :: ramses_amd_halo_rccl = .false.
  ! AMR runs: uold/unew and the tree stay on the device between the hydro routines of amr_step
  ! (ramses_amd_amr_resident); armed by the first courant_fine of the time loop, so that the
  ! initial mesh construction (init_refine_2, init_flow_fine) runs on the host arrays as ever
  logical, save :: ramses_amd_amr_checked = .false.
  logical, save :: ramses_amd_amr_ok = .false.
  logical, save :: ramses_amd_amr_armed = .false.
  integer, save :: ramses_amd_amr_reload_from = 1000   ! levels >= this were rebuilt on the host and await their reload
  integer, save :: ramses_amd_amr_host_from = 1000     ! levels >= this are current on the host (synced or rebuilt) since the last device routine
  ! advanced whenever the reference may have changed the tree (refine_fine); the device copies of son/nbor/father
  ! are re-sent when their epoch is behind
  ! MAXITER of the reference's V-cycle loop (poisson/multigrid_fine_commons.f90:34) = MAXITER of csrc/capi.hip, pois_amr.hip, mg_dist.hip
  integer, parameter :: ramses_amd_mg_maxiter = 10
  integer, parameter :: RAMSES_AMD_EUNSUPPORTED_CODE = -2     ! include/ramses_amd.h: RAMSES_AMD_EUNSUPPORTED
  integer, save :: ramses_amd_tree_epoch = 0
  ! AMR residency with several MPI ranks: count of build_comm calls per level (the device copy of a level's
  ! communicators is re-sent when its epoch is behind); the transport has been chosen (ramses_amd_halo_init)
  integer, save :: ramses_amd_comm_epoch(1:64) = 1
  ! levels whose acceleration force_fine left on the device only (several ranks, resident cell vectors: patch/force_fine.f90)
  logical, save :: ramses_amd_f_on_device(1:64) = .false.
  ! a run with ONE level (levelmin = nlevelmax), several ranks, the distributed dense multigrid and resident cell vectors, from its
  ! second solve on: rho_fine's deposit stays on the device, the solve reads it there, phi stays on the rank's brick
  ! (ramses_amd_mgdist_multigrid); ramses_amd_pois_mpi_sync_host brings both to the host vectors for backup_poisson / load_balance
  logical, save :: ramses_amd_pois_mpi_dev = .false.
  logical, save :: ramses_amd_phi_on_device = .false.
  logical, save :: ramses_amd_amr_halo_ready = .false.
  ! the AMR level whose potential the device multigrid driver has just left on the device (0: none)
  integer, save :: ramses_amd_pois_amr_level = 0
  logical, save :: ramses_amd_mg_mpi_said = .false.
  ! several ranks: the levels of the solve stay on the device between the routines, their virtual boundaries are exchanged
  ! from there (ramses_amd_mg_halo).  (Round 2's path -- every routine exchanging its arrays with the host, the reference's host
  ! exchanges in between -- and its switch are gone.)
  logical, save :: ramses_amd_mg_mpi_resident = .false.
  logical, save :: ramses_amd_mg_comm_done(64) = .false.
  logical, save :: ramses_amd_halo_inited = .false.
  ! levelmin fully refined and periodic, several ranks whose domains are boxes: multigrid_fine through the distributed
  ! dense driver (csrc/mg_dist.hip); RAMSES_AMD_MG_DIST=0 keeps the multigrid of AMR levels (the round-3 path)
  type(c_ptr), save :: ramses_amd_mgdist_ctx = c_null_ptr
  integer, save :: ramses_amd_mgdist_level = 0
  integer, save :: ramses_amd_mgdist_pg(3) = 0
  integer, allocatable, save :: ramses_amd_mgdist_rob(:)
  logical, save :: ramses_amd_mgdist_said = .false.
  ! the level whose potential the distributed solve has just left on the device (0: none), and the rank's box origin:
  ! force_fine of that level computes the acceleration there (ramses_amd_mgdist_force_fine)
  integer, save :: ramses_amd_mgdist_phi_level = 0
  integer, save :: ramses_amd_mgdist_lo(3) = 0
  type(ramses_amd_mg_transport), save, target :: ramses_amd_mgdist_tr

contains

  !---------------------------------------------------------------------------
  ! The oct list of a level as an actual argument.  active(l)%igrid is a POINTER component that the reference
  ! allocates only while active(l)%ngrid > 0 (amr/virtual_boundaries.f90 build_comm; amr/init_amr.f90 leaves it
  ! undefined): with several ranks a rank may hold no oct of a level that exists, and handing the undefined
  ! pointer to an assumed-size dummy makes the compiler inspect a garbage descriptor (contiguity check / copy-in).
  !---------------------------------------------------------------------------
  function ramses_amd_octs(ilevel) result(p)
    use amr_commons, only: active
    integer, intent(in) :: ilevel
    integer, pointer :: p(:)
    if (active(ilevel)%ngrid > 0) then
       p => active(ilevel)%igrid
    else
       p => ramses_amd_no_octs
    end if
  end function ramses_amd_octs

  !---------------------------------------------------------------------------
  ! Run-time A/B switch: RAMSES_AMD=0 in the environment selects the untouched
  ! reference routines (compiled into the same binary under *_reference names).
  !---------------------------------------------------------------------------
  logical function ramses_amd_enabled()
    character(len=16) :: val
    integer :: stat, rc
    type(ramses_amd_hydro_params) :: p
    type(ramses_amd_brick) :: b
    if (.not. ramses_amd_checked) then
       call get_environment_variable('RAMSES_AMD', val, status=stat)
       if (stat == 0) ramses_amd_on = (trim(val) /= '0')
       if (ramses_amd_on) then
          rc = ramses_amd_abi_check(c_sizeof(p), c_sizeof(b))
          if (rc /= 0) call ramses_amd_fatal('ramses_amd_abi_check')
          rc = ramses_amd_set_device_auto(ramses_amd_world_rank())
          if (rc /= 0) call ramses_amd_fatal('ramses_amd_set_device_auto')
          call ramses_amd_check_build()
          call ramses_amd_pin_arrays()
          rc = ramses_amd_warmup()
          if (rc /= 0) call ramses_amd_fatal('ramses_amd_warmup')
       end if
       ramses_amd_checked = .true.
    end if
    ramses_amd_enabled = ramses_amd_on
  end function ramses_amd_enabled

  !---------------------------------------------------------------------------
  ! What the build must look like for the device hydro path: NENER=0.  With non-thermal
  ! energies the reference folds them into the pressure in ctoprim, cmpdt, the Riemann
  ! solvers and set_uold (hydro/godunov_fine.f90:83,166,214; hydro/umuscl.f90;
  ! hydro/godunov_utils.f90:40-79); the device kernels would treat variables 6.. as
  ! passive scalars.  Stop instead (RAMSES_AMD=0 runs the reference path).
  !---------------------------------------------------------------------------
  subroutine ramses_amd_check_build()
    use amr_commons, only: hydro
    use hydro_parameters, only: nener
    if (hydro .and. nener > 0) then
       write(*,*) 'ramses_amd: this binary was built with NENER=', nener, &
            & ' (non-thermal energies); the device hydro path implements NENER=0 only'
       call ramses_amd_fatal('build check (NENER>0)')
    end if
  end subroutine ramses_amd_check_build

  !---------------------------------------------------------------------------
  ! The Poisson entry points of the C ABI hard-code the 3-D tree layout
  ! (ncell = ncoarse+8*ngridmax, nbor(1:ngridmax,1:6)): a NDIM=1/2 build must
  ! take the reference routines with RAMSES_AMD=0.
  !---------------------------------------------------------------------------
  subroutine ramses_amd_need_ndim3(where)
    use amr_parameters, only: ndim
    character(len=*), intent(in) :: where
    if (ndim /= 3) then
       write(*,*) 'ramses_amd: ', where, ' on the device needs an NDIM=3 build (this one has NDIM=', ndim, ')'
       call ramses_amd_fatal(where)
    end if
  end subroutine ramses_amd_need_ndim3

  !---------------------------------------------------------------------------
  ! Page-lock the module arrays the staged entry points copy from and to (allocated once,
  ! hydro/init_hydro.f90:30-32, poisson/init_poisson.f90:24-28, amr/init_amr.f90:52-55,
  ! 227-233): the copies then run at DMA speed.  Never fatal.
  !---------------------------------------------------------------------------
  subroutine ramses_amd_pin_arrays()
    use amr_commons
    use hydro_commons
    use poisson_commons
    integer :: rc
    integer(c_int64_t) :: nc
    nc = int(ncoarse, 8) + int(twotondim, 8) * int(ngridmax, 8)
    if (hydro) then
       if (allocated(uold)) rc = ramses_amd_host_register_dp(uold, nc * int(size(uold, 2), 8) * 8_8)
       if (allocated(unew)) rc = ramses_amd_host_register_dp(unew, nc * int(size(unew, 2), 8) * 8_8)
    end if
    if (poisson) then
       if (allocated(phi)) rc = ramses_amd_host_register_dp(phi, nc * 8_8)
       if (allocated(rho)) rc = ramses_amd_host_register_dp(rho, nc * 8_8)
       if (allocated(f)) rc = ramses_amd_host_register_dp(f, nc * int(size(f, 2), 8) * 8_8)
    end if
    if (allocated(son)) rc = ramses_amd_host_register_int(son, nc * 4_8)
    if (allocated(nbor)) rc = ramses_amd_host_register_int(nbor, int(ngridmax, 8) * int(size(nbor, 2), 8) * 4_8)
    if (allocated(father)) rc = ramses_amd_host_register_int(father, int(ngridmax, 8) * 4_8)
  end subroutine ramses_amd_pin_arrays

  !---------------------------------------------------------------------------
  ! First device routine of an AMR multigrid solve: hand the tree, the fine level and
  ! the multigrid levels the reference has just built to the device.
  !---------------------------------------------------------------------------
  subroutine ramses_amd_mg_ensure()
    use amr_commons
    use poisson_commons
    integer :: rc, l, ilevel, icpu, ntot, n, i, stat
    integer, allocatable :: list(:)
    character(len=16) :: val
    if (ramses_amd_mg_started) return
    ilevel = ramses_amd_mg_level
    if (ncpu == 1) then
       rc = ramses_amd_mgamr_begin(ilevel, int(ngridmax, 8), int(ncoarse, 8), son, nbor, father, lookup_mg, flag2(1), &   ! flag2 is (0:ncell)
            & phi, f, active(ilevel)%ngrid, ramses_amd_octs(ilevel))
       if (rc /= 0) call ramses_amd_fatal('multigrid_fine (AMR level, begin)')
       do l = 1, ilevel - 1
          if (active_mg(myid, l)%ngrid > 0) then
             rc = ramses_amd_mgamr_add_level(l, active_mg(myid, l)%ngrid, active_mg(myid, l)%igrid, &
                  & active_mg(myid, l)%u, active_mg(myid, l)%f)
             if (rc /= 0) call ramses_amd_fatal('multigrid_fine (AMR level, add_level)')
          end if
       end do
    else
       ! several ranks: the level's reception octs follow the active ones (their phi, mask and residual are kept current in
       ! the host cell vectors by the reference's make_virtual_fine_dp); every multigrid level is this rank's buffer
       ! followed by its reception buffers active_mg(icpu,l), resident on the device for the solve
       ramses_amd_mg_mpi_resident = .true.
       ramses_amd_mg_comm_done = .false.
       rc = ramses_amd_mgamr_force_sync(0)
       if (.not. ramses_amd_mg_mpi_said .and. myid == 1) then
          write(*,*) 'ramses_amd: multigrid under MPI: levels resident on the GPUs (own + reception octs), virtual boundaries exchanged from the device'
          ramses_amd_mg_mpi_said = .true.
       end if
       ntot = active(ilevel)%ngrid
       do icpu = 1, ncpu
          ntot = ntot + reception(icpu, ilevel)%ngrid
       end do
       allocate(list(1:max(ntot, 1)))
       n = active(ilevel)%ngrid
       do i = 1, n
          list(i) = active(ilevel)%igrid(i)
       end do
       do icpu = 1, ncpu
          do i = 1, reception(icpu, ilevel)%ngrid
             list(n + i) = reception(icpu, ilevel)%igrid(i)
          end do
          n = n + reception(icpu, ilevel)%ngrid
       end do
       rc = ramses_amd_mgamr_begin(ilevel, int(ngridmax, 8), int(ncoarse, 8), son, nbor, father, lookup_mg, flag2(1), &
            & phi, f, ntot, list)
       if (rc /= 0) call ramses_amd_fatal('multigrid_fine (AMR level, begin)')
       rc = ramses_amd_mgamr_fine_active(active(ilevel)%ngrid)
       if (rc /= 0) call ramses_amd_fatal('multigrid_fine (AMR level, fine_active)')
       deallocate(list)
       do l = 1, ilevel - 1
          ntot = 0
          do icpu = 1, ncpu
             ntot = ntot + active_mg(icpu, l)%ngrid
          end do
          if (ntot == 0) cycle
          rc = ramses_amd_mgamr_level_begin(l, ntot)
          if (rc /= 0) call ramses_amd_fatal('multigrid_fine (AMR level, level_begin)')
          ! this rank's own buffer first (possibly empty), then the others in rank order
          if (active_mg(myid, l)%ngrid > 0) then
             rc = ramses_amd_mgamr_level_block(l, active_mg(myid, l)%ngrid, active_mg(myid, l)%igrid, &
                  & active_mg(myid, l)%u, active_mg(myid, l)%f)
          else
             rc = ramses_amd_mgamr_level_block(l, 0, ramses_amd_octs(ilevel), phi, flag2(1))      ! (arrays unused)
          end if
          if (rc /= 0) call ramses_amd_fatal('multigrid_fine (AMR level, level_block)')
          do icpu = 1, ncpu
             if (icpu == myid .or. active_mg(icpu, l)%ngrid == 0) cycle
             rc = ramses_amd_mgamr_level_block(l, active_mg(icpu, l)%ngrid, active_mg(icpu, l)%igrid, &
                  & active_mg(icpu, l)%u, active_mg(icpu, l)%f)
             if (rc /= 0) call ramses_amd_fatal('multigrid_fine (AMR level, level_block)')
          end do
       end do
    end if
    ramses_amd_mg_started = .true.
  end subroutine ramses_amd_mg_ensure

  !---------------------------------------------------------------------------
  ! the AMR multigrid routines run on the device while a solve is under way (ibit: the routine, kept for the callers' sake)
  !---------------------------------------------------------------------------
  logical function ramses_amd_mg_on_device(ibit)
    integer, intent(in) :: ibit
    ramses_amd_mg_on_device = ramses_amd_mg_active .and. ibit >= 0
  end function ramses_amd_mg_on_device

  ! RAMSES_AMD_PROFILE=1: wall time per shadowed routine and level (table printed when the program ends)
  subroutine ramses_amd_tic(t0)
    integer(8), intent(out) :: t0
    call system_clock(t0)
  end subroutine ramses_amd_tic
  subroutine ramses_amd_toc(name, level, t0)
    character(len=*), intent(in) :: name
    integer, intent(in) :: level
    integer(8), intent(in) :: t0
    integer(8) :: t1, rate
    integer :: rc
    call system_clock(t1, rate)
    rc = ramses_amd_prof_add(trim(name)//c_null_char, level, dble(t1 - t0) / dble(rate))
  end subroutine ramses_amd_toc

  ! AMR multigrid: driver and per-solve setup on the device (default) or the reference's (RAMSES_AMD_MG_DRIVER=host)
  logical function ramses_amd_mg_device_driver()
    character(len=16) :: val
    integer :: stat
    integer, save :: state = -1
    if (state < 0) then
       state = 1
       call get_environment_variable('RAMSES_AMD_MG_DRIVER', val, status=stat)
       if (stat == 0) then
          if (trim(val) == 'host') state = 0
       end if
    end if
    ramses_amd_mg_device_driver = state == 1
  end function ramses_amd_mg_device_driver

  integer function ramses_amd_world_rank()
    use amr_commons, only: myid
    ramses_amd_world_rank = myid - 1
  end function ramses_amd_world_rank

  !---------------------------------------------------------------------------
  ! Device residency of the hydro state across courant_fine / set_unew /
  ! godunov_fine / set_uold (SURVEY.md 8f rank 1).  Only taken when no host
  ! routine reads or writes uold between two hydro steps: a hydro-only,
  ! single-rank, periodic run on one fully refined level.  backup_hydro (the
  ! only remaining host reader) syncs the host array first.  Anything else
  ! uses the staging path (state copied in and out around each sweep).
  ! RAMSES_AMD_RESIDENT=0 forces the staging path.
  !---------------------------------------------------------------------------
  logical function ramses_amd_resident()
    use amr_commons
    use hydro_parameters
    use poisson_parameters, only: gravity_type
#if USE_TURB==1
    use turb_commons, only: turb
#endif
    character(len=16) :: val
    integer :: stat
    if (.not. ramses_amd_res_checked) then
       ramses_amd_res_on = ramses_amd_enabled()
       call get_environment_variable('RAMSES_AMD_RESIDENT', val, status=stat)
       if (stat == 0) then
          if (trim(val) == '0') ramses_amd_res_on = .false.
       end if
       if (ncpu > 1 .or. levelmin /= nlevelmax .or. nboundary > 0) ramses_amd_res_on = .false.
       if (.not. hydro .or. pic .or. rt .or. cooling .or. star .or. sink .or. stellar) ramses_amd_res_on = .false.
       ! with self-gravity the level stays resident too (synchro_hydro_fine, the gravity terms of
       ! courant_fine / godunov_fine / set_uold and force_fine run on the device, rho_fine gets the
       ! density back); RAMSES_AMD_RESIDENT_GRAV=0 keeps such runs on the staging path
       if (poisson) then
          call get_environment_variable('RAMSES_AMD_RESIDENT_GRAV', val, status=stat)
          if (stat == 0) then
             if (trim(val) == '0') ramses_amd_res_on = .false.
          end if
          if (gravity_type > 0 .or. cosmo) ramses_amd_res_on = .false.
       end if
       if (tracer .or. MC_tracer .or. clumpfind .or. lightcone .or. movie .or. aton) ramses_amd_res_on = .false.
       if (static .or. static_gas .or. neq_chem .or. barotropic_eos .or. isothermal .or. metal) ramses_amd_res_on = .false.
       if (difmag > 0.0d0) ramses_amd_res_on = .false.
       if (pressure_fix .or. T2_star > 0.0d0 .or. momentum_feedback > 0 .or. strict_equilibrium > 0) &
            & ramses_amd_res_on = .false.
       if (ndim /= 3) ramses_amd_res_on = .false.
       ! the dense brick entry points cover a box with nx=ny=nz=1
       if (icoarse_max - icoarse_min /= 0 .or. jcoarse_max - jcoarse_min /= 0 .or. kcoarse_max - kcoarse_min /= 0) &
            & ramses_amd_res_on = .false.
#if USE_TURB==1
       ! the turbulent forcing (calc_turb_forcing / synchro_hydro_fine(...,2)) is host code on uold
       if (turb) ramses_amd_res_on = .false.
#endif
       ramses_amd_res_checked = .true.
       if (ramses_amd_res_on .and. myid == 1) &
            & write(*,*) 'ramses_amd: hydro state of level ', levelmin, ' stays resident on the GPU'
    end if
    ramses_amd_resident = ramses_amd_res_on
  end function ramses_amd_resident

  !---------------------------------------------------------------------------
  ! Device residency under MPI (one rank per GPU): a hydro-only periodic run on one fully refined
  ! level whose Hilbert domains are boxes (2^k ranks).  Every rank keeps its box as a brick with a
  ! one-oct ghost layer; courant_fine, set_unew, godunov_fine, set_uold and the two halo routines
  ! of amr_step (make_virtual_reverse_dp on unew, make_virtual_fine_dp on uold) run on the device
  ! (shims godunov_fine.f90, courant_fine.f90, virtual_boundaries.f90).  Decided once, by all ranks
  ! together, the first time the level is stepped (the communicators exist by then).
  ! RAMSES_AMD_RESIDENT=0 keeps the run on the tree-walking sweep + host MPI.
  !---------------------------------------------------------------------------
  logical function ramses_amd_mpi_resident()
    use amr_commons
    use hydro_parameters
    use mpi_mod
#if USE_TURB==1
    use turb_commons, only: turb
#endif
    character(len=16) :: val
    integer :: stat, ok, okall, info
    if (.not. ramses_amd_mpi_checked) then
       ramses_amd_mpi_checked = .true.
       ramses_amd_mpi_on = ramses_amd_enabled() .and. ncpu > 1
#ifdef WITHOUTMPI
       ramses_amd_mpi_on = .false.
#endif
#ifdef LIGHT_MPI_COMM
       ramses_amd_mpi_on = .false.     ! the condensed communicators of LIGHT_MPI_COMM are not mirrored
#endif
       call get_environment_variable('RAMSES_AMD_RESIDENT', val, status=stat)
       if (stat == 0) then
          if (trim(val) == '0') ramses_amd_mpi_on = .false.
       end if
       if (levelmin /= nlevelmax .or. nboundary > 0 .or. nremap > 0) ramses_amd_mpi_on = .false.
       if (.not. hydro .or. poisson .or. pic .or. rt .or. cooling .or. star .or. sink .or. stellar) ramses_amd_mpi_on = .false.
       if (tracer .or. MC_tracer .or. clumpfind .or. lightcone .or. movie .or. aton) ramses_amd_mpi_on = .false.
       if (static .or. static_gas .or. neq_chem .or. barotropic_eos .or. isothermal .or. metal) ramses_amd_mpi_on = .false.
       if (difmag > 0.0d0 .or. pressure_fix .or. T2_star > 0.0d0 .or. momentum_feedback > 0 .or. strict_equilibrium > 0) &
            & ramses_amd_mpi_on = .false.
       if (ndim /= 3) ramses_amd_mpi_on = .false.
       if (icoarse_max - icoarse_min /= 0 .or. jcoarse_max - jcoarse_min /= 0 .or. kcoarse_max - kcoarse_min /= 0) &
            & ramses_amd_mpi_on = .false.
#if USE_TURB==1
       if (turb) ramses_amd_mpi_on = .false.
#endif
#ifndef WITHOUTMPI
       if (ramses_amd_mpi_on) then
          ! every rank's octs must fill a box and its reception lists the shell around it
          ok = 0
          if (ramses_amd_mpi_plan_ok()) ok = 1
          call MPI_ALLREDUCE(ok, okall, 1, MPI_INTEGER, MPI_MIN, MPI_COMM_WORLD, info)
          if (okall == 0) then
             ramses_amd_mpi_on = .false.
             if (myid == 1) write(*,*) 'ramses_amd: the rank domains of level ', levelmin, &
                  & ' are not boxes: tree-walking sweep + host MPI halo'
          end if
       end if
       if (ramses_amd_mpi_on) then
          call ramses_amd_halo_init()
          if (myid == 1) write(*,*) 'ramses_amd: hydro state of level ', levelmin, &
               & ' stays resident on the GPUs (one brick per rank)'
       end if
#endif
    end if
    ramses_amd_mpi_resident = ramses_amd_mpi_on
  end function ramses_amd_mpi_resident

#ifndef WITHOUTMPI
  !---------------------------------------------------------------------------
  ! Concatenated emission / reception oct lists of one level (amr/amr_commons.f90:170-179)
  !---------------------------------------------------------------------------
  subroutine ramses_amd_comm_lists(ilevel, em_n, em_ig, rc_n, rc_ig)
    use amr_commons
    integer, intent(in) :: ilevel
    integer, allocatable, intent(out) :: em_n(:), em_ig(:), rc_n(:), rc_ig(:)
    integer :: icpu, nem, nrc, i
    allocate(em_n(ncpu), rc_n(ncpu))
    nem = 0; nrc = 0
    do icpu = 1, ncpu
       em_n(icpu) = emission(icpu, ilevel)%ngrid
       rc_n(icpu) = reception(icpu, ilevel)%ngrid
       nem = nem + em_n(icpu); nrc = nrc + rc_n(icpu)
    end do
    allocate(em_ig(max(nem, 1)), rc_ig(max(nrc, 1)))
    nem = 0; nrc = 0
    do icpu = 1, ncpu
       do i = 1, em_n(icpu)
          em_ig(nem + i) = emission(icpu, ilevel)%igrid(i)
       end do
       do i = 1, rc_n(icpu)
          rc_ig(nrc + i) = reception(icpu, ilevel)%igrid(i)
       end do
       nem = nem + em_n(icpu); nrc = nrc + rc_n(icpu)
    end do
  end subroutine ramses_amd_comm_lists

  logical function ramses_amd_mpi_plan_ok()
    use amr_commons
    integer, allocatable :: em_n(:), em_ig(:), rc_n(:), rc_ig(:)
    integer :: rc, box(8)
    if (active(levelmin)%ngrid == 0) then
       ramses_amd_mpi_plan_ok = .false.
       return
    end if
    call ramses_amd_comm_lists(levelmin, em_n, em_ig, rc_n, rc_ig)
    rc = ramses_amd_halo_plan(levelmin, active(levelmin)%ngrid, ramses_amd_octs(levelmin), xg, int(ngridmax, 8), ncpu, &
         & em_n, em_ig, rc_n, rc_ig, box, c_null_ptr, c_null_ptr, c_null_ptr, c_null_ptr, 0_8)
    ramses_amd_mpi_plan_ok = (rc == 0)
  end function ramses_amd_mpi_plan_ok

  !---------------------------------------------------------------------------
  ! Load the level onto the device if it is not there (first step, or after build_comm)
  !---------------------------------------------------------------------------
  subroutine ramses_amd_mpires_ensure()
    use amr_commons
    use hydro_commons
    integer, allocatable :: em_n(:), em_ig(:), rc_n(:), rc_ig(:)
    type(ramses_amd_hydro_params) :: p
    integer :: rc, nx_loc
    if (ramses_amd_mpires_active() /= 0) return
    call ramses_amd_fill_hydro_params(p)
    nx_loc = icoarse_max - icoarse_min + 1
    call ramses_amd_comm_lists(levelmin, em_n, em_ig, rc_n, rc_ig)
    rc = ramses_amd_mpires_setup(p, levelmin, active(levelmin)%ngrid, ramses_amd_octs(levelmin), xg, int(ngridmax, 8), &
         & int(ncoarse, 8), nx_loc, uold, unew, ncpu, myid, em_n, em_ig, rc_n, rc_ig)
    if (rc /= 0) call ramses_amd_fatal('device-resident level under MPI (setup)')
  end subroutine ramses_amd_mpires_ensure

  !---------------------------------------------------------------------------
  ! Transport of the halo exchange.  RAMSES_AMD_HALO = rccl | host | auto (default): RCCL neighbour
  ! send/recv over xGMI when every rank has its own GPU; several ranks on one GPU (RCCL refuses that)
  ! or RAMSES_AMD_HALO=host: the program's own MPI on pinned host buffers, and the run says so.
  !---------------------------------------------------------------------------
  subroutine ramses_amd_halo_init()
    use amr_commons
    use mpi_mod
    character(len=16) :: val
    character(kind=c_char) :: id(128)
    integer :: stat, info, rc, rcmax, i, j
    integer(c_int64_t) :: uid
    integer(c_int64_t), allocatable :: alluid(:)
    logical :: dup
    val = 'auto'
    call get_environment_variable('RAMSES_AMD_HALO', val, status=stat)
    if (stat /= 0) val = 'auto'
    ramses_amd_halo_inited = .true.
    ramses_amd_halo_rccl = .false.
    if (trim(val) == 'host') then
       if (myid == 1) write(*,*) 'ramses_amd: halo exchange staged through host MPI (RAMSES_AMD_HALO=host)'
       return
    end if
    allocate(alluid(ncpu))
    rc = ramses_amd_device_uid(uid)
    if (rc /= 0) call ramses_amd_fatal('halo transport (device id)')
    call MPI_ALLGATHER(uid, 1, MPI_INTEGER8, alluid, 1, MPI_INTEGER8, MPI_COMM_WORLD, info)
    dup = .false.
    do i = 1, ncpu
       do j = i + 1, ncpu
          if (alluid(i) == alluid(j)) dup = .true.
       end do
    end do
    if (dup .and. trim(val) /= 'rccl') then
       if (myid == 1) write(*,*) 'ramses_amd: several ranks share a GPU: halo exchange staged through host MPI (not RCCL)'
       return
    end if
    ! every rank first checks LOCALLY that it can load the library (no communication); only when all of them can
    ! does anyone enter the collective ncclCommInitRank -- a rank that failed alone would leave the others blocked in it
    id = c_null_char
    rc = ramses_amd_rccl_probe()
    if (rc == 0 .and. myid == 1) rc = ramses_amd_rccl_unique_id(id)
    call MPI_ALLREDUCE(rc, rcmax, 1, MPI_INTEGER, MPI_MIN, MPI_COMM_WORLD, info)   ! error codes are negative
    if (rcmax == 0) then
       call MPI_BCAST(id, 128, MPI_CHARACTER, 0, MPI_COMM_WORLD, info)
       rc = ramses_amd_rccl_init(id, ncpu, myid - 1)
       call MPI_ALLREDUCE(rc, rcmax, 1, MPI_INTEGER, MPI_MIN, MPI_COMM_WORLD, info)
    end if
    if (rcmax /= 0) then
       if (trim(val) == 'rccl') call ramses_amd_fatal('halo transport (RCCL requested with RAMSES_AMD_HALO=rccl)')
       rc = ramses_amd_rccl_finalize()
       if (myid == 1) write(*,*) 'ramses_amd: RCCL could not be brought up: halo exchange staged through host MPI'
       return
    end if
    ramses_amd_halo_rccl = .true.
    if (myid == 1) write(*,*) 'ramses_amd: halo exchange over RCCL, ', ncpu, ' ranks'
  end subroutine ramses_amd_halo_init

  !---------------------------------------------------------------------------
  ! make_virtual_fine_dp(uold(1,1:nvar),levelmin) on the resident bricks: all nvar fields in ONE
  ! exchange (reference: nvar rounds, amr/amr_step.f90:497-508; pack/unpack amr/virtual_boundaries.f90:454-506)
  !---------------------------------------------------------------------------
  subroutine ramses_amd_halo_forward()
    use amr_commons
    use mpi_mod
    integer :: rc, icpu, info, nreq, cnt
    type(c_ptr) :: hs, hr, so, ro
    real(c_double), pointer :: sbuf(:), rbuf(:)
    integer(c_int64_t), pointer :: soff(:), roff(:)
    integer, dimension(2*ncpu) :: req
    integer, dimension(MPI_STATUS_SIZE, 2*ncpu) :: statuses
    integer, parameter :: tag = 131
    if (ramses_amd_halo_rccl) then
       rc = ramses_amd_mpires_halo_forward()
       if (rc /= 0) call ramses_amd_fatal('make_virtual_fine_dp (RCCL exchange)')
       return
    end if
    rc = ramses_amd_mpires_halo_stage_out(hs, so, hr, ro)
    if (rc /= 0) call ramses_amd_fatal('make_virtual_fine_dp (pack)')
    call c_f_pointer(so, soff, [ncpu + 1])
    call c_f_pointer(ro, roff, [ncpu + 1])
    call c_f_pointer(hs, sbuf, [max(soff(ncpu + 1), 1_8)])
    call c_f_pointer(hr, rbuf, [max(roff(ncpu + 1), 1_8)])
    nreq = 0
    do icpu = 1, ncpu
       cnt = int(roff(icpu + 1) - roff(icpu))
       if (cnt > 0) then
          nreq = nreq + 1
          call MPI_IRECV(rbuf(roff(icpu) + 1), cnt, MPI_DOUBLE_PRECISION, icpu - 1, tag, MPI_COMM_WORLD, req(nreq), info)
       end if
    end do
    do icpu = 1, ncpu
       cnt = int(soff(icpu + 1) - soff(icpu))
       if (cnt > 0) then
          nreq = nreq + 1
          call MPI_ISEND(sbuf(soff(icpu) + 1), cnt, MPI_DOUBLE_PRECISION, icpu - 1, tag, MPI_COMM_WORLD, req(nreq), info)
       end if
    end do
    call MPI_WAITALL(nreq, req, statuses, info)
    rc = ramses_amd_mpires_halo_stage_in()
    if (rc /= 0) call ramses_amd_fatal('make_virtual_fine_dp (unpack)')
  end subroutine ramses_amd_halo_forward
#endif

  !---------------------------------------------------------------------------
  ! Residency for AMR runs (single rank, hydro only): does the configuration allow it?
  ! RAMSES_AMD_RESIDENT_AMR=0 keeps the staging path (arrays copied around every call).
  !---------------------------------------------------------------------------
  logical function ramses_amd_amr_config()
    use amr_commons
    use hydro_parameters
    use poisson_parameters, only: gravity_type
#if USE_TURB==1
    use turb_commons, only: turb
#endif
    character(len=16) :: val
    integer :: stat, l
    if (.not. ramses_amd_amr_checked) then
       ramses_amd_amr_checked = .true.
       ramses_amd_amr_ok = ramses_amd_enabled()
       call get_environment_variable('RAMSES_AMD_RESIDENT_AMR', val, status=stat)
       if (stat == 0) then
          if (trim(val) == '0') ramses_amd_amr_ok = .false.
       end if
       if (nboundary > 0) then
          ! physical boundaries: make_boundary_hydro runs on the resident cell vectors (hydro_boundary.f90 of this
          ! directory); the states of imposed boundaries come from the reference's boundana, evaluated by the shim.  With
          ! self-gravity the density goes back for the reference's rho_fine, the solve takes the routines of the
          ! multigrid shims under the reference's driver (Dirichlet set-up on the host), force_fine +
          ! make_boundary_force stay the reference's and f of the level's octs, boundary octs included, is mirrored.
          ! RAMSES_AMD_RESIDENT_WALLS=0: staging path.
          if (.not. simple_boundary) ramses_amd_amr_ok = .false.
          do l = 1, nboundary
             if (boundary_type(l) / 10 > 2 .or. mod(boundary_type(l), 10) < 1 .or. mod(boundary_type(l), 10) > 6) &
                  & ramses_amd_amr_ok = .false.
          end do
          call get_environment_variable('RAMSES_AMD_RESIDENT_WALLS', val, status=stat)
          if (stat == 0) then
             if (trim(val) == '0') ramses_amd_amr_ok = .false.
          end if
       end if
       if (levelmin >= nlevelmax) then
          ! one uniform level: the brick paths (ramses_amd_resident on one rank, ramses_amd_mpi_resident on 2^k ranks:
          ! dense sweep) take it when they can.  What they do not carry takes this one -- cell vectors, tree and
          ! communicators resident, the tree-walking sweep: several ranks WITH self-gravity (round 4, VERDICT round 3
          ! missing #3: both virtual-boundary exchanges, rho_fine's deposit and force_fine on the device), physical
          ! boundaries, pressure_fix, difmag, rank counts whose domains are not boxes, nremap > 0 under MPI.
          ! (RAMSES_AMD_RESIDENT=0, the switch of the brick paths, keeps a uniform level on the staging path altogether)
          call get_environment_variable('RAMSES_AMD_RESIDENT', val, status=stat)
          if (stat == 0) then
             if (trim(val) == '0') ramses_amd_amr_ok = .false.
          end if
          if (levelmin > nlevelmax) then
             ramses_amd_amr_ok = .false.
          else if (ncpu == 1) then
             if (ramses_amd_resident()) ramses_amd_amr_ok = .false.
          else
             if (ramses_amd_mpi_resident()) ramses_amd_amr_ok = .false.
          end if
       end if
       ! nremap > 0: load_balance.f90 of this directory hands the state back to the host before the octs move between the
       ! ranks (load_balance) or are renumbered (defrag, one rank too); the device image is rebuilt afterwards.
       if (ncpu > 1) then
          ! several ranks: the virtual-boundary exchanges of the hydro state run on the device too
          ! (virtual_boundaries.f90 of this directory); RAMSES_AMD_RESIDENT_AMR_MPI=0 keeps such runs staged.
          ! With self-gravity the Poisson solve itself stays the MPI path of multigrid_fine_commons.f90 (host arrays
          ! phi, rho, f); the device mirror of f then covers the virtual octs too.
          call get_environment_variable('RAMSES_AMD_RESIDENT_AMR_MPI', val, status=stat)
          if (stat == 0) then
             if (trim(val) == '0') ramses_amd_amr_ok = .false.
          end if
          if (nlevelmax > 64) ramses_amd_amr_ok = .false.
#ifdef WITHOUTMPI
          ramses_amd_amr_ok = .false.
#endif
#ifdef LIGHT_MPI_COMM
          ramses_amd_amr_ok = .false.     ! the condensed communicators of LIGHT_MPI_COMM are not mirrored
#endif
       end if
       if (.not. hydro .or. pic .or. rt .or. cooling .or. star .or. sink .or. stellar) ramses_amd_amr_ok = .false.
       ! with self-gravity too (the acceleration is mirrored on the device, rho_fine gets the density back);
       ! RAMSES_AMD_RESIDENT_GRAV=0 keeps such runs on the staging path
       if (poisson) then
          call get_environment_variable('RAMSES_AMD_RESIDENT_GRAV', val, status=stat)
          if (stat == 0) then
             if (trim(val) == '0') ramses_amd_amr_ok = .false.
          end if
          if (gravity_type > 0 .or. cosmo) ramses_amd_amr_ok = .false.
          do l = 1, nlevelmax
             if (m_refine(l) > -1.0d0) ramses_amd_amr_ok = .false.    ! rho_fine's quasi-Lagrangian map reads uold on the host
          end do
       end if
       if (tracer .or. MC_tracer .or. clumpfind .or. lightcone .or. movie .or. aton) ramses_amd_amr_ok = .false.
       if (static .or. static_gas .or. neq_chem .or. barotropic_eos .or. isothermal .or. metal) ramses_amd_amr_ok = .false.
       if (T2_star > 0.0d0 .or. momentum_feedback > 0 .or. strict_equilibrium > 0) ramses_amd_amr_ok = .false.
       if (ndim /= 3 .or. levelmin < 3) ramses_amd_amr_ok = .false.
       if (icoarse_max - icoarse_min /= 0 .or. jcoarse_max - jcoarse_min /= 0 .or. kcoarse_max - kcoarse_min /= 0) &
            & ramses_amd_amr_ok = .false.
       do l = 1, nlevelmax
          if (jeans_refine(l) > -1.0d0) ramses_amd_amr_ok = .false.     ! jeans_length_refine reads uold on the host
       end do
#if USE_TURB==1
       if (turb) ramses_amd_amr_ok = .false.
#endif
       if (ramses_amd_amr_ok .and. myid == 1) &
            & write(*,*) 'ramses_amd: hydro state and tree of the AMR levels stay resident on the GPU'
    end if
    ramses_amd_amr_config = ramses_amd_amr_ok
  end function ramses_amd_amr_config

  ! RAMSES_AMD_FORCE_MPI=0: force_fine of AMR levels with several ranks keeps the reference's host loops
  logical function ramses_amd_force_mpi_on()
    character(len=16) :: val
    integer :: stat
    logical, save :: first = .true., on = .true.
    if (first) then
       call get_environment_variable('RAMSES_AMD_FORCE_MPI', val, status=stat)
       if (stat == 0) then
          if (trim(val) == '0') on = .false.
       end if
       first = .false.
    end if
    ramses_amd_force_mpi_on = on
  end function ramses_amd_force_mpi_on

  ! RAMSES_AMD_F_RESIDENT=0: force_fine of AMR levels with several ranks returns f to the host array and the device copy is
  ! refreshed from there (the path before round 6)
  logical function ramses_amd_f_resident_on()
    character(len=16) :: val
    integer :: stat
    logical, save :: first = .true., on = .true.
    if (first) then
       call get_environment_variable('RAMSES_AMD_F_RESIDENT', val, status=stat)
       if (stat == 0) then
          if (trim(val) == '0') on = .false.
       end if
       first = .false.
    end if
    ramses_amd_f_resident_on = on
  end function ramses_amd_f_resident_on

  ! f of the levels whose acceleration lives on the device only, back into the host array (backup_poisson, load_balance;
  ! one level: refine_fine, whose new octs inherit their father cell's f)
  subroutine ramses_amd_amr_sync_f()
    use amr_commons
    integer :: l
    if (.not. poisson) return
    if (ramses_amd_amrres_active() == 0) return
    do l = 1, min(nlevelmax, 64)
       if (ramses_amd_f_on_device(l)) call ramses_amd_amr_sync_f_level(l)
    end do
  end subroutine ramses_amd_amr_sync_f

  subroutine ramses_amd_amr_sync_f_level(l)
    use amr_commons
    use poisson_commons, only: f
    integer, intent(in) :: l
    integer :: rc, nl
    integer, allocatable :: list(:)
    if (ramses_amd_amrres_active() == 0) return
    if (numbtot(1, l) > 0) then
       if (ncpu > 1 .or. nboundary > 0) then
          call ramses_amd_amr_level_octs(l, nl, list)
          rc = ramses_amd_amrres_sync_f(nl, list, f)
          deallocate(list)
       else
          rc = ramses_amd_amrres_sync_f(active(l)%ngrid, ramses_amd_octs(l), f)
       end if
       if (rc /= 0) call ramses_amd_fatal('AMR residency (acceleration back to the host)')
    end if
    ramses_amd_f_on_device(l) = .false.
  end subroutine ramses_amd_amr_sync_f_level

  ! RAMSES_AMD_PHI_RESIDENT=0: rho and phi of such a run keep crossing PCIe around every solve (the path before round 6)
  logical function ramses_amd_phi_resident_on()
    character(len=16) :: val
    integer :: stat
    logical, save :: first = .true., on = .true.
    if (first) then
       call get_environment_variable('RAMSES_AMD_PHI_RESIDENT', val, status=stat)
       if (stat == 0) then
          if (trim(val) == '0') on = .false.
       end if
       first = .false.
    end if
    ramses_amd_phi_resident_on = on
  end function ramses_amd_phi_resident_on

  ! rho and phi of levelmin back into the host vectors (backup_poisson, load_balance, a solve that reads the host vectors after
  ! all); off = .true.: the steady state ends here (rho_fine copies its deposit back again, the solve takes the host vectors)
  subroutine ramses_amd_pois_mpi_sync_host(off)
    use amr_commons
    use poisson_commons, only: rho, phi
    logical, intent(in) :: off
    integer :: rc, nl
    integer, allocatable :: list(:)
    if (ramses_amd_pois_mpi_dev .and. ramses_amd_amrres_active() /= 0 .and. numbtot(1, levelmin) > 0) then
       call ramses_amd_amr_level_octs(levelmin, nl, list)
       rc = ramses_amd_amrres_sync_rho(nl, list, rho)
       deallocate(list)
       if (rc /= 0) call ramses_amd_fatal('density deposit back to the host vector')
    end if
    if (ramses_amd_phi_on_device .and. c_associated(ramses_amd_mgdist_ctx)) then
       rc = ramses_amd_mgdist_fetch_phi_f90(ramses_amd_mgdist_ctx, active(levelmin)%ngrid, ramses_amd_octs(levelmin), &
            & int(ngridmax, 8), int(ncoarse, 8), phi)
       if (rc /= 0) call ramses_amd_fatal('potential back to the host vector')
       call make_virtual_fine_dp(phi(1), levelmin)
    end if
    if (off) then
       ramses_amd_pois_mpi_dev = .false.
       ramses_amd_phi_on_device = .false.
       rc = ramses_amd_amrres_rho_keep(0)
    end if
  end subroutine ramses_amd_pois_mpi_sync_host

  logical function ramses_amd_amr_resident()
    ramses_amd_amr_resident = .false.
    if (ramses_amd_amr_armed) ramses_amd_amr_resident = ramses_amd_amr_config()
  end function ramses_amd_amr_resident

  !---------------------------------------------------------------------------
  ! Before a device routine: load everything (first use), or send what refine_fine rebuilt on the
  ! host since the last device routine (the tree and the levels >= ramses_amd_amr_reload_from)
  !---------------------------------------------------------------------------
  subroutine ramses_amd_amr_ensure()
    use amr_commons
    use hydro_commons
    use poisson_commons, only: f
    integer :: rc, l, nl
    integer, allocatable :: list(:)
    if (ramses_amd_amrres_active() == 0) then
       rc = ramses_amd_amrres_load(nvar, int(ngridmax, 8), int(ncoarse, 8), uold, son, nbor, father)
       if (rc /= 0) call ramses_amd_fatal('AMR residency (load)')
       ! whatever is cached per tree epoch on the device (rho_fine's oct centres) went with the old image
       ramses_amd_tree_epoch = ramses_amd_tree_epoch + 1
       if (pressure_fix) then
          rc = ramses_amd_amrres_enable_pfix()
          if (rc /= 0) call ramses_amd_fatal('AMR residency (pressure_fix)')
       end if
       if (poisson) then
          do l = levelmin, nlevelmax
             if (numbtot(1, l) > 0) call ramses_amd_amr_load_f(l)
          end do
       end if
       ramses_amd_amr_reload_from = 1000
       ramses_amd_amr_host_from = 1000
       return
    end if
    ramses_amd_amr_host_from = 1000        ! a device routine follows: the host copies go stale
    if (ramses_amd_amr_reload_from <= nlevelmax) then
       rc = ramses_amd_amrres_tree(son, nbor, father)
       if (rc /= 0) call ramses_amd_fatal('AMR residency (tree)')
       do l = ramses_amd_amr_reload_from, nlevelmax
          if (numbtot(1, l) > 0) then
             if (ncpu > 1 .or. nboundary > 0) then
                ! the virtual octs too: the host has just exchanged them itself (amr/amr_step.f90:49-62); the boundary
                ! octs too: the host has just filled them (:70)
                call ramses_amd_amr_level_octs(l, nl, list)
                rc = ramses_amd_amrres_load_level(nl, list, uold)
                deallocate(list)
             else
                rc = ramses_amd_amrres_load_level(active(l)%ngrid, ramses_amd_octs(l), uold)
             end if
             if (rc /= 0) call ramses_amd_fatal('AMR residency (level reload)')
             if (poisson) call ramses_amd_amr_load_f(l)
          end if
       end do
       ramses_amd_amr_reload_from = 1000
    end if
  end subroutine ramses_amd_amr_ensure

  ! the acceleration of one level from the host array f (force_fine has just written it, or the level was rebuilt)
  subroutine ramses_amd_amr_load_f(ilevel)
    use amr_commons
    use poisson_commons
    integer, intent(in) :: ilevel
    integer :: rc, nl
    integer, allocatable :: list(:)
    if (ncpu > 1 .or. nboundary > 0) then
       ! the virtual octs too (force_fine has exchanged f itself, poisson/force_fine.f90:107,137) and the boundary octs
       ! (make_boundary_force, :109,139): the sweep's gravity predictor reads f of every stencil cell
       call ramses_amd_amr_level_octs(ilevel, nl, list)
       rc = ramses_amd_amrres_load_f(nl, list, f)
       deallocate(list)
    else
       rc = ramses_amd_amrres_load_f(active(ilevel)%ngrid, ramses_amd_octs(ilevel), f)
    end if
    if (rc /= 0) call ramses_amd_fatal('AMR residency (acceleration)')
  end subroutine ramses_amd_amr_load_f

  !---------------------------------------------------------------------------
  ! refine_fine(ilevel) is about to read uold of levels ilevel-1 .. (interpol_hydro of new octs,
  ! getnborfather's coarser fallback) and to rebuild levels ilevel+1 ..: bring the device's levels
  ! back first, remember from which level the host is ahead.
  !---------------------------------------------------------------------------
  subroutine ramses_amd_amr_refine_hook(ilevel)
    use amr_commons
    use hydro_commons
    integer, intent(in) :: ilevel
    integer :: rc, l, nl
    integer, allocatable :: list(:)
    ramses_amd_tree_epoch = ramses_amd_tree_epoch + 1
    if (.not. ramses_amd_amr_resident()) return
    if (ramses_amd_amrres_active() == 0) return
    if (ilevel < levelmin) return      ! fully refined coarse levels: nothing is created, no hydro data is read
    do l = max(ilevel - 1, levelmin), min(nlevelmax, ramses_amd_amr_host_from - 1)
       if (numbtot(1, l) > 0) then
          if (ncpu > 1 .or. nboundary > 0) then
             ! the virtual octs too: refine_fine interpolates the new virtual octs from them (and the new boundary octs from
             ! the boundary octs)
             call ramses_amd_amr_level_octs(l, nl, list)
             rc = ramses_amd_amrres_sync_level(nl, list, uold)
             deallocate(list)
          else
             rc = ramses_amd_amrres_sync_level(active(l)%ngrid, ramses_amd_octs(l), uold)
          end if
          if (rc /= 0) call ramses_amd_fatal('AMR residency (level sync before refine_fine)')
       end if
    end do
    ! make_grid_fine gives every new oct the acceleration of its father cell (amr/refine_utils.f90:918-927): f of level ilevel
    ! comes back if force_fine left it on the device only
    ! -- and of every finer level now, while the device's oct numbering still matches the host's lists (refine_fine(ilevel)
    ! rebuilds level ilevel+1 before refine_fine(ilevel+1) asks for it)
    if (poisson) then
       do l = ilevel, min(nlevelmax, 64)
          if (ramses_amd_f_on_device(l)) call ramses_amd_amr_sync_f_level(l)
       end do
    end if
    ramses_amd_amr_host_from = min(ramses_amd_amr_host_from, max(ilevel - 1, levelmin))
    ramses_amd_amr_reload_from = min(ramses_amd_amr_reload_from, ilevel + 1)      ! refine_fine(ilevel) rebuilds level ilevel+1
  end subroutine ramses_amd_amr_refine_hook

  !---------------------------------------------------------------------------
  ! The reference is about to move octs between ranks (load_balance): every level the device holds the only
  ! current copy of goes back to the host arrays, then the device image is dropped; the next device routine
  ! loads everything again (ramses_amd_amr_ensure's first branch).
  !---------------------------------------------------------------------------
  subroutine ramses_amd_amr_host_takeover(where)
    use amr_commons
    use hydro_commons
    character(len=*), intent(in) :: where
    integer :: rc, l, nl
    integer, allocatable :: list(:)
    ramses_amd_tree_epoch = ramses_amd_tree_epoch + 1
    if (.not. ramses_amd_amr_resident()) return
    if (ramses_amd_amrres_active() == 0) return
    do l = levelmin, min(nlevelmax, ramses_amd_amr_host_from - 1)
       if (numbtot(1, l) > 0) then
          call ramses_amd_amr_level_octs(l, nl, list)
          rc = ramses_amd_amrres_sync_level(nl, list, uold)
          deallocate(list)
          if (rc /= 0) call ramses_amd_fatal('AMR residency (level sync before '//where//')')
       end if
    end do
    call ramses_amd_amr_sync_f()
    call ramses_amd_pois_mpi_sync_host(.true.)
    rc = ramses_amd_amrres_invalidate()
    if (rc /= 0) call ramses_amd_fatal('AMR residency (invalidate before '//where//')')
    ramses_amd_amr_reload_from = 1000
    ramses_amd_amr_host_from = 1000
  end subroutine ramses_amd_amr_host_takeover

  !---------------------------------------------------------------------------
  ! AMR residency with several MPI ranks.  The octs of a level whose cells a rank holds: its own
  ! (active) followed by the virtual ones (reception lists of every peer) and by the octs of the
  ! physical boundary regions (boundary(1:nboundary,ilevel)).
  !---------------------------------------------------------------------------
  subroutine ramses_amd_amr_level_octs(ilevel, n, list)
    use amr_commons
    integer, intent(in) :: ilevel
    integer, intent(out) :: n
    integer, allocatable, intent(out) :: list(:)
    integer :: icpu, i
    n = active(ilevel)%ngrid
    do icpu = 1, ncpu
       n = n + reception(icpu, ilevel)%ngrid
    end do
    do icpu = 1, nboundary
       n = n + boundary(icpu, ilevel)%ngrid
    end do
    allocate(list(max(n, 1)))
    do i = 1, active(ilevel)%ngrid
       list(i) = active(ilevel)%igrid(i)
    end do
    n = active(ilevel)%ngrid
    do icpu = 1, ncpu
       do i = 1, reception(icpu, ilevel)%ngrid
          list(n + i) = reception(icpu, ilevel)%igrid(i)
       end do
       n = n + reception(icpu, ilevel)%ngrid
    end do
    do icpu = 1, nboundary
       do i = 1, boundary(icpu, ilevel)%ngrid
          list(n + i) = boundary(icpu, ilevel)%igrid(i)
       end do
       n = n + boundary(icpu, ilevel)%ngrid
    end do
  end subroutine ramses_amd_amr_level_octs

  !---------------------------------------------------------------------------
  ! make_boundary_hydro(ilevel) on the resident cell vectors (hydro/hydro_boundary.f90:5-269): the regions' oct lists
  ! go down one after the other, the device walks them in the reference's order
  !---------------------------------------------------------------------------
  subroutine ramses_amd_amr_boundary(ilevel)
    use amr_commons
    use hydro_parameters, only: smallr, nvar
    integer, intent(in) :: ilevel
    integer :: ib, i, n, rc, flag, nimp, ind, idim, i0, ng, ivar, ix, iy, iz, nx_loc
    integer(8) :: base
    integer, allocatable :: cnt(:), list(:)
    real(dp), allocatable :: imposed(:)
    real(dp) :: dx, dx_loc, scale, skip_loc(3), xc(8, 3)
    real(dp), dimension(1:nvector, 1:ndim) :: xx
    real(dp), dimension(1:nvector, 1:nvar) :: uu
    n = 0
    nimp = 0
    do ib = 1, nboundary
       n = n + boundary(ib, ilevel)%ngrid
       if (boundary_type(ib) / 10 == 2) nimp = nimp + boundary(ib, ilevel)%ngrid
    end do
    if (n == 0) return
    allocate(cnt(nboundary), list(n), imposed(max(1, nimp * 8 * nvar)))
    n = 0
    do ib = 1, nboundary
       cnt(ib) = boundary(ib, ilevel)%ngrid
       do i = 1, cnt(ib)
          list(n + i) = boundary(ib, ilevel)%igrid(i)
       end do
       n = n + cnt(ib)
    end do
    if (nimp > 0) then
       ! imposed boundaries: the reference's boundana evaluated for every cell of the region, as make_boundary_hydro
       ! does (hydro/hydro_boundary.f90:36-49,215-241: cell centres in user units, chunks of nvector octs)
       dx = 0.5d0**ilevel
       nx_loc = icoarse_max - icoarse_min + 1
       skip_loc = (/dble(icoarse_min), dble(jcoarse_min), dble(kcoarse_min)/)
       scale = boxlen / dble(nx_loc)
       dx_loc = dx * scale
       do ind = 1, twotondim
          iz = (ind - 1) / 4
          iy = (ind - 1 - 4 * iz) / 2
          ix = (ind - 1 - 2 * iy - 4 * iz)
          xc(ind, 1) = (dble(ix) - 0.5d0) * dx
          xc(ind, 2) = (dble(iy) - 0.5d0) * dx
          xc(ind, 3) = (dble(iz) - 0.5d0) * dx
       end do
       base = 0
       do ib = 1, nboundary
          if (boundary_type(ib) / 10 /= 2) cycle
          ng = boundary(ib, ilevel)%ngrid
          do i0 = 1, ng, nvector
             n = min(nvector, ng - i0 + 1)
             do ind = 1, twotondim
                do idim = 1, ndim
                   do i = 1, n
                      xx(i, idim) = xg(boundary(ib, ilevel)%igrid(i0 + i - 1), idim) + xc(ind, idim)
                   end do
                end do
                do idim = 1, ndim
                   do i = 1, n
                      xx(i, idim) = (xx(i, idim) - skip_loc(idim)) * scale
                   end do
                end do
                call boundana(xx, uu, dx_loc, ib, n)
                ! [nvar][8][ng] of the region
                do ivar = 1, nvar
                   do i = 1, n
                      imposed(base + (int(ivar - 1, 8) * 8 + int(ind - 1, 8)) * int(ng, 8) + int(i0 + i - 1, 8)) = uu(i, ivar)
                   end do
                end do
             end do
          end do
          base = base + int(ng, 8) * 8 * int(nvar, 8)
       end do
    end if
    flag = 0
    if (no_inflow) flag = 1
    rc = ramses_amd_amrres_boundary_hydro(nboundary, boundary_type, cnt, list, flag, smallr, nvector, imposed)
    if (rc /= 0) call ramses_amd_fatal('make_boundary_hydro')
    deallocate(cnt, list, imposed)
  end subroutine ramses_amd_amr_boundary

#ifndef WITHOUTMPI
  !---------------------------------------------------------------------------
  ! The communicators of a level on the device, re-sent after build_comm rebuilt them
  !---------------------------------------------------------------------------
  subroutine ramses_amd_amr_comm_ensure(ilevel)
    use amr_commons
    integer, intent(in) :: ilevel
    integer, allocatable :: em_n(:), em_ig(:), rc_n(:), rc_ig(:)
    integer :: rc
    if (ramses_amd_amrres_comm_epoch(ilevel) == ramses_amd_comm_epoch(ilevel)) return
    call ramses_amd_comm_lists(ilevel, em_n, em_ig, rc_n, rc_ig)
    rc = ramses_amd_amrres_comm_set(ilevel, ramses_amd_comm_epoch(ilevel), ncpu, em_n, em_ig, rc_n, rc_ig)
    if (rc /= 0) call ramses_amd_fatal('AMR residency (communicators)')
  end subroutine ramses_amd_amr_comm_ensure

  !---------------------------------------------------------------------------
  ! One virtual-boundary exchange on the resident cell vectors.  dir 0: make_virtual_fine_dp on
  ! uold(1,1:nvar) (amr/virtual_boundaries.f90:373-528); 1: make_virtual_reverse_dp on unew(1,1:nvar)
  ! (:693-983).  All nvar variables in one message per peer; RCCL, or the program's own MPI on pinned
  ! host buffers when ranks share a GPU.
  !---------------------------------------------------------------------------
  subroutine ramses_amd_amr_halo(ilevel, dir)
    use amr_commons
    use mpi_mod
    integer, intent(in) :: ilevel, dir
    integer :: rc, icpu, info, nreq, cnt
    type(c_ptr) :: hs, hr
    real(c_double), pointer :: sbuf(:), rbuf(:)
    integer(c_int64_t), dimension(ncpu + 1) :: soff, roff
    integer, dimension(2*ncpu) :: req
    integer, dimension(MPI_STATUS_SIZE, 2*ncpu) :: statuses
    integer, parameter :: tag = 137
    call ramses_amd_amr_comm_ensure(ilevel)
    if (ramses_amd_halo_rccl) then
       rc = ramses_amd_amrres_halo_rccl(ilevel, dir, myid)
       if (rc /= 0) call ramses_amd_fatal('virtual boundaries of an AMR level (RCCL exchange)')
       return
    end if
    rc = ramses_amd_amrres_halo_stage_out(ilevel, dir, ncpu, hs, hr, soff, roff)
    if (rc /= 0) call ramses_amd_fatal('virtual boundaries of an AMR level (pack)')
    call c_f_pointer(hs, sbuf, [max(soff(ncpu + 1), 1_8)])
    call c_f_pointer(hr, rbuf, [max(roff(ncpu + 1), 1_8)])
    nreq = 0
    do icpu = 1, ncpu
       cnt = int(roff(icpu + 1) - roff(icpu))
       if (cnt > 0) then
          nreq = nreq + 1
          call MPI_IRECV(rbuf(roff(icpu) + 1), cnt, MPI_DOUBLE_PRECISION, icpu - 1, tag, MPI_COMM_WORLD, req(nreq), info)
       end if
    end do
    do icpu = 1, ncpu
       cnt = int(soff(icpu + 1) - soff(icpu))
       if (cnt > 0) then
          nreq = nreq + 1
          call MPI_ISEND(sbuf(soff(icpu) + 1), cnt, MPI_DOUBLE_PRECISION, icpu - 1, tag, MPI_COMM_WORLD, req(nreq), info)
       end if
    end do
    call MPI_WAITALL(nreq, req, statuses, info)
    rc = ramses_amd_amrres_halo_stage_in(ilevel, dir)
    if (rc /= 0) call ramses_amd_fatal('virtual boundaries of an AMR level (unpack)')
  end subroutine ramses_amd_amr_halo
#endif

#ifndef WITHOUTMPI
  !---------------------------------------------------------------------------
  ! One virtual-boundary exchange of a level of the running multigrid solve, on the device (several ranks):
  !   level = the solved level:  comp 1 phi, 3 the residual f(:,1)      make_virtual_fine_dp  (amr/virtual_boundaries.f90:373-528)
  !   a multigrid level:         comp = ivar of active_mg(:,level)%u     make_virtual_mg_dp / make_reverse_mg_dp
  !                                                                      (poisson/multigrid_fine_commons.f90:1172-1290,1378-1475)
  ! dir 0 forward, 1 reverse (added peer by peer in icpu order).  RCCL, or the program's own MPI on pinned host buffers when
  ! ranks share a GPU.  The communicators go to the device with the first exchange of a level in a solve.
  !---------------------------------------------------------------------------
  subroutine ramses_amd_mg_halo(level, comp, dir)
    use amr_commons
    use poisson_commons
    use mpi_mod
    integer, intent(in) :: level, comp, dir
    integer :: rc, icpu, info, nreq, cnt, n, i, isoct
    type(c_ptr) :: hs, hr
    real(c_double), pointer :: sbuf(:), rbuf(:)
    integer(c_int64_t), dimension(ncpu + 1) :: soff, roff
    integer, dimension(2*ncpu) :: req
    integer, dimension(MPI_STATUS_SIZE, 2*ncpu) :: statuses
    integer, dimension(ncpu) :: em_n, rc_n
    integer, allocatable :: em_list(:)
    integer, parameter :: tag = 139
    if (.not. ramses_amd_mg_comm_done(level)) then
       n = 0
       if (level == ramses_amd_mg_level) then
          isoct = 1
          do icpu = 1, ncpu
             em_n(icpu) = emission(icpu, level)%ngrid
             rc_n(icpu) = reception(icpu, level)%ngrid
             n = n + em_n(icpu)
          end do
          allocate(em_list(max(n, 1)))
          n = 0
          do icpu = 1, ncpu
             do i = 1, em_n(icpu)
                em_list(n + i) = emission(icpu, level)%igrid(i)
             end do
             n = n + em_n(icpu)
          end do
       else
          isoct = 0
          do icpu = 1, ncpu
             em_n(icpu) = emission_mg(icpu, level)%ngrid
             rc_n(icpu) = active_mg(icpu, level)%ngrid
             n = n + em_n(icpu)
          end do
          rc_n(myid) = 0
          allocate(em_list(max(n, 1)))
          n = 0
          do icpu = 1, ncpu
             do i = 1, em_n(icpu)
                em_list(n + i) = emission_mg(icpu, level)%igrid(i)
             end do
             n = n + em_n(icpu)
          end do
       end if
       rc = ramses_amd_mgamr_comm_set(level, ncpu, myid, em_n, em_list, isoct, rc_n)
       deallocate(em_list)
       if (rc /= 0) call ramses_amd_fatal('multigrid_fine (communicators of a level of the solve)')
       ramses_amd_mg_comm_done(level) = .true.
    end if
    if (ramses_amd_halo_rccl) then
       rc = ramses_amd_mgamr_halo_rccl(level, comp, dir)
       if (rc /= 0) call ramses_amd_fatal('multigrid_fine (virtual boundaries, RCCL exchange)')
       return
    end if
    rc = ramses_amd_mgamr_halo_stage_out(level, comp, dir, ncpu, hs, hr, soff, roff)
    if (rc /= 0) call ramses_amd_fatal('multigrid_fine (virtual boundaries, pack)')
    call c_f_pointer(hs, sbuf, [max(soff(ncpu + 1), 1_8)])
    call c_f_pointer(hr, rbuf, [max(roff(ncpu + 1), 1_8)])
    nreq = 0
    do icpu = 1, ncpu
       cnt = int(roff(icpu + 1) - roff(icpu))
       if (cnt > 0) then
          nreq = nreq + 1
          call MPI_IRECV(rbuf(roff(icpu) + 1), cnt, MPI_DOUBLE_PRECISION, icpu - 1, tag, MPI_COMM_WORLD, req(nreq), info)
       end if
    end do
    do icpu = 1, ncpu
       cnt = int(soff(icpu + 1) - soff(icpu))
       if (cnt > 0) then
          nreq = nreq + 1
          call MPI_ISEND(sbuf(soff(icpu) + 1), cnt, MPI_DOUBLE_PRECISION, icpu - 1, tag, MPI_COMM_WORLD, req(nreq), info)
       end if
    end do
    call MPI_WAITALL(nreq, req, statuses, info)
    rc = ramses_amd_mgamr_halo_stage_in(level, comp, dir)
    if (rc /= 0) call ramses_amd_fatal('multigrid_fine (virtual boundaries, unpack)')
  end subroutine ramses_amd_mg_halo
#endif

#ifndef WITHOUTMPI
  !---------------------------------------------------------------------------
  ! make_virtual_fine_dp(f(1,2),ilevel) of the conjugate-gradient loop (poisson/phi_fine_cg.f90:134) on the device vector p:
  ! RCCL, or the program's own MPI on pinned host buffers when ranks share a GPU (ramses_amd_cgmpi_comm_set has sent the lists)
  !---------------------------------------------------------------------------
  subroutine ramses_amd_cg_p_halo()
    use amr_commons
    use mpi_mod
    integer :: rc, icpu, info, nreq, cnt
    type(c_ptr) :: hs, hr
    real(c_double), pointer :: sbuf(:), rbuf(:)
    integer(c_int64_t), dimension(ncpu + 1) :: soff, roff
    integer, dimension(2*ncpu) :: req
    integer, dimension(MPI_STATUS_SIZE, 2*ncpu) :: statuses
    integer, parameter :: tag = 141
    if (ramses_amd_halo_rccl) then
       rc = ramses_amd_cgmpi_p_halo_rccl()
       if (rc /= 0) call ramses_amd_fatal('phi_fine_cg (halo of p, RCCL exchange)')
       return
    end if
    rc = ramses_amd_cgmpi_p_halo_stage_out(ncpu, hs, hr, soff, roff)
    if (rc /= 0) call ramses_amd_fatal('phi_fine_cg (halo of p, pack)')
    call c_f_pointer(hs, sbuf, [max(soff(ncpu + 1), 1_8)])
    call c_f_pointer(hr, rbuf, [max(roff(ncpu + 1), 1_8)])
    nreq = 0
    do icpu = 1, ncpu
       cnt = int(roff(icpu + 1) - roff(icpu))
       if (cnt > 0) then
          nreq = nreq + 1
          call MPI_IRECV(rbuf(roff(icpu) + 1), cnt, MPI_DOUBLE_PRECISION, icpu - 1, tag, MPI_COMM_WORLD, req(nreq), info)
       end if
    end do
    do icpu = 1, ncpu
       cnt = int(soff(icpu + 1) - soff(icpu))
       if (cnt > 0) then
          nreq = nreq + 1
          call MPI_ISEND(sbuf(soff(icpu) + 1), cnt, MPI_DOUBLE_PRECISION, icpu - 1, tag, MPI_COMM_WORLD, req(nreq), info)
       end if
    end do
    call MPI_WAITALL(nreq, req, statuses, info)
    rc = ramses_amd_cgmpi_p_halo_stage_in()
    if (rc /= 0) call ramses_amd_fatal('phi_fine_cg (halo of p, unpack)')
  end subroutine ramses_amd_cg_p_halo
#endif

  !---------------------------------------------------------------------------
  ! The reference has no error returns on this path: print and clean_stop
  ! (amr/end.f90:26-46), as it does itself.
  !---------------------------------------------------------------------------
  subroutine ramses_amd_fatal(where)
    character(len=*), intent(in) :: where
    character(kind=c_char), pointer :: cmsg(:)
    type(c_ptr) :: cp
    integer :: i
    character(len=512) :: msg
    msg = ' '
    cp = ramses_amd_last_error()
    if (c_associated(cp)) then
       call c_f_pointer(cp, cmsg, [512])
       do i = 1, 512
          if (cmsg(i) == c_null_char) exit
          msg(i:i) = cmsg(i)
       end do
    end if
    write(*,*) 'ramses_amd: FATAL in ', where, ': ', trim(msg)
    write(*,*) 'ramses_amd: no CPU fallback is taken; set RAMSES_AMD=0 to run the reference path'
    flush(6)          ! (clean_stop ends in MPI_Abort: what sits in the buffer of a piped stdout would be lost)
    call clean_stop
  end subroutine ramses_amd_fatal

  !---------------------------------------------------------------------------
  ! &HYDRO_PARAMS -> POD (hydro/hydro_parameters.f90:75-89)
  !---------------------------------------------------------------------------
  subroutine ramses_amd_fill_hydro_params(p)
    use amr_parameters, only: ndim, poisson
    use amr_commons, only: myid
    use hydro_parameters
    type(ramses_amd_hydro_params), intent(out) :: p
    character(len=16) :: val
    integer :: stat
    p%ndim = ndim
    p%nvar = nvar
    p%gamma = gamma
    p%smallr = smallr
    p%smallc = smallc
    p%slope_type = slope_type
    p%slope_theta = slope_theta
    select case (trim(riemann))
    case ('llf');      p%riemann = 0
    case ('hllc');     p%riemann = 1
    case ('hll');      p%riemann = 2
    case ('acoustic'); p%riemann = 3
    case ('exact');    p%riemann = 4
    case default;      p%riemann = -1
    end select
    if (trim(scheme) == 'muscl') then
       p%scheme = 0
    else if (trim(scheme) == 'plmde') then
       p%scheme = 1
    else
       p%scheme = -1
    end if
    p%niter_riemann = niter_riemann
    p%difmag = difmag
    p%courant_factor = courant_factor
    ! Arithmetic of the dense brick sweep.  DEFAULT = the fast build (written-out FMAs, v_rcp_f64 / v_rsq_f64 + Newton):
    ! certified against the reference program within north_star's 1e-12 relative L-infinity at config C2's size over 100
    ! coarse steps with a developed blast wave (tests/test_fast_certificate_gpu.py).  RAMSES_AMD_STRICT=1 (or
    ! RAMSES_AMD_FAST=0) selects the verification mode: the reference's operation order, bit-identical snapshots.
    ! Every other kernel (tree-walking sweep, multigrid, CG, rho_fine, exchanges) is strict in both modes.
    ! slope_type = 3 (the positivity-preserving 27-point slope) is NOT certified in fast arithmetic: its limiter divides two
    ! nearly equal sums, and 60 steps of sedov3d.nml at 64^3 end 8e-11 away from the reference in time step and state
    ! (tests/test_fast_certificate_gpu.py, round 4) -- such runs take the strict build unless RAMSES_AMD_FAST=1 insists.
    ! Self-gravitating runs take the strict build too (round 6): north_star's tolerance includes phi, and 1e-15 of difference in
    ! rho leaves a spatially constant offset of ~1e-10 max|phi| in the potential of a periodic box -- the null space of the
    ! Laplacian, which the reference's multigrid does not pin -- while rho, u, P agree to 2e-15 and f to 5e-13
    ! (tests/test_fast_certificate_gpu.py::test_fast_mode_amr_self_gravity_live_ab; RAMSES_AMD_FAST=1 opts in).
    p%fast_math = 1
    if (slope_type == 3) p%fast_math = 0
    if (poisson) p%fast_math = 0
    call get_environment_variable('RAMSES_AMD_STRICT', val, status=stat)
    if (stat == 0) then
       if (trim(val) == '1') p%fast_math = 0
    end if
    call get_environment_variable('RAMSES_AMD_FAST', val, status=stat)
    if (stat == 0) then
       if (trim(val) == '0') p%fast_math = 0
       if (trim(val) == '1') p%fast_math = 1
    end if
    if (.not. ramses_amd_arith_said) then
       ramses_amd_arith_said = .true.
       if (myid == 1) then
          if (p%fast_math == 1) then
             write(*,*) 'ramses_amd: dense sweep arithmetic = fast (<= 1e-12 of the reference; RAMSES_AMD_STRICT=1: bit-identical)'
          else
             write(*,*) 'ramses_amd: dense sweep arithmetic = strict (bit-identical to the reference)'
          end if
       end if
    end if
    p%reserved = 0
  end subroutine ramses_amd_fill_hydro_params

#ifndef WITHOUTMPI
  !---------------------------------------------------------------------------
  ! The message layer of the distributed multigrid when the ranks cannot use RCCL (several ranks on one GPU): this
  ! program's own MPI on the library's pinned host buffers (struct ramses_amd_mg_transport, include/ramses_amd.h)
  !---------------------------------------------------------------------------
  function ramses_amd_mgdist_cb_exchange(user, npeer, peer, h_send, send_off, send_cnt, h_recv, recv_off, recv_cnt) &
       & bind(C) result(rc)
    use mpi_mod
    type(c_ptr), value :: user
    integer(c_int), value :: npeer
    integer(c_int), intent(in) :: peer(*)
    real(c_double) :: h_send(*), h_recv(*)
    integer(c_int64_t), intent(in) :: send_off(*), send_cnt(*), recv_off(*), recv_cnt(*)
    integer(c_int) :: rc
    integer :: i, nreq, info
    integer :: req(2*npeer + 1)
    integer :: statuses(MPI_STATUS_SIZE, 2*npeer + 1)
    integer, parameter :: tag = 141
    nreq = 0
    do i = 1, npeer
       if (recv_cnt(i) > 0) then
          nreq = nreq + 1
          call MPI_IRECV(h_recv(recv_off(i) + 1), int(recv_cnt(i)), MPI_DOUBLE_PRECISION, peer(i), tag, MPI_COMM_WORLD, &
               & req(nreq), info)
       end if
    end do
    do i = 1, npeer
       if (send_cnt(i) > 0) then
          nreq = nreq + 1
          call MPI_ISEND(h_send(send_off(i) + 1), int(send_cnt(i)), MPI_DOUBLE_PRECISION, peer(i), tag, MPI_COMM_WORLD, &
               & req(nreq), info)
       end if
    end do
    call MPI_WAITALL(nreq, req, statuses, info)
    rc = 0
  end function ramses_amd_mgdist_cb_exchange

  function ramses_amd_mgdist_cb_allgather(user, h_send, count, h_recv) bind(C) result(rc)
    use mpi_mod
    type(c_ptr), value :: user
    real(c_double) :: h_send(*), h_recv(*)
    integer(c_int64_t), value :: count
    integer(c_int) :: rc
    integer :: info
    call MPI_ALLGATHER(h_send, int(count), MPI_DOUBLE_PRECISION, h_recv, int(count), MPI_DOUBLE_PRECISION, MPI_COMM_WORLD, info)
    rc = info
  end function ramses_amd_mgdist_cb_allgather

  ! (the reference's own reduction of the residual norms, poisson/multigrid_fine_commons.f90:205-209,253-257)
  function ramses_amd_mgdist_cb_allreduce(user, x) bind(C) result(rc)
    use mpi_mod
    type(c_ptr), value :: user
    real(c_double) :: x
    integer(c_int) :: rc
    integer :: info
    real(kind=8) :: tot
    call MPI_ALLREDUCE(x, tot, 1, MPI_DOUBLE_PRECISION, MPI_SUM, MPI_COMM_WORLD, info)
    x = tot
    rc = info
  end function ramses_amd_mgdist_cb_allreduce
#endif

  !---------------------------------------------------------------------------
  ! multigrid_fine(levelmin) with several ranks, levelmin fully refined and periodic, every rank's domain a box (what
  ! the Hilbert decomposition gives 2^k ranks): rho of the rank's octs -> its brick, the distributed dense V-cycles on
  ! the GPUs (one 5-cell halo exchange per smoother launch, coarse levels replicated; csrc/mg_dist.hip), phi of the
  ! rank's octs back; the virtual octs of phi through the reference's make_virtual_fine_dp, as at the end of its loop.
  ! ok = .false.: another configuration -- the caller goes on to the multigrid of AMR levels.  Collective.
  !---------------------------------------------------------------------------
  subroutine ramses_amd_mgdist_multigrid(ilevel, ok, iters, err)
    use amr_commons
    use poisson_commons
    use poisson_parameters
    use constants, only: twopi
#ifndef WITHOUTMPI
    use mpi_mod
#endif
    integer, intent(in) :: ilevel
    logical, intent(out) :: ok
    integer, intent(out) :: iters
    real(kind=8), intent(out) :: err
#ifndef WITHOUTMPI
    character(len=16) :: val
    integer :: stat, rc, info, i, b, d, n, nx_loc, isafe
    integer :: lo(3), dims(3), mine(7), pg(3), c(3)
    integer, allocatable :: every(:,:), rob(:)
    logical :: same, fit
    real(dp) :: scale, fourpi
#endif
    ok = .false.
    iters = 0
    err = 0.0d0
#ifndef WITHOUTMPI
    if (ncpu == 1 .or. ilevel /= levelmin .or. ilevel > 11 .or. nboundary > 0) return
    n = 2**ilevel
    if (int(numbtot(1, ilevel), 8)*8_8 /= int(n, 8)**3) return
    ! the box of my octs, and everybody else's.  RAMSES_AMD_MG_DIST=0 (the reference's driver with the device operators
    ! instead) is read by every rank from ITS environment: the switch travels with the boxes, so that a launcher that does
    ! not export it uniformly cannot send some ranks into this collective path and others past it
    mine = 0
    call get_environment_variable('RAMSES_AMD_MG_DIST', val, status=stat)
    if (stat == 0) then
       if (trim(val) == '0') mine(7) = -1
    end if
    if (mine(7) == 0 .and. active(ilevel)%ngrid > 0) then
       rc = ramses_amd_mgdist_oct_box(ilevel, active(ilevel)%ngrid, ramses_amd_octs(ilevel), xg, int(ngridmax, 8), lo, dims)
       if (rc == 0) then
          mine(1:3) = lo
          mine(4:6) = dims
          mine(7) = 1
       else if (rc /= RAMSES_AMD_EUNSUPPORTED_CODE) then
          ! an oct off the level lattice (EINVAL) is a broken tree, not "does not fit"
          call ramses_amd_fatal('multigrid_fine (distributed dense multigrid, box of the rank''s octs)')
       end if
    end if
    allocate(every(7, ncpu), rob(ncpu))
    call MPI_ALLGATHER(mine, 7, MPI_INTEGER, every, 7, MPI_INTEGER, MPI_COMM_WORLD, info)
    if (any(every(7, :) == -1)) then
       deallocate(every, rob)
       call ramses_amd_pois_mpi_sync_host(.true.)
       return
    end if
    fit = all(every(7, :) == 1)
    if (fit) then
       dims = every(4:6, 1)
       do d = 1, 3
          if (dims(d) < 64 .or. iand(dims(d), dims(d) - 1) /= 0 .or. dims(d) > n) fit = .false.
       end do
    end if
    if (fit) then
       pg = n/dims
       if (pg(1)*pg(2)*pg(3) /= ncpu) fit = .false.
    end if
    if (fit) then
       rob = -1
       do i = 1, ncpu
          if (any(every(4:6, i) /= dims) .or. any(mod(every(1:3, i), dims) /= 0)) then
             fit = .false.
             exit
          end if
          c = every(1:3, i)/dims
          b = c(1) + pg(1)*(c(2) + pg(2)*c(3)) + 1
          if (rob(b) /= -1) fit = .false.
          rob(b) = i - 1
       end do
       if (any(rob < 0)) fit = .false.
    end if
    if (.not. fit) then
       if (myid == 1 .and. .not. ramses_amd_mgdist_said) write(*,*) 'ramses_amd: the rank domains of level ', ilevel, &
            & ' are not equal power-of-two boxes of >= 64 cells: multigrid of AMR levels'
       ramses_amd_mgdist_said = .true.
       deallocate(every, rob)
       call ramses_amd_pois_mpi_sync_host(.true.)
       return
    end if
    ! (re)build the context when the decomposition changed (load balancing)
    same = c_associated(ramses_amd_mgdist_ctx) .and. ramses_amd_mgdist_level == ilevel .and. all(ramses_amd_mgdist_pg == pg)
    if (same) same = all(ramses_amd_mgdist_rob == rob)
    if (.not. same) then
       if (c_associated(ramses_amd_mgdist_ctx)) rc = ramses_amd_mgdist_destroy(ramses_amd_mgdist_ctx)
       ramses_amd_mgdist_ctx = c_null_ptr
       if (.not. ramses_amd_halo_inited) call ramses_amd_halo_init()
       if (ramses_amd_halo_rccl) then
          rc = ramses_amd_mgdist_create(ilevel, pg, myid - 1, rob, c_null_ptr, ramses_amd_mgdist_ctx)
       else
          ramses_amd_mgdist_tr%user = c_null_ptr
          ramses_amd_mgdist_tr%exchange = c_funloc(ramses_amd_mgdist_cb_exchange)
          ramses_amd_mgdist_tr%allgather = c_funloc(ramses_amd_mgdist_cb_allgather)
          ramses_amd_mgdist_tr%allreduce_sum = c_funloc(ramses_amd_mgdist_cb_allreduce)
          rc = ramses_amd_mgdist_create(ilevel, pg, myid - 1, rob, c_loc(ramses_amd_mgdist_tr), ramses_amd_mgdist_ctx)
       end if
       if (rc /= 0) call ramses_amd_fatal('multigrid_fine (distributed dense multigrid, setup)')
       ramses_amd_mgdist_level = ilevel
       ramses_amd_mgdist_pg = pg
       if (allocated(ramses_amd_mgdist_rob)) deallocate(ramses_amd_mgdist_rob)
       allocate(ramses_amd_mgdist_rob(ncpu))
       ramses_amd_mgdist_rob = rob
       if (myid == 1) write(*,'(A,I3,A,I2,A,I2,A,I2,A,I5,A,I5,A,I5,A)') ' ramses_amd: multigrid of level ', ilevel, &
            & ' distributed over ', pg(1), ' x', pg(2), ' x', pg(3), ' bricks of ', dims(1), ' x', dims(2), ' x', dims(3), &
            & ' cells, one per rank (dense V-cycles, 5-cell halo per smoother launch)'
    end if
    lo = every(1:3, myid)
    deallocate(every, rob)
    nx_loc = icoarse_max - icoarse_min + 1
    scale = boxlen/dble(nx_loc)
    fourpi = 2*twopi*scale
    if (cosmo) fourpi = 1.5D0*omega_m*aexp*scale
    isafe = 0
    if (safe_mode(ilevel)) isafe = 1
    if (ramses_amd_pois_mpi_dev) then
       ! the steady state of a one-level run: the right-hand side from the deposit on the device, phi stays on the brick
       rc = ramses_amd_mgdist_multigrid_resident_f90(ramses_amd_mgdist_ctx, ilevel, active(ilevel)%ngrid, ramses_amd_octs(ilevel), xg, &
            & int(ngridmax, 8), lo, rho_tot, fourpi, epsilon, isafe, iters, err)
       if (rc /= 0) call ramses_amd_fatal('multigrid_fine (distributed dense multigrid, resident rho / phi)')
       safe_mode(ilevel) = (isafe /= 0)
       ramses_amd_phi_on_device = .true.
    else
       rc = ramses_amd_mgdist_multigrid_f90(ramses_amd_mgdist_ctx, ilevel, active(ilevel)%ngrid, ramses_amd_octs(ilevel), xg, &
            & int(ngridmax, 8), int(ncoarse, 8), lo, rho, phi, rho_tot, fourpi, epsilon, isafe, iters, err)
       if (rc /= 0) call ramses_amd_fatal('multigrid_fine (distributed dense multigrid)')
       safe_mode(ilevel) = (isafe /= 0)
       call make_virtual_fine_dp(phi(1), ilevel)
       ramses_amd_phi_on_device = .false.
       ! from the next solve on (the deposit of THIS step is already in the host vector)
       if (levelmin == nlevelmax .and. ramses_amd_amr_resident() .and. ramses_amd_f_resident_on() &
            & .and. ramses_amd_phi_resident_on()) then
          if (ramses_amd_amrres_active() /= 0) ramses_amd_pois_mpi_dev = .true.
       end if
    end if
    ramses_amd_mgdist_phi_level = ilevel
    ramses_amd_mgdist_lo = lo
    ok = .true.
#endif
  end subroutine ramses_amd_mgdist_multigrid

  !---------------------------------------------------------------------------
  ! force_fine(ilevel,icount) right after ramses_amd_mgdist_multigrid of the same level (several ranks, gravity_type = 0,
  ! periodic, no sinks): halo of phi and gradient_phi on the rank's brick on the device, f(:,1:3) of the rank's cells
  ! back into the host arrays; then, like the reference (poisson/force_fine.f90:135-138,182-188): the virtual octs of
  ! f through make_virtual_fine_dp, the two diagnostics reduced over the ranks.
  !---------------------------------------------------------------------------
  subroutine ramses_amd_mgdist_force_fine(ilevel)
    use amr_commons
    use poisson_commons
    use constants, only: twopi
#ifndef WITHOUTMPI
    use mpi_mod
#endif
    integer, intent(in) :: ilevel
#ifndef WITHOUTMPI
    integer :: rc, info, idim, nx_loc
    real(dp) :: dx, scale, dx_loc, fourpi, fact
    real(kind=8) :: diag(2), epot_all, rho_all
    logical :: resident_f
    nx_loc = icoarse_max - icoarse_min + 1
    dx = 0.5D0**ilevel
    scale = boxlen/dble(nx_loc)
    dx_loc = dx*scale
    fourpi = 2*twopi
    if (cosmo) fourpi = 1.5D0*omega_m*aexp
    fact = -dx_loc**ndim/fourpi/2.0D0
    resident_f = .false.
    if (ramses_amd_amr_resident() .and. ramses_amd_f_resident_on() .and. ilevel >= 1 .and. ilevel <= 64) then
       ! the cell vectors are resident and the level has no finer octs (a uniform run): f goes from the brick into the resident
       ! acceleration on the device, the energy sum runs there in the reference's order, the virtual octs follow with one
       ! exchange of the device arrays -- nothing of f crosses PCIe (the host array is fetched by backup_poisson)
       resident_f = ramses_amd_amrres_active() /= 0
       if (ilevel < nlevelmax) then
          if (numbtot(1, ilevel + 1) > 0) resident_f = .false.
       end if
    end if
    if (resident_f) then
       call ramses_amd_amr_ensure()
       if (ramses_amd_pois_mpi_dev .and. ramses_amd_phi_on_device) then
          rc = ramses_amd_mgdist_force_resident_dev_f90(ramses_amd_mgdist_ctx, ilevel, active(ilevel)%ngrid, &
               & ramses_amd_octs(ilevel), int(ngridmax, 8), int(ncoarse, 8), nvector, fact, diag)
       else
          rc = ramses_amd_mgdist_force_resident_f90(ramses_amd_mgdist_ctx, ilevel, active(ilevel)%ngrid, ramses_amd_octs(ilevel), &
               & int(ngridmax, 8), int(ncoarse, 8), rho, nvector, fact, diag)
       end if
       if (rc /= 0) call ramses_amd_fatal('force_fine (distributed dense multigrid, resident f)')
       call ramses_amd_amr_halo(ilevel, 7)
       ramses_amd_f_on_device(ilevel) = .true.
    else
       rc = ramses_amd_mgdist_force_f90(ramses_amd_mgdist_ctx, ilevel, active(ilevel)%ngrid, ramses_amd_octs(ilevel), xg, &
            & int(ngridmax, 8), int(ncoarse, 8), ramses_amd_mgdist_lo, f, rho, son, nvector, fact, diag)
       if (rc /= 0) call ramses_amd_fatal('force_fine (distributed dense multigrid)')
       do idim = 1, ndim
          call make_virtual_fine_dp(f(1, idim), ilevel)
       end do
    end if
    call MPI_ALLREDUCE(diag(1), epot_all, 1, MPI_DOUBLE_PRECISION, MPI_SUM, MPI_COMM_WORLD, info)
    call MPI_ALLREDUCE(diag(2), rho_all, 1, MPI_DOUBLE_PRECISION, MPI_MAX, MPI_COMM_WORLD, info)
    epot_tot = epot_tot + epot_all
    rho_max(ilevel) = rho_all
#endif
  end subroutine ramses_amd_mgdist_force_fine

end module ramses_amd_iface
