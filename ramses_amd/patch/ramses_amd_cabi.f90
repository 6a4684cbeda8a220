!==============================================================================
! ramses_amd_cabi -- the C ABI of libramses_amd.so as Fortran sees it (include/ramses_amd.h): bind(C) derived types
! mirroring the PODs and one interface per entry point the shims of ramses_amd/patch call.  Declarations only;
! ramses_amd_iface (the state and the procedures of the patch) uses this module and passes it on.
!
! Only plain C types cross the boundary: the shims pass the reference's own module arrays by address (sequence
! association to assumed-size dummies) plus a POD of solver knobs; C never touches Fortran module variables.
!==============================================================================
module ramses_amd_cabi
  use iso_c_binding
  implicit none

  ! struct ramses_amd_hydro_params (include/ramses_amd.h)
  type, bind(C) :: ramses_amd_hydro_params
     integer(c_int32_t) :: ndim, nvar
     real(c_double)     :: gamma, smallr, smallc
     integer(c_int32_t) :: slope_type, riemann
     real(c_double)     :: slope_theta
     integer(c_int32_t) :: scheme, niter_riemann
     real(c_double)     :: difmag, courant_factor
     integer(c_int32_t) :: fast_math, reserved
  end type ramses_amd_hydro_params

  ! struct ramses_amd_brick, only needed for the ABI size check
  type, bind(C) :: ramses_amd_brick
     integer(c_int32_t) :: nx, ny, nz, ng
     integer(c_int64_t) :: pitch_y, pitch_z, pitch_var
  end type ramses_amd_brick

  ! struct ramses_amd_mg_transport: the caller's message layer of the distributed multigrid (host buffers)
  type, bind(C) :: ramses_amd_mg_transport
     type(c_ptr)    :: user
     type(c_funptr) :: exchange, allgather, allreduce_sum
  end type ramses_amd_mg_transport

  interface
     ! distributed dense multigrid of a fully refined periodic level, one brick per rank (csrc/mg_dist.hip)
     function ramses_amd_mgdist_create(level, pgrid, rank, rank_of_brick, transport, ctx) &
          & bind(C, name='ramses_amd_mgdist_create') result(rc)
       import :: c_int, c_ptr
       integer(c_int), value :: level, rank
       integer(c_int) :: pgrid(3), rank_of_brick(*)
       type(c_ptr), value :: transport        ! c_loc of a ramses_amd_mg_transport, or c_null_ptr: RCCL inside the library
       type(c_ptr) :: ctx
       integer(c_int) :: rc
     end function ramses_amd_mgdist_create

     function ramses_amd_mgdist_destroy(ctx) bind(C, name='ramses_amd_mgdist_destroy') result(rc)
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx
       integer(c_int) :: rc
     end function ramses_amd_mgdist_destroy

     function ramses_amd_mgdist_oct_box(ilevel, ngrid, igrid, xg, ngridmax, lo, dims) &
          & bind(C, name='ramses_amd_mgdist_oct_box') result(rc)
       import :: c_int, c_int64_t, c_double
       integer(c_int), value :: ilevel, ngrid
       integer(c_int) :: igrid(*)
       real(c_double) :: xg(*)
       integer(c_int64_t), value :: ngridmax
       integer(c_int) :: lo(3), dims(3)
       integer(c_int) :: rc
     end function ramses_amd_mgdist_oct_box

     function ramses_amd_mgdist_multigrid_f90(ctx, ilevel, ngrid, igrid, xg, ngridmax, ncoarse, lo, rho, phi, &
          & rho_tot, fourpi, epsilon, safe_mode, iters, err) bind(C, name='ramses_amd_mgdist_multigrid_f90') result(rc)
       import :: c_int, c_int64_t, c_double, c_ptr
       type(c_ptr), value :: ctx
       integer(c_int), value :: ilevel, ngrid
       integer(c_int) :: igrid(*)
       real(c_double) :: xg(*)
       integer(c_int64_t), value :: ngridmax, ncoarse
       integer(c_int) :: lo(3)
       real(c_double) :: rho(*), phi(*)
       real(c_double), value :: rho_tot, fourpi, epsilon
       integer(c_int) :: safe_mode, iters
       real(c_double) :: err
       integer(c_int) :: rc
     end function ramses_amd_mgdist_multigrid_f90

     function ramses_amd_mgdist_force_f90(ctx, ilevel, ngrid, igrid, xg, ngridmax, ncoarse, lo, f, rho, son, nvec, fact, diag) &
          & bind(C, name='ramses_amd_mgdist_force_f90') result(rc)
       import :: c_int, c_int64_t, c_double, c_ptr
       type(c_ptr), value :: ctx
       integer(c_int), value :: ilevel, ngrid, nvec
       integer(c_int) :: igrid(*), lo(3), son(*)
       real(c_double) :: xg(*), f(*), rho(*)
       integer(c_int64_t), value :: ngridmax, ncoarse
       real(c_double), value :: fact
       real(c_double) :: diag(2)
       integer(c_int) :: rc
     end function ramses_amd_mgdist_force_f90
     function ramses_amd_mgdist_multigrid_resident_f90(ctx, ilevel, ngrid, igrid, xg, ngridmax, lo, rho_tot, fourpi, epsilon, &
          & safe_mode, iters, err) bind(C, name='ramses_amd_mgdist_multigrid_resident_f90') result(rc)
       import :: c_int, c_int64_t, c_double, c_ptr
       type(c_ptr), value :: ctx
       integer(c_int), value :: ilevel, ngrid
       integer(c_int) :: igrid(*)
       real(c_double) :: xg(*)
       integer(c_int64_t), value :: ngridmax
       integer(c_int) :: lo(3)
       real(c_double), value :: rho_tot, fourpi, epsilon
       integer(c_int) :: safe_mode, iters
       real(c_double) :: err
       integer(c_int) :: rc
     end function ramses_amd_mgdist_multigrid_resident_f90
     function ramses_amd_mgdist_fetch_phi_f90(ctx, ngrid, igrid, ngridmax, ncoarse, phi) &
          & bind(C, name='ramses_amd_mgdist_fetch_phi_f90') result(rc)
       import :: c_int, c_int64_t, c_double, c_ptr
       type(c_ptr), value :: ctx
       integer(c_int), value :: ngrid
       integer(c_int) :: igrid(*)
       integer(c_int64_t), value :: ngridmax, ncoarse
       real(c_double) :: phi(*)
       integer(c_int) :: rc
     end function ramses_amd_mgdist_fetch_phi_f90
     function ramses_amd_mgdist_force_resident_dev_f90(ctx, ilevel, ngrid, igrid, ngridmax, ncoarse, nvec, fact, diag) &
          & bind(C, name='ramses_amd_mgdist_force_resident_dev_f90') result(rc)
       import :: c_int, c_int64_t, c_double, c_ptr
       type(c_ptr), value :: ctx
       integer(c_int), value :: ilevel, ngrid, nvec
       integer(c_int) :: igrid(*)
       integer(c_int64_t), value :: ngridmax, ncoarse
       real(c_double), value :: fact
       real(c_double) :: diag(2)
       integer(c_int) :: rc
     end function ramses_amd_mgdist_force_resident_dev_f90
     function ramses_amd_amrres_rho_keep(on) bind(C, name='ramses_amd_amrres_rho_keep') result(rc)
       import :: c_int
       integer(c_int), value :: on
       integer(c_int) :: rc
     end function ramses_amd_amrres_rho_keep
     function ramses_amd_amrres_sync_rho(ngrid, igrid, rho) bind(C, name='ramses_amd_amrres_sync_rho') result(rc)
       import :: c_int, c_double
       integer(c_int), value :: ngrid
       integer(c_int) :: igrid(*)
       real(c_double) :: rho(*)
       integer(c_int) :: rc
     end function ramses_amd_amrres_sync_rho
     function ramses_amd_mgdist_force_resident_f90(ctx, ilevel, ngrid, igrid, ngridmax, ncoarse, rho, nvec, fact, diag) &
          & bind(C, name='ramses_amd_mgdist_force_resident_f90') result(rc)
       import :: c_int, c_int64_t, c_double, c_ptr
       type(c_ptr), value :: ctx
       integer(c_int), value :: ilevel, ngrid, nvec
       integer(c_int) :: igrid(*)
       real(c_double) :: rho(*)
       integer(c_int64_t), value :: ngridmax, ncoarse
       real(c_double), value :: fact
       real(c_double) :: diag(2)
       integer(c_int) :: rc
     end function ramses_amd_mgdist_force_resident_f90

     function ramses_amd_abi_check(sz_params, sz_brick) bind(C, name='ramses_amd_abi_check') result(rc)
       import :: c_size_t, c_int
       integer(c_size_t), value :: sz_params, sz_brick
       integer(c_int) :: rc
     end function ramses_amd_abi_check

     function ramses_amd_set_device_auto(world_rank) bind(C, name='ramses_amd_set_device_auto') result(rc)
       import :: c_int
       integer(c_int), value :: world_rank
       integer(c_int) :: rc
     end function ramses_amd_set_device_auto

     function ramses_amd_last_error() bind(C, name='ramses_amd_last_error') result(msg)
       import :: c_ptr
       type(c_ptr) :: msg
     end function ramses_amd_last_error

     ! f_or_dummy: f(1,1) when has_f/=0, any valid array otherwise (not read)
     function ramses_amd_godunov_fine_f90(p, ilevel, ngrid, igrid, xg, ngridmax, ncoarse, nx_loc, &
          & uold, unew, f_or_dummy, has_f, dx, dt) bind(C, name='ramses_amd_godunov_fine_f90') result(rc)
       import :: ramses_amd_hydro_params, c_int, c_int64_t, c_double
       type(ramses_amd_hydro_params), intent(in) :: p
       integer(c_int), value :: ilevel, ngrid
       integer(c_int) :: igrid(*)
       real(c_double) :: xg(*)
       integer(c_int64_t), value :: ngridmax, ncoarse
       integer(c_int), value :: nx_loc
       real(c_double) :: uold(*), unew(*), f_or_dummy(*)
       integer(c_int), value :: has_f
       real(c_double), value :: dx, dt
       integer(c_int) :: rc
     end function ramses_amd_godunov_fine_f90
     ! NDIM = 1, 2 builds: a fully refined level, active octs + the octs of the boundary regions (csrc/capi_host.hip)
     function ramses_amd_godunov_fine_lowdim_f90(p, ilevel, ngrid, igrid, nbound, igrid_bound, xg, ngridmax, ncoarse, skip, nloc, &
          & uold, unew, dx, dt) bind(C, name='ramses_amd_godunov_fine_lowdim_f90') result(rc)
       import :: ramses_amd_hydro_params, c_int, c_int64_t, c_double
       type(ramses_amd_hydro_params), intent(in) :: p
       integer(c_int), value :: ilevel, ngrid, nbound
       integer(c_int) :: igrid(*), igrid_bound(*), skip(*), nloc(*)
       real(c_double) :: xg(*)
       integer(c_int64_t), value :: ngridmax, ncoarse
       real(c_double) :: uold(*), unew(*)
       real(c_double), value :: dx, dt
       integer(c_int) :: rc
     end function ramses_amd_godunov_fine_lowdim_f90
     function ramses_amd_lowdim_note_reference(ilevel) bind(C, name='ramses_amd_lowdim_note_reference') result(rc)
       import :: c_int
       integer(c_int), value :: ilevel
       integer(c_int) :: rc
     end function ramses_amd_lowdim_note_reference
     function ramses_amd_multigrid_fine_f90(ilevel, ngrid, igrid, xg, ngridmax, ncoarse, nx_loc, &
          & rho, phi, rho_tot, fourpi, epsilon, safe_mode, iters, err) &
          & bind(C, name='ramses_amd_multigrid_fine_f90') result(rc)
       import :: c_int, c_int64_t, c_double
       integer(c_int), value :: ilevel, ngrid
       integer(c_int) :: igrid(*)
       real(c_double) :: xg(*)
       integer(c_int64_t), value :: ngridmax, ncoarse
       integer(c_int), value :: nx_loc
       real(c_double) :: rho(*), phi(*)
       real(c_double), value :: rho_tot, fourpi, epsilon
       integer(c_int) :: safe_mode, iters
       real(c_double) :: err
       integer(c_int) :: rc
     end function ramses_amd_multigrid_fine_f90

     ! ---- AMR level: the reference's tree arrays by address ----
     function ramses_amd_godunov_fine_amr_f90(p, ilevel, ngrid, igrid, son, nbor, father, ngridmax, ncoarse, &
          & uold, unew, f_or_dummy, has_f, divu_or_dummy, enew_or_dummy, has_pfix, dx, dt, nvector, &
          & interpol_var, interpol_type) bind(C, name='ramses_amd_godunov_fine_amr_f90') result(rc)
       import :: ramses_amd_hydro_params, c_int, c_int64_t, c_double
       type(ramses_amd_hydro_params), intent(in) :: p
       integer(c_int), value :: ilevel, ngrid
       integer(c_int) :: igrid(*), son(*), nbor(*), father(*)
       integer(c_int64_t), value :: ngridmax, ncoarse
       real(c_double) :: uold(*), unew(*), f_or_dummy(*), divu_or_dummy(*), enew_or_dummy(*)
       integer(c_int), value :: has_f, has_pfix
       real(c_double), value :: dx, dt
       integer(c_int), value :: nvector, interpol_var, interpol_type
       integer(c_int) :: rc
     end function ramses_amd_godunov_fine_amr_f90

     function ramses_amd_force_fine_f90(ilevel, ngrid, igrid, xg, ngridmax, ncoarse, nx_loc, phi, f, &
          & rho, son, has_son, fact, diag2) bind(C, name='ramses_amd_force_fine_f90') result(rc)
       import :: c_int, c_int64_t, c_double
       integer(c_int), value :: ilevel, ngrid, nx_loc, has_son
       integer(c_int64_t), value :: ngridmax, ncoarse
       integer(c_int) :: igrid(*), son(*)
       real(c_double) :: xg(*), phi(*), f(*), rho(*), diag2(2)
       real(c_double), value :: fact
       integer(c_int) :: rc
     end function ramses_amd_force_fine_f90
     ! ---- Poisson branch on the resident level ----
     function ramses_amd_resident_rho_fine_f90(p, ilevel, ngrid, igrid, xg, ngridmax, ncoarse, nx_loc, uold, boxlen, &
          & nvector, multipole4) bind(C, name='ramses_amd_resident_rho_fine_f90') result(rc)
       import :: ramses_amd_hydro_params, c_int, c_int64_t, c_double
       type(ramses_amd_hydro_params), intent(in) :: p
       integer(c_int), value :: ilevel, ngrid, nx_loc, nvector
       integer(c_int64_t), value :: ngridmax, ncoarse
       integer(c_int) :: igrid(*)
       real(c_double) :: xg(*), uold(*), multipole4(4)
       real(c_double), value :: boxlen
       integer(c_int) :: rc
     end function ramses_amd_resident_rho_fine_f90
     function ramses_amd_resident_multigrid_f90(ilevel, rho_tot, fourpi, epsilon, safe_mode, iters, err) &
          & bind(C, name='ramses_amd_resident_multigrid_f90') result(rc)
       import :: c_int, c_double
       integer(c_int), value :: ilevel
       real(c_double), value :: rho_tot, fourpi, epsilon
       integer(c_int) :: safe_mode, iters
       real(c_double) :: err
       integer(c_int) :: rc
     end function ramses_amd_resident_multigrid_f90
     function ramses_amd_resident_force_fine_f90(ilevel, fact, diag2) &
          & bind(C, name='ramses_amd_resident_force_fine_f90') result(rc)
       import :: c_int, c_double
       integer(c_int), value :: ilevel
       real(c_double), value :: fact
       real(c_double) :: diag2(2)
       integer(c_int) :: rc
     end function ramses_amd_resident_force_fine_f90
     function ramses_amd_resident_sync_poisson_f90(phi, f, rho) bind(C, name='ramses_amd_resident_sync_poisson_f90') result(rc)
       import :: c_int, c_double
       real(c_double) :: phi(*), f(*), rho(*)
       integer(c_int) :: rc
     end function ramses_amd_resident_sync_poisson_f90
     function ramses_amd_host_register_dp(p, bytes) bind(C, name='ramses_amd_host_register') result(rc)
       import :: c_int, c_int64_t, c_double
       real(c_double) :: p(*)
       integer(c_int64_t), value :: bytes
       integer(c_int) :: rc
     end function ramses_amd_host_register_dp
     function ramses_amd_host_register_int(p, bytes) bind(C, name='ramses_amd_host_register') result(rc)
       import :: c_int, c_int64_t
       integer(c_int) :: p(*)
       integer(c_int64_t), value :: bytes
       integer(c_int) :: rc
     end function ramses_amd_host_register_int

     ! ---- conjugate-gradient solver on an AMR level (include/ramses_amd.h) ----
     function ramses_amd_cg_solve_host(ilevel, ngrid, igrid, son, nbor, ngridmax, ncoarse, phi, f, rho, rho_tot, &
          & fact, ncell_level, epsilon, itermax, ordered, iter, err) bind(C, name='ramses_amd_cg_solve_host') result(rc)
       import :: c_int, c_int64_t, c_double
       integer(c_int), value :: ilevel, ngrid, itermax, ordered
       integer(c_int64_t), value :: ngridmax, ncoarse
       real(c_double), value :: rho_tot, fact, ncell_level, epsilon
       integer(c_int) :: igrid(*), son(*), nbor(*)
       real(c_double) :: phi(*), f(*), rho(*), err(3)
       integer(c_int) :: iter
       integer(c_int) :: rc
     end function ramses_amd_cg_solve_host
     ! the same loop with several MPI ranks, one routine at a time (the shim owns the MPI_ALLREDUCEs and the halo of p)
     function ramses_amd_cgmpi_begin(ilevel, ngrid, igrid, son, nbor, ngridmax, ncoarse, phi, f, rho, rho_tot, fact, ordered, &
          & out2) bind(C, name='ramses_amd_cgmpi_begin') result(rc)
       import :: c_int, c_int64_t, c_double
       integer(c_int), value :: ilevel, ngrid, ordered
       integer(c_int64_t), value :: ngridmax, ncoarse
       real(c_double), value :: rho_tot, fact
       integer(c_int) :: igrid(*), son(*), nbor(*)
       real(c_double) :: phi(*), f(*), rho(*), out2(2)
       integer(c_int) :: rc
     end function ramses_amd_cgmpi_begin
     function ramses_amd_cgmpi_get(slot, val) bind(C, name='ramses_amd_cgmpi_get') result(rc)
       import :: c_int, c_double
       integer(c_int), value :: slot
       real(c_double) :: val
       integer(c_int) :: rc
     end function ramses_amd_cgmpi_get
     function ramses_amd_cgmpi_set(slot, val) bind(C, name='ramses_amd_cgmpi_set') result(rc)
       import :: c_int, c_double
       integer(c_int), value :: slot
       real(c_double), value :: val
       integer(c_int) :: rc
     end function ramses_amd_cgmpi_set
     function ramses_amd_cgmpi_step(step, iter) bind(C, name='ramses_amd_cgmpi_step') result(rc)
       import :: c_int
       integer(c_int), value :: step, iter
       integer(c_int) :: rc
     end function ramses_amd_cgmpi_step
     function ramses_amd_cgmpi_p_cells(n, igrid, to_host) bind(C, name='ramses_amd_cgmpi_p_cells') result(rc)
       import :: c_int
       integer(c_int), value :: n, to_host
       integer(c_int) :: igrid(*)
       integer(c_int) :: rc
     end function ramses_amd_cgmpi_p_cells
     function ramses_amd_cgmpi_comm_set(ncpu, em_n, em_ig, rc_n, rc_ig) bind(C, name='ramses_amd_cgmpi_comm_set') result(rc)
       import :: c_int
       integer(c_int), value :: ncpu
       integer(c_int) :: em_n(*), em_ig(*), rc_n(*), rc_ig(*)
       integer(c_int) :: rc
     end function ramses_amd_cgmpi_comm_set
     function ramses_amd_cgmpi_p_halo_stage_out(ncpu, h_send_addr, h_recv_addr, send_off, recv_off) &
          & bind(C, name='ramses_amd_cgmpi_p_halo_stage_out') result(rc)
       import :: c_int, c_int64_t, c_ptr
       integer(c_int), value :: ncpu
       type(c_ptr) :: h_send_addr, h_recv_addr        ! int64_t* on the C side: the addresses of the pinned buffers
       integer(c_int64_t) :: send_off(*), recv_off(*)
       integer(c_int) :: rc
     end function ramses_amd_cgmpi_p_halo_stage_out
     function ramses_amd_cgmpi_p_halo_stage_in() bind(C, name='ramses_amd_cgmpi_p_halo_stage_in') result(rc)
       import :: c_int
       integer(c_int) :: rc
     end function ramses_amd_cgmpi_p_halo_stage_in
     function ramses_amd_cgmpi_p_halo_rccl() bind(C, name='ramses_amd_cgmpi_p_halo_rccl') result(rc)
       import :: c_int
       integer(c_int) :: rc
     end function ramses_amd_cgmpi_p_halo_rccl
     function ramses_amd_cgmpi_end(phi, f) bind(C, name='ramses_amd_cgmpi_end') result(rc)
       import :: c_int, c_double
       real(c_double) :: phi(*), f(*)
       integer(c_int) :: rc
     end function ramses_amd_cgmpi_end

     ! ---- multigrid on AMR levels (include/ramses_amd.h) ----
     function ramses_amd_mgamr_begin(ilevel, ngridmax, ncoarse, son, nbor, father, lookup_mg, flag2, phi, f, &
          & ngrid, igrid) bind(C, name='ramses_amd_mgamr_begin') result(rc)
       import :: c_int, c_int64_t, c_double
       integer(c_int), value :: ilevel, ngrid
       integer(c_int64_t), value :: ngridmax, ncoarse
       integer(c_int) :: son(*), nbor(*), father(*), lookup_mg(*), flag2(*), igrid(*)
       real(c_double) :: phi(*), f(*)
       integer(c_int) :: rc
     end function ramses_amd_mgamr_begin
     function ramses_amd_mgamr_add_level(level, ngrid, igrid, u, fscan) &
          & bind(C, name='ramses_amd_mgamr_add_level') result(rc)
       import :: c_int, c_double
       integer(c_int), value :: level, ngrid
       integer(c_int) :: igrid(*), fscan(*)
       real(c_double) :: u(*)
       integer(c_int) :: rc
     end function ramses_amd_mgamr_add_level
     function ramses_amd_mgamr_gauss_seidel(level, redstep, safe) bind(C, name='ramses_amd_mgamr_gauss_seidel') result(rc)
       import :: c_int
       integer(c_int), value :: level, redstep, safe
       integer(c_int) :: rc
     end function ramses_amd_mgamr_gauss_seidel
     function ramses_amd_mgamr_residual(level) bind(C, name='ramses_amd_mgamr_residual') result(rc)
       import :: c_int
       integer(c_int), value :: level
       integer(c_int) :: rc
     end function ramses_amd_mgamr_residual
     function ramses_amd_mgamr_norm2(level, norm2) bind(C, name='ramses_amd_mgamr_norm2') result(rc)
       import :: c_int, c_double
       integer(c_int), value :: level
       real(c_double) :: norm2
       integer(c_int) :: rc
     end function ramses_amd_mgamr_norm2
     function ramses_amd_mgamr_restrict(finelevel) bind(C, name='ramses_amd_mgamr_restrict') result(rc)
       import :: c_int
       integer(c_int), value :: finelevel
       integer(c_int) :: rc
     end function ramses_amd_mgamr_restrict
     function ramses_amd_mgamr_interpolate(finelevel) bind(C, name='ramses_amd_mgamr_interpolate') result(rc)
       import :: c_int
       integer(c_int), value :: finelevel
       integer(c_int) :: rc
     end function ramses_amd_mgamr_interpolate
     function ramses_amd_mgamr_level_begin(level, ngrid_total) bind(C, name='ramses_amd_mgamr_level_begin') result(rc)
       import :: c_int
       integer(c_int), value :: level, ngrid_total
       integer(c_int) :: rc
     end function ramses_amd_mgamr_level_begin
     function ramses_amd_mgamr_level_block(level, ngrid, igrid, u, fscan) bind(C, name='ramses_amd_mgamr_level_block') result(rc)
       import :: c_int, c_double
       integer(c_int), value :: level, ngrid
       integer(c_int) :: igrid(*), fscan(*)
       real(c_double) :: u(*)
       integer(c_int) :: rc
     end function ramses_amd_mgamr_level_block
     function ramses_amd_mgamr_fine_active(nact) bind(C, name='ramses_amd_mgamr_fine_active') result(rc)
       import :: c_int
       integer(c_int), value :: nact
       integer(c_int) :: rc
     end function ramses_amd_mgamr_fine_active
     function ramses_amd_mgamr_force_sync(on) bind(C, name='ramses_amd_mgamr_force_sync') result(rc)
       import :: c_int
       integer(c_int), value :: on
       integer(c_int) :: rc
     end function ramses_amd_mgamr_force_sync
     function ramses_amd_mgamr_end() bind(C, name='ramses_amd_mgamr_end') result(rc)
       import :: c_int
       integer(c_int) :: rc
     end function ramses_amd_mgamr_end
     function ramses_amd_mgamr_comm_set(level, ncpu, myid, em_n, em_list, list_is_octs, rc_n) &
          & bind(C, name='ramses_amd_mgamr_comm_set') result(rc)
       import :: c_int
       integer(c_int), value :: level, ncpu, myid, list_is_octs
       integer(c_int) :: em_n(*), em_list(*), rc_n(*)
       integer(c_int) :: rc
     end function ramses_amd_mgamr_comm_set
     function ramses_amd_mgamr_halo_stage_out(level, comp, dir, ncpu, h_send_addr, h_recv_addr, send_off, recv_off) &
          & bind(C, name='ramses_amd_mgamr_halo_stage_out') result(rc)
       import :: c_int, c_int64_t, c_ptr
       integer(c_int), value :: level, comp, dir, ncpu
       type(c_ptr) :: h_send_addr, h_recv_addr        ! int64_t* on the C side: the addresses of the pinned buffers
       integer(c_int64_t) :: send_off(*), recv_off(*)
       integer(c_int) :: rc
     end function ramses_amd_mgamr_halo_stage_out
     function ramses_amd_mgamr_halo_stage_in(level, comp, dir) bind(C, name='ramses_amd_mgamr_halo_stage_in') result(rc)
       import :: c_int
       integer(c_int), value :: level, comp, dir
       integer(c_int) :: rc
     end function ramses_amd_mgamr_halo_stage_in
     function ramses_amd_mgamr_halo_rccl(level, comp, dir) bind(C, name='ramses_amd_mgamr_halo_rccl') result(rc)
       import :: c_int
       integer(c_int), value :: level, comp, dir
       integer(c_int) :: rc
     end function ramses_amd_mgamr_halo_rccl
     function ramses_amd_poisamr_tree(epoch, ngridmax, ncoarse, son, nbor, father) &
          & bind(C, name='ramses_amd_poisamr_tree') result(rc)
       import :: c_int, c_int64_t
       integer(c_int), value :: epoch
       integer(c_int64_t), value :: ngridmax, ncoarse
       integer(c_int) :: son(*), nbor(*), father(*)
       integer(c_int) :: rc
     end function ramses_amd_poisamr_tree
     function ramses_amd_warmup() bind(C, name='ramses_amd_warmup') result(rc)
       import :: c_int
       integer(c_int) :: rc
     end function ramses_amd_warmup
     function ramses_amd_poisamr_force(ilevel, ngrid, igrid, ngrid_c, igrid_c, phi, phi_old, rho, f, tfrac, interp, fresh, &
          & fact, diag) bind(C, name='ramses_amd_poisamr_force') result(rc)
       import :: c_int, c_double
       integer(c_int), value :: ilevel, ngrid, ngrid_c, interp, fresh
       integer(c_int) :: igrid(*), igrid_c(*)
       real(c_double) :: phi(*), phi_old(*), rho(*), f(*), diag(2)
       real(c_double), value :: tfrac, fact
       integer(c_int) :: rc
     end function ramses_amd_poisamr_force
     function ramses_amd_poisamr_force_mpi(ilevel, ngrid_own, ngrid_all, igrid_all, ngrid_c_all, igrid_c_all, phi, phi_old, rho, f, &
          & tfrac, interp, fact, diag) bind(C, name='ramses_amd_poisamr_force_mpi') result(rc)
       import :: c_int, c_double
       integer(c_int), value :: ilevel, ngrid_own, ngrid_all, ngrid_c_all, interp
       integer(c_int) :: igrid_all(*), igrid_c_all(*)
       real(c_double) :: phi(*), phi_old(*), rho(*), f(*), diag(2)
       real(c_double), value :: tfrac, fact
       integer(c_int) :: rc
     end function ramses_amd_poisamr_force_mpi
     function ramses_amd_poisamr_force_mpi_resident(ilevel, ngrid_own, ngrid_all, igrid_all, ngrid_c_all, igrid_c_all, phi, phi_old, &
          & rho, tfrac, interp, fact, diag) bind(C, name='ramses_amd_poisamr_force_mpi_resident') result(rc)
       import :: c_int, c_double
       integer(c_int), value :: ilevel, ngrid_own, ngrid_all, ngrid_c_all, interp
       integer(c_int) :: igrid_all(*), igrid_c_all(*)
       real(c_double) :: phi(*), phi_old(*), rho(*), diag(2)
       real(c_double), value :: tfrac, fact
       integer(c_int) :: rc
     end function ramses_amd_poisamr_force_mpi_resident
     function ramses_amd_prof_add(name, level, seconds) bind(C, name='ramses_amd_prof_add') result(rc)
       import :: c_int, c_double, c_char
       character(kind=c_char) :: name(*)
       integer(c_int), value :: level
       real(c_double), value :: seconds
       integer(c_int) :: rc
     end function ramses_amd_prof_add
     function ramses_amd_poisamr_multigrid(ilevel, ngrid, igrid, ngrid_c, igrid_c, phi, phi_old, rho, flag2, rho_tot, fourpi, &
          & tfrac, interp, epsilon, ngs_fine, ngs_coarse, ncycles_coarse_safe, safe_mode, iters, err) &
          & bind(C, name='ramses_amd_poisamr_multigrid') result(rc)
       import :: c_int, c_double
       integer(c_int), value :: ilevel, ngrid, ngrid_c, interp, ngs_fine, ngs_coarse, ncycles_coarse_safe
       integer(c_int) :: igrid(*), igrid_c(*), flag2(*)
       real(c_double) :: phi(*), phi_old(*), rho(*)
       real(c_double), value :: rho_tot, fourpi, tfrac, epsilon
       integer(c_int) :: safe_mode, iters
       real(c_double) :: err
       integer(c_int) :: rc
     end function ramses_amd_poisamr_multigrid

     ! ---- device-resident level (include/ramses_amd.h) ----
     function ramses_amd_resident_courant_f90(p, ilevel, ngrid, igrid, xg, ngridmax, ncoarse, nx_loc, &
          & uold, dx, dt_in, out4) bind(C, name='ramses_amd_resident_courant_f90') result(rc)
       import :: ramses_amd_hydro_params, c_int, c_int64_t, c_double
       type(ramses_amd_hydro_params), intent(in) :: p
       integer(c_int), value :: ilevel, ngrid
       integer(c_int) :: igrid(*)
       real(c_double) :: xg(*)
       integer(c_int64_t), value :: ngridmax, ncoarse
       integer(c_int), value :: nx_loc
       real(c_double) :: uold(*)
       real(c_double), value :: dx, dt_in
       real(c_double) :: out4(4)
       integer(c_int) :: rc
     end function ramses_amd_resident_courant_f90
     function ramses_amd_resident_godunov_f90(p, ilevel, ngrid, igrid, xg, ngridmax, ncoarse, nx_loc, &
          & uold, dx, dt) bind(C, name='ramses_amd_resident_godunov_f90') result(rc)
       import :: ramses_amd_hydro_params, c_int, c_int64_t, c_double
       type(ramses_amd_hydro_params), intent(in) :: p
       integer(c_int), value :: ilevel, ngrid
       integer(c_int) :: igrid(*)
       real(c_double) :: xg(*)
       integer(c_int64_t), value :: ngridmax, ncoarse
       integer(c_int), value :: nx_loc
       real(c_double) :: uold(*)
       real(c_double), value :: dx, dt
       integer(c_int) :: rc
     end function ramses_amd_resident_godunov_f90
     function ramses_amd_resident_set_uold_f90(ilevel) bind(C, name='ramses_amd_resident_set_uold_f90') result(rc)
       import :: c_int
       integer(c_int), value :: ilevel
       integer(c_int) :: rc
     end function ramses_amd_resident_set_uold_f90
     function ramses_amd_resident_sync_host_f90(uold) bind(C, name='ramses_amd_resident_sync_host_f90') result(rc)
       import :: c_int, c_double
       real(c_double) :: uold(*)
       integer(c_int) :: rc
     end function ramses_amd_resident_sync_host_f90
     ! ---- gravity on the resident level ----
     function ramses_amd_resident_synchro_f90(p, ilevel, ngrid, igrid, xg, ngridmax, ncoarse, nx_loc, &
          & uold, f, dteff) bind(C, name='ramses_amd_resident_synchro_f90') result(rc)
       import :: ramses_amd_hydro_params, c_int, c_int64_t, c_double
       type(ramses_amd_hydro_params), intent(in) :: p
       integer(c_int), value :: ilevel, ngrid
       integer(c_int) :: igrid(*)
       real(c_double) :: xg(*)
       integer(c_int64_t), value :: ngridmax, ncoarse
       integer(c_int), value :: nx_loc
       real(c_double) :: uold(*), f(*)
       real(c_double), value :: dteff
       integer(c_int) :: rc
     end function ramses_amd_resident_synchro_f90
     function ramses_amd_resident_courant_grav_f90(p, ilevel, ngrid, igrid, xg, ngridmax, ncoarse, nx_loc, &
          & uold, f, dx, dt_in, out4) bind(C, name='ramses_amd_resident_courant_grav_f90') result(rc)
       import :: ramses_amd_hydro_params, c_int, c_int64_t, c_double
       type(ramses_amd_hydro_params), intent(in) :: p
       integer(c_int), value :: ilevel, ngrid
       integer(c_int) :: igrid(*)
       real(c_double) :: xg(*)
       integer(c_int64_t), value :: ngridmax, ncoarse
       integer(c_int), value :: nx_loc
       real(c_double) :: uold(*), f(*)
       real(c_double), value :: dx, dt_in
       real(c_double) :: out4(4)
       integer(c_int) :: rc
     end function ramses_amd_resident_courant_grav_f90
     function ramses_amd_resident_godunov_grav_f90(p, ilevel, ngrid, igrid, xg, ngridmax, ncoarse, nx_loc, &
          & uold, f, dx, dt) bind(C, name='ramses_amd_resident_godunov_grav_f90') result(rc)
       import :: ramses_amd_hydro_params, c_int, c_int64_t, c_double
       type(ramses_amd_hydro_params), intent(in) :: p
       integer(c_int), value :: ilevel, ngrid
       integer(c_int) :: igrid(*)
       real(c_double) :: xg(*)
       integer(c_int64_t), value :: ngridmax, ncoarse
       integer(c_int), value :: nx_loc
       real(c_double) :: uold(*), f(*)
       real(c_double), value :: dx, dt
       integer(c_int) :: rc
     end function ramses_amd_resident_godunov_grav_f90
     function ramses_amd_resident_set_uold_grav_f90(p, ilevel, dt) &
          & bind(C, name='ramses_amd_resident_set_uold_grav_f90') result(rc)
       import :: ramses_amd_hydro_params, c_int, c_double
       type(ramses_amd_hydro_params), intent(in) :: p
       integer(c_int), value :: ilevel
       real(c_double), value :: dt
       integer(c_int) :: rc
     end function ramses_amd_resident_set_uold_grav_f90
     function ramses_amd_resident_sync_density_f90(uold) bind(C, name='ramses_amd_resident_sync_density_f90') result(rc)
       import :: c_int, c_double
       real(c_double) :: uold(*)
       integer(c_int) :: rc
     end function ramses_amd_resident_sync_density_f90
     ! ---- MPI: one rank per GPU (include/ramses_amd.h) ----
     function ramses_amd_device_uid(uid) bind(C, name='ramses_amd_device_uid') result(rc)
       import :: c_int, c_int64_t
       integer(c_int64_t) :: uid
       integer(c_int) :: rc
     end function ramses_amd_device_uid
     function ramses_amd_rccl_probe() bind(C, name='ramses_amd_rccl_probe') result(rc)
       import :: c_int
       integer(c_int) :: rc
     end function ramses_amd_rccl_probe
     function ramses_amd_rccl_unique_id(id) bind(C, name='ramses_amd_rccl_unique_id') result(rc)
       import :: c_int, c_char
       character(kind=c_char) :: id(128)
       integer(c_int) :: rc
     end function ramses_amd_rccl_unique_id
     function ramses_amd_rccl_init(id, nranks, rank) bind(C, name='ramses_amd_rccl_init') result(rc)
       import :: c_int, c_char
       character(kind=c_char) :: id(128)
       integer(c_int), value :: nranks, rank
       integer(c_int) :: rc
     end function ramses_amd_rccl_init
     function ramses_amd_rccl_finalize() bind(C, name='ramses_amd_rccl_finalize') result(rc)
       import :: c_int
       integer(c_int) :: rc
     end function ramses_amd_rccl_finalize
     function ramses_amd_halo_plan(ilevel, ngrid, igrid, xg, ngridmax, ncpu, em_ngrid, em_igrid, rc_ngrid, rc_igrid, &
          & out_box, act_org, em_org, rc_src, rc_org, rc_cap) bind(C, name='ramses_amd_halo_plan') result(rc)
       import :: c_int, c_int64_t, c_double, c_ptr
       integer(c_int), value :: ilevel, ngrid, ncpu
       integer(c_int64_t), value :: ngridmax, rc_cap
       integer(c_int) :: igrid(*), em_ngrid(*), em_igrid(*), rc_ngrid(*), rc_igrid(*), out_box(8)
       real(c_double) :: xg(*)
       type(c_ptr), value :: act_org, em_org, rc_src, rc_org
       integer(c_int) :: rc
     end function ramses_amd_halo_plan
     function ramses_amd_mpires_setup(p, ilevel, ngrid, igrid, xg, ngridmax, ncoarse, nx_loc, uold, unew, ncpu, myid, &
          & em_ngrid, em_igrid, rc_ngrid, rc_igrid) bind(C, name='ramses_amd_mpires_setup') result(rc)
       import :: ramses_amd_hydro_params, c_int, c_int64_t, c_double
       type(ramses_amd_hydro_params), intent(in) :: p
       integer(c_int), value :: ilevel, ngrid, nx_loc, ncpu, myid
       integer(c_int64_t), value :: ngridmax, ncoarse
       integer(c_int) :: igrid(*), em_ngrid(*), em_igrid(*), rc_ngrid(*), rc_igrid(*)
       real(c_double) :: xg(*), uold(*), unew(*)
       integer(c_int) :: rc
     end function ramses_amd_mpires_setup
     function ramses_amd_mpires_active() bind(C, name='ramses_amd_mpires_active') result(rc)
       import :: c_int
       integer(c_int) :: rc
     end function ramses_amd_mpires_active
     function ramses_amd_mpires_which(xx) bind(C, name='ramses_amd_mpires_which') result(k)
       import :: c_int, c_double
       real(c_double) :: xx(*)
       integer(c_int) :: k
     end function ramses_amd_mpires_which
     function ramses_amd_mpires_courant(p, dx, dt_in, out4) bind(C, name='ramses_amd_mpires_courant') result(rc)
       import :: ramses_amd_hydro_params, c_int, c_double
       type(ramses_amd_hydro_params), intent(in) :: p
       real(c_double), value :: dx, dt_in
       real(c_double) :: out4(4)
       integer(c_int) :: rc
     end function ramses_amd_mpires_courant
     function ramses_amd_mpires_godunov(p, dx, dt) bind(C, name='ramses_amd_mpires_godunov') result(rc)
       import :: ramses_amd_hydro_params, c_int, c_double
       type(ramses_amd_hydro_params), intent(in) :: p
       real(c_double), value :: dx, dt
       integer(c_int) :: rc
     end function ramses_amd_mpires_godunov
     function ramses_amd_mpires_reverse_unew() bind(C, name='ramses_amd_mpires_reverse_unew') result(rc)
       import :: c_int
       integer(c_int) :: rc
     end function ramses_amd_mpires_reverse_unew
     function ramses_amd_mpires_set_uold() bind(C, name='ramses_amd_mpires_set_uold') result(rc)
       import :: c_int
       integer(c_int) :: rc
     end function ramses_amd_mpires_set_uold
     function ramses_amd_mpires_halo_forward() bind(C, name='ramses_amd_mpires_halo_forward') result(rc)
       import :: c_int
       integer(c_int) :: rc
     end function ramses_amd_mpires_halo_forward
     function ramses_amd_mpires_halo_stage_out(h_send, send_off, h_recv, recv_off) &
          & bind(C, name='ramses_amd_mpires_halo_stage_out') result(rc)
       import :: c_int, c_ptr
       type(c_ptr) :: h_send, send_off, h_recv, recv_off
       integer(c_int) :: rc
     end function ramses_amd_mpires_halo_stage_out
     function ramses_amd_mpires_halo_stage_in() bind(C, name='ramses_amd_mpires_halo_stage_in') result(rc)
       import :: c_int
       integer(c_int) :: rc
     end function ramses_amd_mpires_halo_stage_in
     function ramses_amd_mpires_sync_host(uold) bind(C, name='ramses_amd_mpires_sync_host') result(rc)
       import :: c_int, c_double
       real(c_double) :: uold(*)
       integer(c_int) :: rc
     end function ramses_amd_mpires_sync_host
     function ramses_amd_mpires_invalidate() bind(C, name='ramses_amd_mpires_invalidate') result(rc)
       import :: c_int
       integer(c_int) :: rc
     end function ramses_amd_mpires_invalidate

     ! ---- residency for AMR runs (include/ramses_amd.h) ----
     function ramses_amd_amrres_active() bind(C, name='ramses_amd_amrres_active') result(rc)
       import :: c_int
       integer(c_int) :: rc
     end function ramses_amd_amrres_active
     function ramses_amd_amrres_load(nvar, ngridmax, ncoarse, uold, son, nbor, father) &
          & bind(C, name='ramses_amd_amrres_load') result(rc)
       import :: c_int, c_int64_t, c_double
       integer(c_int), value :: nvar
       integer(c_int64_t), value :: ngridmax, ncoarse
       real(c_double) :: uold(*)
       integer(c_int) :: son(*), nbor(*), father(*)
       integer(c_int) :: rc
     end function ramses_amd_amrres_load
     function ramses_amd_amrres_tree(son, nbor, father) bind(C, name='ramses_amd_amrres_tree') result(rc)
       import :: c_int
       integer(c_int) :: son(*), nbor(*), father(*)
       integer(c_int) :: rc
     end function ramses_amd_amrres_tree
     function ramses_amd_amrres_invalidate() bind(C, name='ramses_amd_amrres_invalidate') result(rc)
       import :: c_int
       integer(c_int) :: rc
     end function ramses_amd_amrres_invalidate
     function ramses_amd_amrres_sync_level(ngrid, igrid, uold) bind(C, name='ramses_amd_amrres_sync_level') result(rc)
       import :: c_int, c_double
       integer(c_int), value :: ngrid
       integer(c_int) :: igrid(*)
       real(c_double) :: uold(*)
       integer(c_int) :: rc
     end function ramses_amd_amrres_sync_level
     function ramses_amd_amrres_load_level(ngrid, igrid, uold) bind(C, name='ramses_amd_amrres_load_level') result(rc)
       import :: c_int, c_double
       integer(c_int), value :: ngrid
       integer(c_int) :: igrid(*)
       real(c_double) :: uold(*)
       integer(c_int) :: rc
     end function ramses_amd_amrres_load_level
     function ramses_amd_amrres_boundary_hydro(nregion, btype, ngrid, igrid, no_inflow, smallr, nvector, imposed) &
          & bind(C, name='ramses_amd_amrres_boundary_hydro') result(rc)
       import :: c_int, c_double
       integer(c_int), value :: nregion, no_inflow, nvector
       integer(c_int) :: btype(*), ngrid(*), igrid(*)
       real(c_double), value :: smallr
       real(c_double) :: imposed(*)
       integer(c_int) :: rc
     end function ramses_amd_amrres_boundary_hydro
     function ramses_amd_amrres_sync_all(uold) bind(C, name='ramses_amd_amrres_sync_all') result(rc)
       import :: c_int, c_double
       real(c_double) :: uold(*)
       integer(c_int) :: rc
     end function ramses_amd_amrres_sync_all
     function ramses_amd_amrres_set_unew(ngrid, igrid) bind(C, name='ramses_amd_amrres_set_unew') result(rc)
       import :: c_int
       integer(c_int), value :: ngrid
       integer(c_int) :: igrid(*)
       integer(c_int) :: rc
     end function ramses_amd_amrres_set_unew
     function ramses_amd_amrres_set_uold(p, ngrid, igrid) bind(C, name='ramses_amd_amrres_set_uold') result(rc)
       import :: ramses_amd_hydro_params, c_int
       type(ramses_amd_hydro_params), intent(in) :: p
       integer(c_int), value :: ngrid
       integer(c_int) :: igrid(*)
       integer(c_int) :: rc
     end function ramses_amd_amrres_set_uold
     function ramses_amd_amrres_upload_fine(p, ngrid, igrid, interpol_var) bind(C, name='ramses_amd_amrres_upload_fine') result(rc)
       import :: ramses_amd_hydro_params, c_int
       type(ramses_amd_hydro_params), intent(in) :: p
       integer(c_int), value :: ngrid, interpol_var
       integer(c_int) :: igrid(*)
       integer(c_int) :: rc
     end function ramses_amd_amrres_upload_fine
     function ramses_amd_amrres_load_f(ngrid, igrid, f) bind(C, name='ramses_amd_amrres_load_f') result(rc)
       import :: c_int, c_double
       integer(c_int), value :: ngrid
       integer(c_int) :: igrid(*)
       real(c_double) :: f(*)
       integer(c_int) :: rc
     end function ramses_amd_amrres_load_f
     function ramses_amd_amrres_sync_f(ngrid, igrid, f) bind(C, name='ramses_amd_amrres_sync_f') result(rc)
       import :: c_int, c_double
       integer(c_int), value :: ngrid
       integer(c_int) :: igrid(*)
       real(c_double) :: f(*)
       integer(c_int) :: rc
     end function ramses_amd_amrres_sync_f
     function ramses_amd_amrres_compare_f(ngrid, igrid, f, maxdiff, ndiff) bind(C, name='ramses_amd_amrres_compare_f') result(rc)
       import :: c_int, c_double, c_int64_t
       integer(c_int), value :: ngrid
       integer(c_int) :: igrid(*)
       real(c_double) :: f(*)
       real(c_double) :: maxdiff
       integer(c_int64_t) :: ndiff
       integer(c_int) :: rc
     end function ramses_amd_amrres_compare_f
     function ramses_amd_amrres_f_traffic(out2) bind(C, name='ramses_amd_amrres_f_traffic') result(rc)
       import :: c_int, c_int64_t
       integer(c_int64_t) :: out2(2)
       integer(c_int) :: rc
     end function ramses_amd_amrres_f_traffic
     function ramses_amd_amrres_sync_density(ngrid, igrid, uold) bind(C, name='ramses_amd_amrres_sync_density') result(rc)
       import :: c_int, c_double
       integer(c_int), value :: ngrid
       integer(c_int) :: igrid(*)
       real(c_double) :: uold(*)
       integer(c_int) :: rc
     end function ramses_amd_amrres_sync_density
     function ramses_amd_amrres_synchro(p, ngrid, igrid, dteff) bind(C, name='ramses_amd_amrres_synchro') result(rc)
       import :: ramses_amd_hydro_params, c_int, c_double
       type(ramses_amd_hydro_params), intent(in) :: p
       integer(c_int), value :: ngrid
       integer(c_int) :: igrid(*)
       real(c_double), value :: dteff
       integer(c_int) :: rc
     end function ramses_amd_amrres_synchro
     function ramses_amd_amrres_set_uold_grav(p, ngrid, igrid, dt) bind(C, name='ramses_amd_amrres_set_uold_grav') result(rc)
       import :: ramses_amd_hydro_params, c_int, c_double
       type(ramses_amd_hydro_params), intent(in) :: p
       integer(c_int), value :: ngrid
       integer(c_int) :: igrid(*)
       real(c_double), value :: dt
       integer(c_int) :: rc
     end function ramses_amd_amrres_set_uold_grav
     function ramses_amd_amrres_enable_pfix() bind(C, name='ramses_amd_amrres_enable_pfix') result(rc)
       import :: c_int
       integer(c_int) :: rc
     end function ramses_amd_amrres_enable_pfix
     function ramses_amd_amrres_set_unew_pfix(p, ngrid, igrid) bind(C, name='ramses_amd_amrres_set_unew_pfix') result(rc)
       import :: ramses_amd_hydro_params, c_int
       type(ramses_amd_hydro_params), intent(in) :: p
       integer(c_int), value :: ngrid
       integer(c_int) :: igrid(*)
       integer(c_int) :: rc
     end function ramses_amd_amrres_set_unew_pfix
     function ramses_amd_amrres_set_uold_pfix(p, ngrid, igrid, dt, dx_loc, beta_fix, hexp) &
          & bind(C, name='ramses_amd_amrres_set_uold_pfix') result(rc)
       import :: ramses_amd_hydro_params, c_int, c_double
       type(ramses_amd_hydro_params), intent(in) :: p
       integer(c_int), value :: ngrid
       integer(c_int) :: igrid(*)
       real(c_double), value :: dt, dx_loc, beta_fix, hexp
       integer(c_int) :: rc
     end function ramses_amd_amrres_set_uold_pfix
     function ramses_amd_amrres_courant(p, ngrid, igrid, dx, dt_in, out4) bind(C, name='ramses_amd_amrres_courant') result(rc)
       import :: ramses_amd_hydro_params, c_int, c_double
       type(ramses_amd_hydro_params), intent(in) :: p
       integer(c_int), value :: ngrid
       integer(c_int) :: igrid(*)
       real(c_double), value :: dx, dt_in
       real(c_double) :: out4(4)
       integer(c_int) :: rc
     end function ramses_amd_amrres_courant
     function ramses_amd_amrres_xg(xg) bind(C, name='ramses_amd_amrres_xg') result(rc)
       import :: c_int, c_double
       real(c_double) :: xg(*)
       integer(c_int) :: rc
     end function ramses_amd_amrres_xg
     function ramses_amd_amrres_rho_fine(p, ilevel, nlevelmax, levelmin, nvector, first, igrid_all, boxlen_over_nx, rho, mp4) &
          & bind(C, name='ramses_amd_amrres_rho_fine') result(rc)
       import :: ramses_amd_hydro_params, c_int, c_double
       type(ramses_amd_hydro_params), intent(in) :: p
       integer(c_int), value :: ilevel, nlevelmax, levelmin, nvector
       integer(c_int) :: first(*), igrid_all(*)
       real(c_double), value :: boxlen_over_nx
       real(c_double) :: rho(*), mp4(4)
       integer(c_int) :: rc
     end function ramses_amd_amrres_rho_fine
     function ramses_amd_amrres_rho_mpi_multipole(p, ilevel, n_own, n_all, igrid_all, boxlen_over_nx) &
          & bind(C, name='ramses_amd_amrres_rho_mpi_multipole') result(rc)
       import :: ramses_amd_hydro_params, c_int, c_double
       type(ramses_amd_hydro_params), intent(in) :: p
       integer(c_int), value :: ilevel, n_own, n_all
       integer(c_int) :: igrid_all(*)
       real(c_double), value :: boxlen_over_nx
       integer(c_int) :: rc
     end function ramses_amd_amrres_rho_mpi_multipole
     function ramses_amd_amrres_rho_mpi_deposit(ilevel, nvector, boxlen_over_nx) &
          & bind(C, name='ramses_amd_amrres_rho_mpi_deposit') result(rc)
       import :: c_int, c_double
       integer(c_int), value :: ilevel, nvector
       real(c_double), value :: boxlen_over_nx
       integer(c_int) :: rc
     end function ramses_amd_amrres_rho_mpi_deposit
     function ramses_amd_amrres_rho_mpi_finish(ilevel, levelmin, nvector, igrid_all, rho, mp4) &
          & bind(C, name='ramses_amd_amrres_rho_mpi_finish') result(rc)
       import :: c_int, c_double
       integer(c_int), value :: ilevel, levelmin, nvector
       integer(c_int) :: igrid_all(*)
       real(c_double) :: rho(*), mp4(4)
       integer(c_int) :: rc
     end function ramses_amd_amrres_rho_mpi_finish
     function ramses_amd_amrres_hydro_flag(p, ngrid, igrid, egd, egp, egu, fld, flp, flu, cells, ncells) &
          & bind(C, name='ramses_amd_amrres_hydro_flag') result(rc)
       import :: ramses_amd_hydro_params, c_int, c_double
       type(ramses_amd_hydro_params), intent(in) :: p
       integer(c_int), value :: ngrid
       integer(c_int) :: igrid(*), cells(*)
       integer(c_int), intent(out) :: ncells
       real(c_double), value :: egd, egp, egu, fld, flp, flu
       integer(c_int) :: rc
     end function ramses_amd_amrres_hydro_flag
     ! ---- AMR residency under MPI: the virtual-boundary exchanges on the resident cell vectors ----
     function ramses_amd_which_column(xx, base, ncell, ncol) bind(C, name='ramses_amd_which_column') result(k)
       import :: c_int, c_int64_t, c_double
       real(c_double) :: xx(*), base(*)
       integer(c_int64_t), value :: ncell
       integer(c_int), value :: ncol
       integer(c_int) :: k
     end function ramses_amd_which_column
     function ramses_amd_amrres_comm_epoch(ilevel) bind(C, name='ramses_amd_amrres_comm_epoch') result(e)
       import :: c_int
       integer(c_int), value :: ilevel
       integer(c_int) :: e
     end function ramses_amd_amrres_comm_epoch
     function ramses_amd_amrres_comm_set(ilevel, epoch, ncpu, em_n, em_ig, rc_n, rc_ig) &
          & bind(C, name='ramses_amd_amrres_comm_set') result(rc)
       import :: c_int
       integer(c_int), value :: ilevel, epoch, ncpu
       integer(c_int) :: em_n(*), em_ig(*), rc_n(*), rc_ig(*)
       integer(c_int) :: rc
     end function ramses_amd_amrres_comm_set
     function ramses_amd_amrres_zero_unew_virtual(ilevel) bind(C, name='ramses_amd_amrres_zero_unew_virtual') result(rc)
       import :: c_int
       integer(c_int), value :: ilevel
       integer(c_int) :: rc
     end function ramses_amd_amrres_zero_unew_virtual
     function ramses_amd_amrres_halo_rccl(ilevel, dir, myid) bind(C, name='ramses_amd_amrres_halo_rccl') result(rc)
       import :: c_int
       integer(c_int), value :: ilevel, dir, myid
       integer(c_int) :: rc
     end function ramses_amd_amrres_halo_rccl
     function ramses_amd_amrres_halo_stage_out(ilevel, dir, ncpu, h_send_addr, h_recv_addr, send_off, recv_off) &
          & bind(C, name='ramses_amd_amrres_halo_stage_out') result(rc)
       import :: c_int, c_int64_t, c_ptr
       integer(c_int), value :: ilevel, dir, ncpu
       type(c_ptr) :: h_send_addr, h_recv_addr        ! int64_t* on the C side: the addresses of the pinned buffers
       integer(c_int64_t) :: send_off(*), recv_off(*)
       integer(c_int) :: rc
     end function ramses_amd_amrres_halo_stage_out
     function ramses_amd_amrres_halo_stage_in(ilevel, dir) bind(C, name='ramses_amd_amrres_halo_stage_in') result(rc)
       import :: c_int
       integer(c_int), value :: ilevel, dir
       integer(c_int) :: rc
     end function ramses_amd_amrres_halo_stage_in
     function ramses_amd_amrres_godunov(p, ilevel, ngrid, igrid, dx, dt, nvector, interpol_var, interpol_type) &
          & bind(C, name='ramses_amd_amrres_godunov') result(rc)
       import :: ramses_amd_hydro_params, c_int, c_double
       type(ramses_amd_hydro_params), intent(in) :: p
       integer(c_int), value :: ilevel, ngrid, nvector, interpol_var, interpol_type
       integer(c_int) :: igrid(*)
       real(c_double), value :: dx, dt
       integer(c_int) :: rc
     end function ramses_amd_amrres_godunov
  end interface
end module ramses_amd_cabi
