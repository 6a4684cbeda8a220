/*
 * ramses_amd.h -- C ABI of libramses_amd.so, the MI355X (gfx950) native
 * implementation of the RAMSES per-level hot path: Godunov hydro sweep,
 * multigrid Poisson smoother and virtual-boundary halo packing.
 *
 * Plain C types only (pointers, sizes, PODs): this is what the reference's
 * Fortran side binds through ISO_C_BINDING (see ramses_amd/patch/ and
 * INTEGRATION.md) and what the Python host mirror binds through ctypes.
 *
 * Every entry point returns 0 on success, a negative RAMSES_AMD_E* code on
 * failure; ramses_amd_last_error() returns a description.  The reference has
 * no error returns on this path (it prints and calls clean_stop ->
 * MPI_ABORT, amr/end.f90:26-46); the Fortran shims call clean_stop when a
 * status is non-zero.  There is NO CPU fallback anywhere behind this ABI.
 *
 * Device layout ("level brick").  A fully refined level (or one rank's
 * Hilbert/octant share of it) is stored as one dense SoA block per conserved
 * variable:      u[ivar][k][j][i]     (i fastest, FP64)
 * with variable order rho, rho*u, rho*v, rho*w, E (hydro/condinit.f90:17-20)
 * and optional ghost layers of width g (0 or >=2 cells) on every side.  With
 * g = 0 the sweep wraps periodically in-kernel (single-rank periodic box);
 * with g >= 2 the ghost cells must have been filled (halo exchange or
 * physical boundary fill) before the sweep, exactly like the reference's
 * virtual-boundary octs (amr/virtual_boundaries.f90:373-528).
 */
#ifndef RAMSES_AMD_H
#define RAMSES_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RAMSES_AMD_OK 0
#define RAMSES_AMD_EINVAL (-1)      /* bad argument                          */
#define RAMSES_AMD_EUNSUPPORTED (-2)/* knob combination not implemented      */
#define RAMSES_AMD_EHIP (-3)        /* HIP runtime error                     */
#define RAMSES_AMD_ENODEVICE (-4)   /* no gfx950 device visible              */

/* riemann= of &HYDRO_PARAMS (hydro/umuscl.f90:791-804) */
enum { RAMSES_AMD_RIEMANN_LLF = 0, RAMSES_AMD_RIEMANN_HLLC = 1,
       RAMSES_AMD_RIEMANN_HLL = 2, RAMSES_AMD_RIEMANN_ACOUSTIC = 3,
       RAMSES_AMD_RIEMANN_EXACT = 4 };
/* scheme= of &HYDRO_PARAMS (hydro/umuscl.f90:73-94) */
enum { RAMSES_AMD_SCHEME_MUSCL = 0, RAMSES_AMD_SCHEME_PLMDE = 1 };

/* Solver knobs = the &HYDRO_PARAMS namelist group
 * (hydro/read_hydro_params.f90:43-54, defaults hydro/hydro_parameters.f90:75-89).
 * Replaces the module variables of hydro_parameters that the reference's
 * unsplit() reads implicitly. */
typedef struct ramses_amd_hydro_params {
  int32_t ndim;           /* NDIM of the RAMSES build (3 supported on device) */
  int32_t nvar;           /* NVAR (= ndim+2, NENER=0)                         */
  double gamma;
  double smallr;
  double smallc;
  int32_t slope_type;     /* 1 minmod, 2 moncen, 3 positivity, 7 van Leer, 8 theta */
  int32_t riemann;        /* RAMSES_AMD_RIEMANN_*                             */
  double slope_theta;
  int32_t scheme;         /* RAMSES_AMD_SCHEME_*                              */
  int32_t niter_riemann;
  double difmag;
  double courant_factor;
  /* arithmetic mode: 0 = strict (operation order of the reference, no FMA
   * contraction, IEEE division: bit-identical to the reference's x86-64
   * build), 1 = fast (FMA contraction + reciprocal reuse; <=1e-12 relative
   * L-infinity of the strict path, the tolerance north_star states). */
  int32_t fast_math;
  int32_t reserved;
} ramses_amd_hydro_params;

/* Geometry of one level brick on the device. */
typedef struct ramses_amd_brick {
  int32_t nx, ny, nz;     /* interior cells per direction                     */
  int32_t ng;             /* ghost width on every side: 0 (periodic wrap) or >=2 */
  int64_t pitch_y;        /* stride between j rows, in doubles                */
  int64_t pitch_z;        /* stride between k planes, in doubles              */
  int64_t pitch_var;      /* stride between variables, in doubles             */
} ramses_amd_brick;

/* Fills pitches for a densely packed brick of (n+2*ng)^3 cells. */
void ramses_amd_brick_dense(ramses_amd_brick *b, int nx, int ny, int nz, int ng);

const char *ramses_amd_last_error(void);
/* ABI guard for foreign-language bindings: pass the binding's own sizeof of
 * the two structs; returns 0 when they match this library's layout. */
int ramses_amd_abi_check(size_t sizeof_hydro_params, size_t sizeof_brick);
/* Number of HIP devices; fills name (<=255 chars) of device 0. */
/* MPI: one rank per GPU.  Selects device (local rank) mod (device count); the local rank
 * comes from the launcher's environment (OMPI_COMM_WORLD_LOCAL_RANK, MPI_LOCALRANKID,
 * PMI_LOCAL_RANK, SLURM_LOCALID, LOCAL_RANK) or, failing that, from world_rank. */
int ramses_amd_set_device_auto(int world_rank);
int ramses_amd_device_info(char *name, size_t name_len, int *n_cu, size_t *hbm_bytes);

/* ---------------------------------------------------------------------------
 * godunov_fine(ilevel) on a fully refined level brick.
 * Replaces: hydro/godunov_fine.f90:5-35 (godunov_fine) + :486-911 (godfine1)
 *           + hydro/umuscl.f90:22-171 (unsplit) and everything it calls, and
 *           fuses set_unew (hydro/godunov_fine.f90:40-130): on exit
 *           unew = uold + sum_d (F_left - F_right)   (x, then y, then z).
 * d_uold, d_unew: device pointers, brick layout b (distinct buffers).
 * d_grav: device pointer to the gravitational acceleration f(:,1:ndim) in the
 *         same brick layout with ndim variables, or NULL when poisson=.false.
 * dx = cell size of the level, dt = dtnew(ilevel).
 * stream: hipStream_t (as void*), NULL = default stream.  Asynchronous.
 * ------------------------------------------------------------------------- */
int ramses_amd_godunov_brick(const ramses_amd_hydro_params *p,
                             const ramses_amd_brick *b, const double *d_uold,
                             const double *d_grav, double *d_unew, double dx,
                             double dt, void *stream);

/* The same sweep split in two launches sets for communication/computation
 * overlap (the reference has none: make_virtual_fine_dp runs after the sweep,
 * amr/amr_step.f90:388-510): _shell updates the tiles and planes that hold the
 * cells within 2 of a brick face (what the neighbour ranks receive as ghost
 * octs), _interior everything else.  shell + interior == ramses_amd_godunov_brick
 * bit for bit; the halo exchange of the new state can start after _shell. */
int ramses_amd_godunov_brick_shell(const ramses_amd_hydro_params *p,
                                   const ramses_amd_brick *b, const double *d_uold,
                                   const double *d_grav, double *d_unew, double dx,
                                   double dt, void *stream);
int ramses_amd_godunov_brick_interior(const ramses_amd_hydro_params *p,
                                      const ramses_amd_brick *b, const double *d_uold,
                                      const double *d_grav, double *d_unew, double dx,
                                      double dt, void *stream);

/* Tuning knobs of the sweep (tile rows per workgroup, planes per z-chunk);
 * 0 keeps the built-in default.  Results do not depend on them. */
int ramses_amd_godunov_tune(int tile_rows, int zchunk);

/* ---------------------------------------------------------------------------
 * courant_fine(ilevel) on a level brick: the CFL time step.
 * Replaces: hydro/courant_fine.f90:1-159 + cmpdt hydro/godunov_utils.f90:5-120.
 * d_out: device pointer to 4 doubles {dt, mass, e_kin+e_int (total energy),
 *        e_int}; dt = min over cells, the sums are over interior cells * dx^3
 *        (courant_fine.f90:100-124).  d_out must be initialised by the caller
 *        through ramses_amd_courant_init().  Asynchronous on stream.
 * ------------------------------------------------------------------------- */
int ramses_amd_courant_init(const ramses_amd_hydro_params *p, double dx,
                            double *d_out, void *stream);
int ramses_amd_courant_brick(const ramses_amd_hydro_params *p,
                             const ramses_amd_brick *b, const double *d_uold,
                             const double *d_grav, double dx, double *d_out,
                             void *stream);

/* ---------------------------------------------------------------------------
 * Periodic ghost fill of a brick with ng>=2 from its own interior (the
 * single-rank limit of make_virtual_fine_dp, amr/virtual_boundaries.f90:373-528:
 * on one rank a periodic neighbour oct IS the oct on the other side).
 * axes: bit 0 = x, bit 1 = y, bit 2 = z.  nvar variables.
 * ------------------------------------------------------------------------- */
int ramses_amd_fill_ghosts_periodic(const ramses_amd_brick *b, double *d_u,
                                    int nvar, int axes, void *stream);

/* ---------------------------------------------------------------------------
 * Halo pack / unpack for the multi-rank exchange.
 * Replaces the pack (amr/virtual_boundaries.f90:454-464) and unpack
 * (:492-506) loops of make_virtual_fine_dp, with all nvar fields fused in one
 * message per peer.  A face slab is the 2-cell-thick (= one oct) layer next
 * to face `face` (0:-x 1:+x 2:-y 3:+y 4:-z 5:+z).  pack copies the INTERIOR
 * slab adjacent to the face into d_buf; unpack copies d_buf into the GHOST
 * slab beyond the face.  Slabs span the full allocated extent (ghosts
 * included) of the axes already exchanged, so exchanging x, then y, then z
 * also fills edges and corners (the 26-neighbour stencil of godfine1).
 * Returns the number of doubles in the slab (also when d_buf is NULL).
 * ------------------------------------------------------------------------- */
int64_t ramses_amd_halo_slab_size(const ramses_amd_brick *b, int nvar, int face);
int ramses_amd_halo_pack(const ramses_amd_brick *b, const double *d_u, int nvar,
                         int face, double *d_buf, void *stream);
int ramses_amd_halo_unpack(const ramses_amd_brick *b, double *d_u, int nvar,
                           int face, const double *d_buf, void *stream);

/* ---------------------------------------------------------------------------
 * multigrid_fine(ilevel,icount) + force_fine on a fully refined PERIODIC level
 * (n = 2^level cells per direction, dense brick phi[k][j][i], no ghosts).
 * Replaces: poisson/multigrid_fine_commons.f90:25-296 (V-cycle driver,
 *   recursive_multigrid_coarse :307-390), the operators of
 *   poisson/multigrid_fine_fine.f90 (gauss_seidel_mg_fine :332-451,
 *   cmp_residual_mg_fine :147-249, cmp_residual_norm2_fine :254-287,
 *   restrict_residual_fine_reverse :528-590, interpolate_and_correct_fine
 *   :596-698) and their coarse twins in multigrid_fine_coarse.f90, with the
 *   per-solve communicator construction (build_parent_comms_mg) replaced by
 *   arithmetic indexing; make_fine_bc_rhs (:1058-1159) on an unmasked level.
 * d_rho: source (rho), rho_tot: box mean, fourpi = 2*twopi*scale (or the
 *   cosmological 1.5*omega_m*aexp*scale) as the reference computes it.
 * d_phi: in = first guess (zero at levelmin), out = potential.
 * d_f1 / d_f2: the reference's f(:,1) (minus residual) and f(:,2) (RHS).
 * d_work: ramses_amd_mg_workspace_doubles(level) doubles of device scratch
 *   (the coarse hierarchy active_mg(:,l)%u(:,1:3) for l = 1..level-1).
 * safe_mode: in/out, the reference's safe_mode(ilevel).
 * iters / err: V-cycles done and the final error
 *   sqrt(res^2/(res0^2+1e-20 rho_tot^2)); constants MAXITER=10, ngs=2.
 * Synchronises the stream once per V-cycle (the convergence test).
 * ------------------------------------------------------------------------- */
int64_t ramses_amd_mg_workspace_doubles(int level);
int ramses_amd_multigrid_fine_brick(int level, const double *d_rho, double rho_tot,
                                    double fourpi, double epsilon, int *safe_mode,
                                    double *d_phi, double *d_f1, double *d_f2,
                                    double *d_work, int *iters, double *err,
                                    void *stream);
/* gradient_phi (poisson/force_fine.f90:199-324): d_f holds f(:,1:3), 3*n^3. */
int ramses_amd_gradient_phi_brick(int level, const double *d_phi, double *d_f,
                                  void *stream);
/* The individual operators (n^3 periodic bricks), exposed for parity tests. */
int ramses_amd_mg_gauss_seidel(double *d_phi, const double *d_rhs, int n, double dx2,
                               int redstep, void *stream);
int ramses_amd_mg_residual(const double *d_phi, const double *d_rhs, double *d_res,
                           int n, double dx, double *d_work, double *d_norm2,
                           void *stream);
int ramses_amd_mg_restrict(const double *d_res_f, double *d_rhs_c, double *d_u1_c,
                           int nf, void *stream);
int ramses_amd_mg_interp_correct(double *d_phi_f, const double *d_corr_c, int nf,
                                 void *stream);
/* Fused smoother: npass (2 or 4) red/black colour passes from d_phi_in into
 * d_phi_out (distinct buffers) in one time-skewed pass; with d_res != NULL also
 * the residual, with d_norm2 != NULL its dx^3-scaled squared norm.  Same values
 * as npass calls of ramses_amd_mg_gauss_seidel (+ ramses_amd_mg_residual). */
int ramses_amd_mg_smooth_fused(const double *d_phi_in, double *d_phi_out,
                               const double *d_rhs, double *d_res, double *d_work,
                               double *d_norm2, int n, double dx, int npass, void *stream);
/* How multigrid_fine smooths levels with n>=64.  1 (default): fused smoother, 2+2 colour passes per
 * smoothing step on 32-row tiles; 12|16|24|32: the same on that many tile rows; 4: one launch of 4
 * colour passes (24-row tiles); 0: one kernel per colour pass.  Results do not depend on it. */
int ramses_amd_mg_tune(int fused);

/* ---------------------------------------------------------------------------
 * Coarse <-> fine hydro operators between a periodic coarse brick (nc^3) and
 * its fully refined child brick ((2nc)^3), brick layout, nvar = 5.
 * ramses_amd_interpol_hydro_brick replaces interpol_hydro + compute_limiter_minmod
 *   / compute_limiter_central / compute_central (hydro/interpol_hydro.f90:268-444,
 *   449-637): every coarse cell and its 6 neighbours -> its 8 children, with the
 *   &REFINE_PARAMS knobs interpol_var (0 conservative, 1 internal energy,
 *   2 velocity + internal energy) and interpol_type (1 minmod, 2 central-limited,
 *   3 unlimited central, 4 = 3 for velocities and 2 otherwise).
 * ramses_amd_upload_fine_brick replaces upload_fine / upl (:5-68, 73-263): every
 *   coarse cell = mean of its 8 children (density floored), internal-energy
 *   averaging when interpol_var is 1 or 2.
 * ------------------------------------------------------------------------- */
int ramses_amd_interpol_hydro_brick(int nc, int nvar, int interpol_var, int interpol_type,
                                    double smallr, const double *d_coarse, double *d_fine,
                                    void *stream);
int ramses_amd_upload_fine_brick(int nc, int nvar, int interpol_var, double smallr,
                                 const double *d_fine, double *d_coarse, void *stream);

/* ---------------------------------------------------------------------------
 * godunov_fine(ilevel) on the reference's OWN arrays (host memory, Fortran
 * layout): the entry point the Fortran shim ramses_amd/patch/godunov_fine.f90
 * binds.  Replaces hydro/godunov_fine.f90:5-35 + :486-911 for a level that is
 * fully refined on this rank (periodic box, nx=ny=nz=1).
 *   igrid[ngrid]      = active(ilevel)%igrid(1:ngrid)        (1-based oct slots)
 *   xg                = xg(1:ngridmax,1:3)                    (oct centres)
 *   uold, unew        = (1:ncell,1:nvar), ncell = ncoarse+8*ngridmax
 *   f                 = f(1:ncell,1:3) or NULL when poisson=.false.
 * Cell addressing icell = ncoarse+(ind-1)*ngridmax+igrid (hydro/godunov_fine.f90:600).
 * On exit unew(active cells of the level) holds uold + flux differences, as
 * after the reference's set_unew + godunov_fine.  Synchronous.  Anything the
 * device path does not cover (AMR level, several ranks, nx>1) is an error:
 * there is no CPU fallback behind this entry.
 * ------------------------------------------------------------------------- */
int ramses_amd_godunov_fine_host(const ramses_amd_hydro_params *p, int ilevel, int ngrid,
                                 const int *igrid, const double *xg, int64_t ngridmax,
                                 int64_t ncoarse, int nx_loc, const double *uold,
                                 double *unew, const double *f, double dx, double dt);

/* Same, for Fortran callers that cannot pass a NULL array: f_or_dummy must be a
 * valid address and is read only when has_f != 0. */
int ramses_amd_godunov_fine_f90(const ramses_amd_hydro_params *p, int ilevel, int ngrid,
                                const int *igrid, const double *xg, int64_t ngridmax,
                                int64_t ncoarse, int nx_loc, const double *uold,
                                double *unew, const double *f_or_dummy, int has_f,
                                double dx, double dt);

/* godunov_fine(ilevel) of an NDIM=1 / NDIM=2 build of the reference on its own arrays (hydro/godunov_fine.f90:5-35 with
 * twotondim = 2 / 4 cells per oct; BASELINE config C1 = namelist/sedov1d.nml on one uniform level): a fully refined level
 * without finer octs, one rank, hydro variables only (NVAR = NDIM + 2).  igrid = active(ilevel)%igrid; igrid_bound = the octs
 * of boundary(1:nboundary,ilevel) (their cells were filled by make_boundary_hydro, amr/amr_step.f90:293: they become the
 * ghost cells of the level's brick; directions without boundary octs are periodic); xg(1:ngridmax,1:ndim) in coarse-cell
 * units, skip = (icoarse_min, jcoarse_min), nloc = interior coarse cells per direction.  The problem is embedded in a 3-D
 * brick (ny and/or nz = 1) and swept by the dense kernel: bit-identical to the NDIM=1/2 reference in strict mode.  On exit
 * unew(active cells) = uold + flux differences.  AMR levels of such builds are refused (the caller keeps the reference's
 * routine for them). */
int ramses_amd_godunov_fine_lowdim_f90(const ramses_amd_hydro_params *p, int ilevel, int ngrid, const int *igrid, int nbound,
                                       const int *igrid_bound, const double *xg, int64_t ngridmax, int64_t ncoarse, const int *skip,
                                       const int *nloc, const double *uold, double *unew, double dx, double dt);
/* NDIM<3 builds: the drop-in reports every level it hands to the reference's host godunov_fine (a level that is not uniform,
 * gravity, ..., or a refusal of the entry above with RAMSES_AMD_EUNSUPPORTED); the library counts its own sweeps and prints both,
 * level by level, in one line at exit. */
int ramses_amd_lowdim_note_reference(int ilevel);
int64_t ramses_amd_lowdim_device_sweeps(void);
int64_t ramses_amd_lowdim_reference_sweeps(void);

/* ---------------------------------------------------------------------------
 * multigrid_fine(ilevel,icount) on the reference's OWN arrays: the entry point
 * ramses_amd/patch/multigrid_fine_commons.f90 binds.  Replaces
 * poisson/multigrid_fine_commons.f90:25-296 at levelmin of a periodic
 * single-rank run (first guess phi=0 as make_multipole_phi sets it, all cells
 * unmasked).  rho, phi = (1:ncell) cell vectors; on exit phi of the level's
 * cells holds the potential.  safe_mode in/out (0/1), iters/err as printed by
 * the reference ('==> Level= Step= Error=').  Synchronous; no CPU fallback.
 * ------------------------------------------------------------------------- */
int ramses_amd_multigrid_fine_f90(int ilevel, int ngrid, const int *igrid, const double *xg,
                                  int64_t ngridmax, int64_t ncoarse, int nx_loc,
                                  const double *rho, double *phi, double rho_tot,
                                  double fourpi, double epsilon, int *safe_mode,
                                  int *iters, double *err);

/* One-shot halo: up to 26 regions (faces, edges, corners) of a ghost-layer brick packed into /
 * unpacked from one buffer by a single launch; the exchange is then one grouped send/recv with
 * one message per peer (pack/unpack loops of make_virtual_fine_dp, amr/virtual_boundaries.f90:454-506).
 * boxes: nbox x {org x,y,z (allocated coordinates), ext x,y,z}; offsets (doubles) into d_buf. */
int ramses_amd_halo_multi(const ramses_amd_brick *b, double *d_u, int nvar, int nbox, const int *boxes,
                          const int64_t *offsets, double *d_buf, int pack, void *stream);

/* make_boundary_hydro (hydro/hydro_boundary.f90:5-269) on a ghost-layer brick: fills the
 * ghost layers of one face (0:-x 1:+x 2:-y 3:+y 4:-z 5:+z) over the full extent of the
 * other directions.  bound_type is the reference's code: face+1 reflexive, 10+face+1
 * outflow, 20+face+1 imposed (conserved state `imposed[nvar]`, boundana).  Call the faces
 * in x, y, z order, after the periodic/halo fill of the other faces. */
int ramses_amd_make_boundary_hydro(const ramses_amd_hydro_params *p, const ramses_amd_brick *b, double *d_uold,
                                   int face, int bound_type, const double *imposed, int no_inflow, void *stream);

/* ---------------------------------------------------------------------------
 * godunov_fine(ilevel) on an AMR level (hydro/godunov_fine.f90:5-35, godfine1
 * :486-911 with every AMR branch: stencil cells of missing octs interpolated
 * from the father level (get3cubefather amr/nbors_utils.f90:5-194,
 * getnborfather :404-525, interpol_hydro hydro/interpol_hydro.f90:268-444),
 * fluxes zeroed at refined interfaces (:720-747), unew += flux differences
 * (:751-792), corrections of the coarser level's leaf cells (:798-908)).
 * Arrays are the reference's own, by address: son(1:ncell),
 * nbor(1:ngridmax,1:6), father(1:ngridmax), uold/unew(1:ncell,1:nvar).
 * f_or_null = f(1:ncell,1:3) when poisson (the gravity predictor of ctoprim; missing octs take
 * the father cell's f, :637-647), NULL otherwise.
 * divu_or_null / enew_or_null = divu(1:ncell), enew(1:ncell) when pressure_fix (updated with cmpflxm's
 * normal-velocity and internal-energy fluxes, :771-786 and the coarse-level twins :825-905), NULL otherwise.
 * nvector = the reference build's NVECTOR: it fixes the order in which the
 * coarse-level corrections are accumulated (bit parity).  Needs ilevel >= 3.
 * ------------------------------------------------------------------------- */
int ramses_amd_godunov_fine_amr_host(const ramses_amd_hydro_params *p, int ilevel, int ngrid,
                                     const int *igrid, const int *son, const int *nbor,
                                     const int *father, int64_t ngridmax, int64_t ncoarse,
                                     const double *uold, double *unew, const double *f_or_null,
                                     double *divu_or_null, double *enew_or_null, double dx, double dt,
                                     int nvector, int interpol_var, int interpol_type);
/* Fortran-friendly: f_or_dummy is always a valid array, read only when has_f != 0 */
int ramses_amd_godunov_fine_amr_f90(const ramses_amd_hydro_params *p, int ilevel, int ngrid,
                                    const int *igrid, const int *son, const int *nbor,
                                    const int *father, int64_t ngridmax, int64_t ncoarse,
                                    const double *uold, double *unew, const double *f_or_dummy, int has_f,
                                    double *divu_or_dummy, double *enew_or_dummy, int has_pfix,
                                    double dx, double dt, int nvector, int interpol_var, int interpol_type);
/* the same with every array already on the device; d_work holds
 * ramses_amd_godunov_fine_amr_workspace(ngrid, ngridmax) bytes, *d_err (zeroed by the
 * caller) counts tree inconsistencies */
int64_t ramses_amd_godunov_fine_amr_workspace(int ngrid, int64_t ngridmax);
int ramses_amd_godunov_fine_amr_device(const ramses_amd_hydro_params *p, int ilevel, int ngrid,
                                       const int *d_igrid, const int *d_son, const int *d_nbor,
                                       const int *d_father, int64_t ngridmax, int64_t ncoarse,
                                       const double *d_uold, double *d_unew, const double *d_grav_or_null,
                                       double *d_divu_or_null, double *d_enew_or_null, double dx, double dt,
                                       int nvector, int interpol_var, int interpol_type,
                                       void *d_work, int *d_err, void *stream);

/* ---------------------------------------------------------------------------
 * Distributed multigrid: one rank's nx x ny x nz brick of a periodic level
 * (power-of-two extents; the bricks of 2 or 4 ranks in the reference's cubic
 * box are not cubes) with ng ghost layers (pitches nx+2ng, ny+2ng), ghosts
 * filled by the halo exchange (make_virtual_mg_dp,
 * poisson/multigrid_fine_commons.f90:1172-1270, replaced by
 * ramses_amd_halo_pack/unpack + RCCL).  The fused smoother recomputes the
 * neighbours' updates inside its ghost layers, so ONE ng-wide exchange replaces
 * the exchange after every colour pass (multigrid_fine_commons.f90:197-202).
 * Same arithmetic and operation order as the dense entry points.
 * ------------------------------------------------------------------------- */
int ramses_amd_mg_smooth_fused_ghost(const double *d_phi_in, double *d_phi_out, const double *d_rhs,
                                     double *d_res, double *d_work, double *d_norm2, int nx, int ny, int nz, int ng,
                                     double dx, int npass, void *stream);
/* make_fine_bc_rhs (multigrid_fine_commons.f90:1058-1159), unmasked: f2 = fourpi*(rho-rho_tot) */
int ramses_amd_mg_rhs(const double *d_rho, double *d_f2, int64_t N, double fourpi, double rho_tot, void *stream);
/* restrict_residual_fine_reverse (multigrid_fine_fine.f90:528-590) on local bricks (nf*: the FINE brick) */
int ramses_amd_mg_restrict_ghost(const double *d_res_f, double *d_rhs_c, int nfx, int nfy, int nfz, int ngf, int ngc, void *stream);
/* interpolate_and_correct_fine (:596-698): coarse correction = local brick with >= 1 valid
 * ghost layer (cglob = 0) or a replicated dense periodic cglob^3 level whose cell
 * coarse_origin[] is this rank's first coarse cell */
int ramses_amd_mg_interp_correct_ghost(double *d_phi_f, int nfx, int nfy, int nfz, int ngf, const double *d_corr_c, int ngc,
                                       int cglob, const int *coarse_origin, void *stream);
/* gradient_phi (poisson/force_fine.f90:199-324) on a local brick, d_f dense [3][nz][ny][nx] */
int ramses_amd_gradient_phi_ghost(const double *d_phi, double *d_f, int nx, int ny, int nz, int ng, double dx, void *stream);
/* recursive_multigrid_coarse (multigrid_fine_commons.f90:307-390) on a dense periodic level */
int ramses_amd_mg_coarse_solve_dense(int level, const double *d_rhs, double *d_u1, double *d_work, int safe,
                                     void *stream);

/* ---------------------------------------------------------------------------
 * Multigrid on AMR levels (partially refined level, masked cells, Dirichlet
 * boundaries captured by the mask).  The reference's driver and per-solve setup
 * (multigrid_fine, recursive_multigrid_coarse, make_initial_phi, masks,
 * build_parent_comms_mg, scan flags: poisson/multigrid_fine_commons.f90) stay
 * host code; the compute routines they call are replaced one for one:
 *   gauss_seidel   gauss_seidel_mg_fine   multigrid_fine_fine.f90:332-451 / _coarse multigrid_fine_coarse.f90:411-593
 *   residual       cmp_residual_mg_fine   :147-249 / _coarse :167-329
 *   norm2          cmp_residual_norm2_fine :254-287
 *   restrict       restrict_residual_fine_reverse :528-590 / _coarse_reverse :692-764
 *                  (also zeroes the coarser level's rhs and correction, which the driver resets around it)
 *   interpolate    interpolate_and_correct_fine :596-698 / _coarse :769-886
 * begin(): the tree (son(1:ncell), nbor(1:ngridmax,1:6), father, lookup_mg), the fine level's
 * host arrays (flag2, phi(1:ncell), f(1:ncell,1:3)) and its active oct list; add_level(): a
 * multigrid level's active_mg(myid,l)%igrid, %u(1:8*ngrid,1:4), %f(1:8*ngrid,1).  All levels
 * then stay on the device until end() writes phi back.  RAMSES_AMD_MG_SYNC=1: every routine
 * reloads its inputs from the host arrays and writes its outputs back (debug).
 * ------------------------------------------------------------------------- */
int ramses_amd_mgamr_begin(int ilevel, int64_t ngridmax, int64_t ncoarse, const int *son, const int *nbor,
                           const int *father, const int *lookup_mg, const int *flag2, double *phi, double *f,
                           int ngrid, const int *igrid);
int ramses_amd_mgamr_add_level(int level, int ngrid, const int *igrid, double *u, const int *fscan);
/* MPI: a level is the calling rank's own octs followed by the reception octs of the other ranks (active_mg(icpu,l) for
 * icpu /= myid, whose values the reference's make_virtual_mg_dp / make_reverse_mg_dp keep current on the host): level_begin
 * announces the total, level_block passes one rank buffer (igrid, u(1:8*ngrid,1:4), f(1:8*ngrid,1)) -- the caller's own
 * first; only that one is updated, the restriction also adds into the others.  fine_active: the list given to begin() holds
 * nact active octs followed by the level's reception octs.  Several ranks imply RAMSES_AMD_MG_SYNC=1 semantics (every
 * routine exchanges its arrays with the host, where the reference's halo routines work). */
int ramses_amd_mgamr_level_begin(int level, int ngrid_total);
int ramses_amd_mgamr_level_block(int level, int ngrid, const int *igrid, double *u, const int *fscan);
int ramses_amd_mgamr_fine_active(int nact);
int ramses_amd_mgamr_force_sync(int on);
int ramses_amd_mgamr_gauss_seidel(int level, int redstep, int safe);
int ramses_amd_mgamr_residual(int level);
int ramses_amd_mgamr_norm2(int level, double *norm2);
int ramses_amd_mgamr_restrict(int finelevel);
int ramses_amd_mgamr_interpolate(int finelevel);
int ramses_amd_mgamr_end(void);
/* Several ranks, levels of the solve resident on the device (ramses_amd_mgamr_force_sync(0)): the virtual boundaries of
 * the solved level -- make_virtual_fine_dp(phi / f(:,1)), amr/virtual_boundaries.f90:373-528 -- and of the multigrid levels
 * -- make_virtual_mg_dp / make_reverse_mg_dp, poisson/multigrid_fine_commons.f90:1172-1290,1378-1475 -- exchanged from the
 * device.  comm_set: per peer icpu the emission list of the level (octs of emission(icpu,level)%igrid with list_is_octs = 1,
 * positions 1..n in the rank's own buffer as in emission_mg(icpu,level)%igrid with 0), concatenated in icpu order, and the
 * number of reception octs (their blocks follow the rank's own octs in the layout, in icpu order).  comp = 1..4 (u(:,1..4):
 * phi/correction, rhs, residual, mask), dir 0 forward, 1 reverse (added peer by peer in icpu order).  halo_rccl: one grouped
 * RCCL send/recv; halo_stage_out / _in: the caller's own MPI between them on pinned host buffers (message of peer icpu at
 * h_send + send_off[icpu-1]; addresses as integers for c_f_pointer).  stats: [0] bytes / [1] copies of level arrays that
 * crossed PCIe after the first routine of the solve, [2] bytes / [3] number of halo exchanges. */
int ramses_amd_mgamr_comm_set(int level, int ncpu, int myid, const int *em_n, const int *em_list, int list_is_octs, const int *rc_n);
int ramses_amd_mgamr_halo_stage_out(int level, int comp, int dir, int ncpu, int64_t *h_send_addr, int64_t *h_recv_addr,
                                    int64_t *send_off, int64_t *recv_off);
int ramses_amd_mgamr_halo_stage_in(int level, int comp, int dir);
int ramses_amd_mgamr_halo_rccl(int level, int comp, int dir);
int ramses_amd_mgamr_stats(int64_t *out4);

/* ---------------------------------------------------------------------------
 * multigrid_fine(ilevel,icount) on an AMR level of a periodic single-rank run, driver AND per-solve
 * setup on the device (csrc/pois_amr.hip): make_initial_phi + interpol_phi (poisson/phi_fine_cg.f90:452-521,
 * poisson/interpol_phi.f90:1-81), make_fine_mask, make_fine_bc_rhs (poisson/multigrid_fine_commons.f90:982-1159),
 * build_parent_comms_mg (:400-894), restrict_mask_fine/coarse_reverse (multigrid_fine_fine.f90:88-141,
 * multigrid_fine_coarse.f90:105-160), set_scan_flag_fine/_coarse (:705-771, :892-985), the iteration loop
 * (:176-282) and recursive_multigrid_coarse (:307-390).
 *   poisamr_tree       son(1:ncell), nbor(1:ngridmax,1:6), father(1:ngridmax); copied only when `epoch` (a counter the
 *                      caller advances whenever the tree may have changed) differs from the cached one
 *   poisamr_multigrid  igrid = active(ilevel)%igrid, igrid_c = active(ilevel-1)%igrid; phi, phi_old, rho = the
 *                      reference's cell vectors (host).  Read: rho on the level, phi and phi_old on the level above
 *                      (interp=1: first guess and boundary values interpolated, tfrac = dtnew(l)/dtold(l-1)*(icount-1));
 *                      written: phi on the level.  flag2 = the reference's work array flag2(1:ncell) (its element 1, not
 *                      0): read and updated on the level's cells the way set_scan_flag_fine does (the stale-flag
 *                      behaviour of :763-768 decides which cells take the scanning branch), flag2(1:n) receives the
 *                      coarse oct lists as in build_parent_comms_mg.  safe_mode in/out; iters/err = what the reference prints.
 * ------------------------------------------------------------------------- */
int ramses_amd_poisamr_tree(int epoch, int64_t ngridmax, int64_t ncoarse, const int *son, const int *nbor, const int *father);
int ramses_amd_poisamr_multigrid(int ilevel, int ngrid, const int *igrid, int ngrid_c, const int *igrid_c, double *phi,
                                 const double *phi_old, const double *rho, int *flag2, double rho_tot, double fourpi,
                                 double tfrac, int interp, double epsilon, int ngs_fine, int ngs_coarse,
                                 int ncycles_coarse_safe, int *safe_mode, int *iters, double *err);
int ramses_amd_poisamr_levelmin_mg(void);
/* force_fine(ilevel) on an AMR level of the same kind of run (poisson/force_fine.f90:5-194, gradient_phi :199-324 with
 * interpol_phi at the level's edge): f(1:ncell,1:3) written on the level's cells; diag[0] = the level's term of epot_tot,
 * diag[1] = rho_max(ilevel).  fresh=1: poisamr_multigrid has just solved this level (phi, rho and the level above are still
 * on the device), else they are read from the host vectors. */
int ramses_amd_poisamr_force(int ilevel, int ngrid, const int *igrid, int ngrid_c, const int *igrid_c, const double *phi,
                             const double *phi_old, const double *rho, double *f, double tfrac, int interp, int fresh,
                             double fact, double *diag);
/* The same with several ranks: igrid_all = the rank's ngrid_own own octs of the level followed by its reception octs (ngrid_all in
 * all), igrid_c_all likewise for the level above.  phi / rho of all of them are read from the host vectors (the virtual cells
 * carry what the solver's last make_virtual_fine_dp left there), f is written on the own octs' cells -- the caller's
 * make_virtual_fine_dp(f(1,idim),ilevel) follows (poisson/force_fine.f90:137-139) -- and diag holds the RANK's share of the two
 * diagnostics, before the caller's MPI_ALLREDUCEs (:181-186). */
int ramses_amd_poisamr_force_mpi(int ilevel, int ngrid_own, int ngrid_all, const int *igrid_all, int ngrid_c_all, const int *igrid_c_all,
                                 const double *phi, const double *phi_old, const double *rho, double *f, double tfrac, int interp,
                                 double fact, double *diag);
/* The same, f of the own cells going into the resident acceleration on the device instead of the host array
 * (ramses_amd_amrres_take_f_device); the caller exchanges the virtual octs on the device. */
int ramses_amd_poisamr_force_mpi_resident(int ilevel, int ngrid_own, int ngrid_all, const int *igrid_all, int ngrid_c_all,
                                          const int *igrid_c_all, const double *phi, const double *phi_old, const double *rho,
                                          double tfrac, int interp, double fact, double *diag);
/* create the HIP context and load the device code of every kernel now (one-time ~0.2 s, otherwise paid by the
 * first call of each kernel family inside the reference's timed loop) */
int ramses_amd_warmup(void);
/* RAMSES_AMD_PROFILE=1: accumulate wall time per shadowed routine and level (printed at exit) */
int ramses_amd_prof_add(const char *name, int level, double seconds);

/* -------------------------------------------------------------------------
 * phi_fine_cg(ilevel,icount) -- poisson/phi_fine_cg.f90:5-206: the iteration loop (:88-187)
 * of the conjugate-gradient Poisson solver with cmp_Ap_cg (:344-447), on one AMR level in the
 * reference's own arrays (host pointers): phi, f = (r, p, A p) as f(1:ncell,1:3), tree son/nbor,
 * the level's oct list.  The caller has done the reference's pre-loop steps (initial guess,
 * make_virtual_fine_dp, make_boundary_phi, cmp_residual_cg :52-85).  On return phi and f hold
 * what the reference's loop leaves; *iter = iterations, err[0] = last rms residual
 * (:186), err[1] = the first, err[2] = rhs_norm (:63-78, 0 if rho is NULL).
 * fact = fourpi*dx^2/6 (:45), ncell_level = twotondim*numbtot(1,ilevel).  One rank (several: the
 * ramses_amd_cgmpi_* routines below).  ordered = 1: the dot products are summed in the
 * reference's order by the parallel parity scan (bit-identical; the default); 2: in the same
 * order by one lane (slow; the scan's check); 0: fixed parallel tree (deterministic, equal to
 * rounding only -- CG amplifies it to ~1e-9 on f); < 0: as RAMSES_AMD_CG_ORDERED says
 * ("1", "chain", "0"; unset: 1).
 * ------------------------------------------------------------------------- */
int ramses_amd_cg_solve_host(int ilevel, int ngrid, const int *igrid, const int *son, const int *nbor,
                             int64_t ngridmax, int64_t ncoarse, double *phi, double *f, const double *rho_or_null,
                             double rho_tot, double fact, double ncell_level, double epsilon, int itermax,
                             int ordered, int *iter, double *err);
/* The strictly sequential sum  s <- fl(s + x[i]), i = 0..n-1  (the reference's accumulation loops), bit for bit, in
 * parallel (csrc/parity_scan.hpp): *d_out = the sum.  d_x, d_out, d_scratch are device pointers;
 * ramses_amd_ordered_sum_scratch(n) bytes of scratch. */
size_t ramses_amd_ordered_sum_scratch(int64_t n);
int ramses_amd_ordered_sum_device(const double *d_x, int64_t n, double *d_out, void *d_scratch, void *stream);
/* The same loop with several MPI ranks (the two MPI_ALLREDUCEs per iteration poisson/phi_fine_cg.f90:108,154 and the halo
 * exchange of p :134 stay with the caller): cgmpi_begin uploads the state (arguments as above; ngrid may be 0) and returns
 * out2 = {local rhs norm^2, local r.r}; cgmpi_step runs one routine of the loop body on the rank's octs -- 0: p = r + beta p,
 * 1: z = A p and the local p.z, 2: x += alpha p, r -= alpha z and the local r.r -- with alpha, beta formed on the device from
 * the device scalars (slot 0 r.r, 1 r.r of the previous iteration, 2 p.z), which the caller reads (cgmpi_get), reduces over the
 * ranks and writes back (cgmpi_set).  cgmpi_p_cells moves the cells of the listed octs of p between the device and the host
 * array f(:,2) (emission octs out before, reception octs in after the reference's make_virtual_fine_dp(f(1,2),ilevel)).
 * cgmpi_end brings phi and f back. */
int ramses_amd_cgmpi_begin(int ilevel, int ngrid, const int *igrid, const int *son, const int *nbor, int64_t ngridmax, int64_t ncoarse,
                           const double *phi, double *f, const double *rho_or_null, double rho_tot, double fact, int ordered,
                           double *out2);
int ramses_amd_cgmpi_get(int slot, double *val);
int ramses_amd_cgmpi_set(int slot, double val);
int ramses_amd_cgmpi_step(int step, int iter);
int ramses_amd_cgmpi_p_cells(int n, const int *igrid, int to_host);
int ramses_amd_cgmpi_end(double *phi, double *f);
/* The halo of p (make_virtual_fine_dp(f(1,2),ilevel), poisson/phi_fine_cg.f90:134) on the device vector: comm_set = the level's
 * emission / reception oct lists per peer (em_n / rc_n [ncpu], lists concatenated in icpu order), once per solve; then per
 * iteration either p_halo_rccl (one grouped RCCL send/recv) or p_halo_stage_out / the caller's MPI on the pinned buffers /
 * p_halo_stage_in (message of peer icpu at h_send + send_off[icpu-1]; addresses as integers for c_f_pointer). */
int ramses_amd_cgmpi_comm_set(int ncpu, const int *em_n, const int *em_ig, const int *rc_n, const int *rc_ig);
int ramses_amd_cgmpi_p_halo_stage_out(int ncpu, int64_t *h_send_addr, int64_t *h_recv_addr, int64_t *send_off, int64_t *recv_off);
int ramses_amd_cgmpi_p_halo_stage_in(void);
int ramses_amd_cgmpi_p_halo_rccl(void);

/* ---------------------------------------------------------------------------
 * Device-resident level.  For a fully refined periodic
 * level of a single-rank hydro-only run the state stays on the GPU across
 *   newdt_fine/courant_fine   hydro/courant_fine.f90:1-159
 *   set_unew                  hydro/godunov_fine.f90:40-130   (fused: no-op)
 *   godunov_fine              hydro/godunov_fine.f90:5-35
 *   set_uold                  hydro/godunov_fine.f90:135-232  (buffer swap)
 * and is written back to the reference's host array only on demand
 * (backup_hydro, hydro/output_hydro.f90:63-163, calls sync_host first).
 * The first call loads the level from the host array `uold`; later calls
 * with the same (level, ngrid, array) reuse the device copy.
 * ------------------------------------------------------------------------- */
/* ---------------------------------------------------------------------------
 * Gravity on the device-resident level (first part): the acceleration f(:,1:3)
 * sits next to the hydro state on the device (loaded from the host array on first use, rewritten by
 * ramses_amd_force_fine_f90), and the routines of amr_step's gravity branch that touch uold run there:
 *   ramses_amd_resident_synchro_f90        synchro_hydro_fine(ilevel,dteff,1)  hydro/synchro_hydro_fine.f90:5-136
 *   ramses_amd_resident_courant_grav_f90   courant_fine with cmpdt's gravity term  hydro/courant_fine.f90:77-85
 *   ramses_amd_resident_godunov_grav_f90   set_unew + godunov_fine with the predictor's source term
 *   ramses_amd_resident_set_uold_grav_f90  add_gravity_source_terms + set_uold  hydro/godunov_fine.f90:135-289
 *   ramses_amd_resident_sync_density_f90   uold(:,1) back to the host for rho_fine (pm/rho_fine.f90:666-820)
 * ------------------------------------------------------------------------- */
int ramses_amd_resident_synchro_f90(const ramses_amd_hydro_params *p, int ilevel, int ngrid,
                                    const int *igrid, const double *xg, int64_t ngridmax,
                                    int64_t ncoarse, int nx_loc, const double *uold, const double *f, double dteff);
int ramses_amd_resident_courant_grav_f90(const ramses_amd_hydro_params *p, int ilevel, int ngrid,
                                         const int *igrid, const double *xg, int64_t ngridmax,
                                         int64_t ncoarse, int nx_loc, const double *uold, const double *f,
                                         double dx, double dt_in, double *out4);
int ramses_amd_resident_godunov_grav_f90(const ramses_amd_hydro_params *p, int ilevel, int ngrid,
                                         const int *igrid, const double *xg, int64_t ngridmax,
                                         int64_t ncoarse, int nx_loc, const double *uold, const double *f,
                                         double dx, double dt);
int ramses_amd_resident_set_uold_grav_f90(const ramses_amd_hydro_params *p, int ilevel, double dt);
int ramses_amd_resident_sync_density_f90(double *uold);

/* force_fine(ilevel,icount) -- poisson/force_fine.f90:5-194 with gradient_phi :199-324 -- on the
 * reference's own arrays (host pointers): f(1:ncell,1:3) of the level's cells from phi, for a fully
 * refined periodic level of a single-rank run (gravity_type = 0).  The diagnostics of :158-190 are
 * reduced on the device: diag2 = {sum over leaf cells and directions of fact*f**2 (the level's
 * contribution to epot_tot), max |rho| (rho_max(ilevel))}; rho = rho(1:ncell), son_or_dummy = son(1:ncell)
 * read when has_son != 0 (a finer level exists: only cells with son == 0 count), fact =
 * -dx_loc**ndim/fourpi/2.  The maximum is exact; the sum is a fixed reduction tree (deterministic,
 * equal to the reference's serial loop up to rounding -- it feeds the energy-conservation print only). */
int ramses_amd_force_fine_f90(int ilevel, int ngrid, const int *igrid, const double *xg,
                              int64_t ngridmax, int64_t ncoarse, int nx_loc, const double *phi, double *f,
                              const double *rho, const int *son_or_dummy, int has_son, double fact, double *diag2);

/* The Poisson branch of amr_step on the device-resident level: no host array is
 * touched between rho_fine and force_fine; phi, f and rho reach the host on demand (backup_poisson).
 *   ramses_amd_resident_rho_fine_f90     rho_fine's hydro deposit (pm/rho_fine.f90: multipole_fine :666-820,
 *       cic_from_multipole :825-891, cic_cell :896-1142): every cell's mass max(rho,smallr)*vol is CIC-deposited
 *       at its centre of mass (m*x)/m, a target cell adds what it receives in the order of the reference's
 *       loop nest (batch of nvector octs, ind_son, CIC corner, oct in batch); multipole4 = multipole(1:4),
 *       strictly sequential sums in list order (rho_tot = multipole(1)/scale**ndim, :179).  Bit-identical.
 *   ramses_amd_resident_multigrid_f90    multigrid_fine on that deposit (poisson/multigrid_fine_commons.f90:25-296)
 *   ramses_amd_resident_force_fine_f90   force_fine into the resident acceleration + the diagnostics (see above)
 *   ramses_amd_resident_sync_poisson_f90 phi(1:ncell), f(1:ncell,1:3), rho(1:ncell) of the level back to the host */
int ramses_amd_resident_rho_fine_f90(const ramses_amd_hydro_params *p, int ilevel, int ngrid,
                                     const int *igrid, const double *xg, int64_t ngridmax,
                                     int64_t ncoarse, int nx_loc, const double *uold, double boxlen,
                                     int nvector, double *multipole4);
int ramses_amd_resident_multigrid_f90(int ilevel, double rho_tot, double fourpi, double epsilon, int *safe_mode,
                                      int *iters, double *err);
int ramses_amd_resident_force_fine_f90(int ilevel, double fact, double *diag2);
int ramses_amd_resident_sync_poisson_f90(double *phi, double *f, double *rho);

/* Page-lock a host array that the staged entry points (…_host/_f90) copy from and to.  The
 * reference allocates uold/unew (hydro/init_hydro.f90:30-32), phi/rho/f (poisson/init_poisson.f90:24-28)
 * and the tree (amr/init_amr.f90:52-55,227-233) once, with fixed ngridmax: their addresses are stable
 * for the run.  Never fatal: if the driver refuses, the copies take the pageable path.
 * Opt-in: a no-op unless RAMSES_AMD_PIN=1 (measured gain so far 5-10 % of the staged calls). */
int ramses_amd_host_register(void *p, int64_t bytes);

/* out4 = {dt_loc (min with dt_in), mass_loc, sum(E*vol), eint_loc} */
int ramses_amd_resident_courant_f90(const ramses_amd_hydro_params *p, int ilevel, int ngrid,
                                    const int *igrid, const double *xg, int64_t ngridmax,
                                    int64_t ncoarse, int nx_loc, const double *uold, double dx,
                                    double dt_in, double *out4);
int ramses_amd_resident_godunov_f90(const ramses_amd_hydro_params *p, int ilevel, int ngrid,
                                    const int *igrid, const double *xg, int64_t ngridmax,
                                    int64_t ncoarse, int nx_loc, const double *uold, double dx,
                                    double dt);
int ramses_amd_resident_set_uold_f90(int ilevel);
int ramses_amd_resident_sync_host_f90(double *uold);
int ramses_amd_resident_invalidate(void);

/* ---------------------------------------------------------------------------
 * MPI: one rank per GPU.  The virtual-boundary exchange on the device.
 *
 * Transport: neighbour send/recv over xGMI with RCCL.  librccl.so is loaded on first use; the
 * communicator is bootstrapped by the caller's own launcher: rank 0 calls ramses_amd_rccl_unique_id,
 * broadcasts the 128 bytes (the Fortran shim: MPI_BCAST; the Python mirror: torch.distributed), every
 * rank calls ramses_amd_rccl_init after selecting its device (ramses_amd_set_device_auto).
 * ramses_amd_rccl_exchange: ONE grouped ncclSend/ncclRecv, message i to/from rank peer[i], offsets and
 * counts in doubles into device buffers (replaces the MPI_ISEND/MPI_IRECV rounds of make_virtual_fine_dp,
 * amr/virtual_boundaries.f90:373-528, with all nvar fields fused); peer[i] may be the caller's own rank (the
 * send to self matches the receive from self of the same group; counts must agree).  ramses_amd_rccl_allreduce: in-place
 * reduction of n device doubles, op 0 sum / 1 min / 2 max (courant_fine's MPI_ALLREDUCE,
 * hydro/courant_fine.f90:133-140; multigrid norms; CG dot products).
 * ------------------------------------------------------------------------- */
#define RAMSES_AMD_RCCL_ID_BYTES 128
/* equal for two processes exactly when they drive the same GPU of the same host (RCCL refuses that) */
int ramses_amd_device_uid(int64_t *uid);
/* local half of the bring-up (dlopen + symbols, no communication): the launcher reduces the result over the
 * ranks BEFORE anyone enters the collective ramses_amd_rccl_init, so that one rank without librccl.so makes
 * every rank fall back to the host transport instead of leaving the others blocked in ncclCommInitRank */
int ramses_amd_rccl_probe(void);
int ramses_amd_rccl_unique_id(char *id128);
int ramses_amd_rccl_init(const char *id128, int nranks, int rank);
int ramses_amd_rccl_ready(void);
int ramses_amd_rccl_finalize(void);
int ramses_amd_rccl_exchange(int npeer, const int *peer, const double *d_send, const int64_t *send_off,
                             const int64_t *send_cnt, double *d_recv, const int64_t *recv_off,
                             const int64_t *recv_cnt, void *stream);
/* one pointer per message: send i to send_peer[i], receive i from recv_peer[i] (per peer in posting order) */
int ramses_amd_rccl_sendrecv(int nsend, const double *const *send_ptr, const int64_t *send_cnt, const int *send_peer,
                             int nrecv, double *const *recv_ptr, const int64_t *recv_cnt, const int *recv_peer, void *stream);
int ramses_amd_rccl_allreduce(double *d_buf, int n, int op, void *stream);
/* d_recv[r*count .. (r+1)*count) = d_send of rank r */
int ramses_amd_rccl_allgather(const double *d_send, int64_t count, double *d_recv, void *stream);

/* ---------------------------------------------------------------------------
 * multigrid_fine of a periodic, fully refined level cut into one brick per rank, the V-cycle driver WITH its halo
 * exchanges behind the C ABI (csrc/mg_dist.hip).  Replaces multigrid_fine + recursive_multigrid_coarse
 * (poisson/multigrid_fine_commons.f90:25-296,307-390) and their make_virtual_mg_dp / make_virtual_fine_dp rounds
 * (:1172-1290; amr/virtual_boundaries.f90:373-528) for a level whose rank domains are boxes: the 2^level cube on
 * pgrid[0] x pgrid[1] x pgrid[2] ranks (powers of two; bricks of (2^level / pgrid[d]) cells, every extent >= 64).
 * Brick b = x + pgrid[0]*(y + pgrid[1]*z) belongs to rank rank_of_brick[b] (NULL: the identity) -- the reference's
 * Hilbert order of the domains is the caller's to state.
 * transport NULL: RCCL inside the library (ramses_amd_rccl_init first).  Otherwise the caller's message layer on
 * pinned HOST buffers, blocking calls (several ranks sharing one GPU; the Fortran shim's MPI; CPU protocol tests):
 *   exchange      message i: send h_send[send_off[i] .. +send_cnt[i]) to rank peer[i], receive h_recv[recv_off[i] ..
 *                 +recv_cnt[i]) from it (doubles; one message per peer, the caller's own rank never appears)
 *   allgather     h_recv[r*count ..) = h_send of rank r
 *   allreduce_sum *value = sum over the ranks, the same on every rank
 * each returning 0 on success.
 * ramses_amd_mgdist_solve: d_rho = this rank's dense [nz][ny][nx] brick of the density; phi starts from zero;
 * convergence test, MAXITER and the safe-mode switch of multigrid_fine_commons.f90:261-282 (the flag persists in the
 * context like the reference's safe_mode(ilevel)).  _get_phi / _set_phi: dense bricks; _force: halo of phi +
 * gradient_phi into a dense [3][nz][ny][nx] array (poisson/force_fine.f90:199-324).
 * ------------------------------------------------------------------------- */
typedef struct ramses_amd_mg_transport {
  void *user;
  int (*exchange)(void *user, int npeer, const int *peer, const double *h_send, const int64_t *send_off,
                  const int64_t *send_cnt, double *h_recv, const int64_t *recv_off, const int64_t *recv_cnt);
  int (*allgather)(void *user, const double *h_send, int64_t count, double *h_recv);
  int (*allreduce_sum)(void *user, double *value);
} ramses_amd_mg_transport;
typedef struct ramses_amd_mgdist ramses_amd_mgdist;
int ramses_amd_mgdist_create(int level, const int *pgrid, int rank, const int *rank_of_brick,
                             const ramses_amd_mg_transport *transport, ramses_amd_mgdist **out);
int ramses_amd_mgdist_destroy(ramses_amd_mgdist *ctx);
/* any output may be NULL: brick extents and coordinates of this rank, number of distributed levels, first replicated
 * level (0: none), the safe-mode flag, halo exchanges so far */
int ramses_amd_mgdist_info(const ramses_amd_mgdist *ctx, int *dims, int *coords, int *n_distributed_levels,
                           int *first_replicated_level, int *safe_mode, int64_t *exchanges);
int ramses_amd_mgdist_set_safe_mode(ramses_amd_mgdist *ctx, int safe_mode);
/* The order in which the reference adds the squared residuals of this rank's cells (cmp_residual_norm2_fine,
 * poisson/multigrid_fine_fine.f90:254-287: octant by octant over active(ilevel)%igrid): order[k] = index of the k-th cell of that
 * loop in the rank's dense brick, a permutation of 0 .. N-1 (host array); n = 0 switches back.  With an order set, the two norms
 * of every iteration (poisson/multigrid_fine_commons.f90:205-211, 261-276) are strictly sequential sums in it -- the reference's
 * bits and therefore its convergence decision -- instead of the smoother's reduction tree.  ramses_amd_mgdist_multigrid_f90
 * sets it from the list it is called with. */
int ramses_amd_mgdist_set_order(ramses_amd_mgdist *ctx, const int *order, int64_t n);
int ramses_amd_mgdist_solve(ramses_amd_mgdist *ctx, const double *d_rho, double rho_tot, double fourpi, double epsilon,
                            int *iters_out, double *err_out, void *stream);
int ramses_amd_mgdist_get_phi(ramses_amd_mgdist *ctx, double *d_phi, void *stream);
int ramses_amd_mgdist_set_phi(ramses_amd_mgdist *ctx, const double *d_phi, void *stream);
int ramses_amd_mgdist_force(ramses_amd_mgdist *ctx, double *d_f, void *stream);
/* The Fortran shim's side (patch/multigrid_fine_commons.f90, several ranks, levelmin fully refined and periodic):
 * _oct_box: the box the rank's octs of the level fill -- lo[3] first cell, dims[3] cells -- from their centres xg
 * (host only; RAMSES_AMD_EUNSUPPORTED when they do not fill one: the caller keeps the multigrid of AMR levels);
 * _multigrid_f90: multigrid_fine(ilevel,icount) on the reference's own arrays: rho / phi are the cell vectors
 * (1:ncoarse+8*ngridmax), igrid = active(ilevel)%igrid; phi of the rank's own cells is written back (the caller
 * refreshes the virtual octs with make_virtual_fine_dp like the reference, multigrid_fine_commons.f90:284-287);
 * safe_mode in/out = the reference's safe_mode(ilevel). */
int ramses_amd_mgdist_oct_box(int ilevel, int ngrid, const int *igrid, const double *xg, int64_t ngridmax, int *lo, int *dims);
/* force_fine(ilevel,icount) (poisson/force_fine.f90:5-194) right after _multigrid_f90 of the same level: halo of phi +
 * gradient_phi on the brick the solve left on the device, f(:,1:3) of the rank's own cells into the host cell vectors
 * f (ncell,3); diag[0] = the local sum of fact*f**2 over the leaf cells in the reference's order (:162-172, batches of
 * nvector octs), diag[1] = max |rho| (:174-176) for the caller's two MPI_ALLREDUCEs (:182-184).  The caller refreshes
 * the virtual octs of f with make_virtual_fine_dp (:136-138). */
int ramses_amd_mgdist_force_f90(ramses_amd_mgdist *ctx, int ilevel, int ngrid, const int *igrid, const double *xg,
                                int64_t ngridmax, int64_t ncoarse, const int *lo, double *f, const double *rho, const int *son,
                                int nvector, double fact, double *diag);
/* The same for a run whose cell vectors are resident, on a level without finer octs: f goes from the brick into the resident
 * acceleration on the device (ramses_amd_amrres_take_f_device) and the potential-energy sum of
 * /root/reference/poisson/force_fine.f90:150-176 runs there in the reference's order; nothing of f crosses PCIe.  The caller
 * exchanges the virtual octs on the device (ramses_amd_amrres_halo_*, direction 7). */
int ramses_amd_mgdist_force_resident_f90(ramses_amd_mgdist *M, int ilevel, int ngrid, const int *igrid, int64_t ngridmax,
                                         int64_t ncoarse, const double *rho, int nvector, double fact, double *diag);
/* multigrid_fine of that level for a resident run with ONE level (levelmin = nlevelmax) and several ranks (round 6): the right-hand
 * side is gathered on the device from rho_fine's deposit (ramses_amd_amrres_rho_keep(1) keeps it there;
 * ramses_amd_amrres_rho_to_brick), the potential stays on the rank's brick for force_fine; ramses_amd_mgdist_fetch_phi_f90
 * writes it into the host vector for backup_poisson (/root/reference/poisson/output_poisson.f90).  No level array crosses PCIe
 * in a step.  ramses_amd_mgdist_traffic: bytes of rho (out2[0], host -> device) and phi (out2[1], device -> host) the Fortran
 * entries of the distributed solve moved since the start. */
int ramses_amd_mgdist_multigrid_resident_f90(ramses_amd_mgdist *M, int ilevel, int ngrid, const int *igrid, const double *xg,
                                             int64_t ngridmax, const int *lo, double rho_tot, double fourpi, double epsilon,
                                             int *safe_mode, int *iters, double *err);
int ramses_amd_mgdist_fetch_phi_f90(ramses_amd_mgdist *M, int ngrid, const int *igrid, int64_t ngridmax, int64_t ncoarse, double *phi);
int ramses_amd_mgdist_traffic(int64_t *out2);
/* ramses_amd_mgdist_force_resident_f90 with the deposit on the device (rho = NULL there): the Fortran binding */
int ramses_amd_mgdist_force_resident_dev_f90(ramses_amd_mgdist *M, int ilevel, int ngrid, const int *igrid, int64_t ngridmax,
                                             int64_t ncoarse, int nvector, double fact, double *diag);
/* host only (no device): the deep-halo plan ramses_amd_mgdist_create builds for one rank and a level whose bricks have
 * dims[3] cells inside ng ghost layers -- 26 send / receive regions (org x,y,z + ext x,y,z in allocated coordinates) with
 * their positions in the message buffers, and the messages (one per peer; the caller's own rank where the box wraps onto
 * itself).  For the CPU tests of the multi-rank protocol. */
int ramses_amd_mgdist_plan(const int *pgrid, int rank, const int *rank_of_brick, const int *dims, int ng,
                           int *send_boxes, int64_t *send_offs, int *recv_boxes, int64_t *recv_offs,
                           int *npeer, int *seg_peer, int64_t *seg_send_off, int64_t *seg_send_cnt,
                           int64_t *seg_recv_off, int64_t *seg_recv_cnt, int64_t *total);
int ramses_amd_mgdist_multigrid_f90(ramses_amd_mgdist *ctx, int ilevel, int ngrid, const int *igrid, const double *xg,
                                    int64_t ngridmax, int64_t ncoarse, const int *lo, const double *rho, double *phi,
                                    double rho_tot, double fourpi, double epsilon, int *safe_mode, int *iters, double *err);

/* ---------------------------------------------------------------------------
 * Device image of the reference's communicators (type communicator, amr/amr_commons.f90:108-119;
 * emission(:,l) / reception(:,l) built by build_comm, amr/virtual_boundaries.f90:1286-1648) for one
 * rank's share of a fully refined periodic level: a box of octs, stored as a brick with a one-oct
 * ghost layer.  Lists are passed concatenated over icpu = 1..ncpu: em_ngrid[ncpu] / rc_ngrid[ncpu] =
 * emission(icpu,l)%ngrid / reception(icpu,l)%ngrid, em_igrid / rc_igrid = the %igrid arrays one after
 * the other (1-based oct slots).
 * ramses_amd_halo_plan: host-side only (no device needed): the box {olo[3], odim[3] in octs, self_axes
 *   (bit d: direction d spans the whole box: periodic self-copy instead of a peer), number of ghost
 *   positions} and the brick offsets of every list entry.  Fails when the rank's octs do not fill a box
 *   or the reception lists do not cover the one-oct shell around it.
 * ------------------------------------------------------------------------- */
int ramses_amd_halo_plan(int ilevel, int ngrid, const int *igrid, const double *xg, int64_t ngridmax, int ncpu,
                         const int *em_ngrid, const int *em_igrid, const int *rc_ngrid, const int *rc_igrid,
                         int *out_box, int64_t *act_org, int64_t *em_org, int *rc_src, int64_t *rc_org, int64_t rc_cap);

/* Device-resident level under MPI (hydro): the routines of amr_step that touch uold/unew of the level,
 *   ramses_amd_mpires_setup         load (active + ghost octs) from the host arrays uold/unew(1:ncell,1:nvar)
 *   ramses_amd_mpires_courant       courant_fine: out4 = local {dt, mass, sum E vol, eint}; the caller reduces over ranks
 *   ramses_amd_mpires_godunov       set_unew + godunov_fine           hydro/godunov_fine.f90:5-130
 *   ramses_amd_mpires_reverse_unew  make_virtual_reverse_dp(unew(1,1:nvar),l)   amr/virtual_boundaries.f90:693-983
 *   ramses_amd_mpires_set_uold      set_uold (buffer swap)            hydro/godunov_fine.f90:135-232
 *   ramses_amd_mpires_halo_forward  make_virtual_fine_dp(uold(1,1:nvar),l) over RCCL   :373-528
 *   ramses_amd_mpires_halo_stage_out / _stage_in   the same with the caller's MPI as transport (pinned host buffers)
 *   ramses_amd_mpires_which         +ivar / -ivar when xx is uold(1,ivar) / unew(1,ivar) of the level, else 0
 *   ramses_amd_mpires_sync_host     host array refreshed on demand (backup_hydro)
 * ------------------------------------------------------------------------- */
int ramses_amd_mpires_setup(const ramses_amd_hydro_params *p, int ilevel, int ngrid, const int *igrid, const double *xg,
                            int64_t ngridmax, int64_t ncoarse, int nx_loc, const double *uold, const double *unew,
                            int ncpu, int myid, const int *em_ngrid, const int *em_igrid, const int *rc_ngrid,
                            const int *rc_igrid);
int ramses_amd_mpires_active(void);
int ramses_amd_mpires_which(const double *xx);
int ramses_amd_mpires_courant(const ramses_amd_hydro_params *p, double dx, double dt_in, double *out4);
int ramses_amd_mpires_godunov(const ramses_amd_hydro_params *p, double dx, double dt);
int ramses_amd_mpires_reverse_unew(void);
int ramses_amd_mpires_set_uold(void);
int ramses_amd_mpires_halo_forward(void);
int ramses_amd_mpires_halo_stage_out(double **h_send, const int64_t **send_off, double **h_recv, const int64_t **recv_off);
int ramses_amd_mpires_halo_stage_out_f90(int64_t *h_send_addr, int64_t *h_recv_addr, int64_t *send_off, int64_t *recv_off, int ncpu);
int ramses_amd_mpires_halo_stage_in(void);
int ramses_amd_mpires_sync_host(double *uold);
int ramses_amd_mpires_invalidate(void);

/* ---------------------------------------------------------------------------
 * Residency for AMR runs (hydro, with self-gravity on one rank).  The reference's own cell vectors
 * uold/unew(1:ncell,1:nvar) and tree arrays stay on the device between the routines of amr_step:
 *   ramses_amd_amrres_load         uold + son(1:ncell), nbor(1:ngridmax,1:6), father(1:ngridmax)
 *   ramses_amd_amrres_tree         the tree again (after refine_fine)
 *   ramses_amd_amrres_sync_level / _load_level   uold of one level's cells to / from the host array
 *                                  (igrid = active(l)%igrid): before refine_fine reads levels, after it rebuilt them
 *   ramses_amd_amrres_sync_all     the whole uold back (backup_hydro)
 *   ramses_amd_amrres_set_unew     set_unew      hydro/godunov_fine.f90:40-130
 *   ramses_amd_amrres_godunov      godunov_fine  hydro/godunov_fine.f90:5-35,486-911 (the tree-walking sweep)
 *   ramses_amd_amrres_set_uold     set_uold      hydro/godunov_fine.f90:135-232 (incl. the passive-scalar fix :176-190)
 *   ramses_amd_amrres_upload_fine  upload_fine / upl  hydro/interpol_hydro.f90:5-263
 *   ramses_amd_amrres_courant      courant_fine  hydro/courant_fine.f90:1-159 (leaf cells; out4 as ramses_amd_resident_courant_f90)
 *   ramses_amd_amrres_hydro_flag   hydro_flag's gradient criteria (hydro/hydro_flag.f90:84-140, hydro_refine
 *                                  hydro/godunov_utils.f90:125-263): cells[0..*ncells) = the cells (1-based, ascending) that ask
 *                                  for refinement, compacted on the device; capacity of cells: 8*ngrid
 * ------------------------------------------------------------------------- */
int ramses_amd_amrres_active(void);
int ramses_amd_amrres_load(int nvar, int64_t ngridmax, int64_t ncoarse, const double *uold, const int *son, const int *nbor,
                           const int *father);
int ramses_amd_amrres_tree(const int *son, const int *nbor, const int *father);
/* The coarsest level the last ramses_amd_amrres_tree (or _load) numbered again on the device: its octs and those of every finer
 * level have new device indices and NO data -- the caller sends them again (ramses_amd_amrres_load_level, _load_f) before
 * anything reads them, as refine_fine's shim does (patch/ramses_amd_iface.f90: ramses_amd_amr_reload_from); the coarser
 * levels keep indices, data and sweep plans.  nlev + 1: nothing changed; 0: the host's numbering is in force. */
int ramses_amd_amrres_first_changed(void);
int ramses_amd_amrres_invalidate(void);
int ramses_amd_amrres_sync_level(int ngrid, const int *igrid, double *uold);
int ramses_amd_amrres_load_level(int ngrid, const int *igrid, const double *uold);
int ramses_amd_amrres_sync_all(double *uold);
int ramses_amd_amrres_set_unew(int ngrid, const int *igrid);
int ramses_amd_amrres_set_uold(const ramses_amd_hydro_params *p, int ngrid, const int *igrid);
int ramses_amd_amrres_upload_fine(const ramses_amd_hydro_params *p, int ngrid, const int *igrid, int interpol_var);
int ramses_amd_amrres_courant(const ramses_amd_hydro_params *p, int ngrid, const int *igrid, double dx, double dt_in, double *out4);
/* with self-gravity: the acceleration f(1:ncell,1:3) stays the host array's (force_fine writes it there); the device copy
 * is refreshed per level (load_f: after force_fine(ilevel), after a regrid) and feeds synchro_hydro_fine
 * (hydro/synchro_hydro_fine.f90:5-136), cmpdt's gravity term, the sweep's gravity predictor and add_gravity_source_terms
 * (hydro/godunov_fine.f90:237-289, set_uold_grav).  sync_density: uold(:,1) of one level back to the host for rho_fine's
 * multipole_fine (pm/rho_fine.f90:666-770), which reads nothing else of the hydro state. */
int ramses_amd_amrres_load_f(int ngrid, const int *igrid, const double *f);
/* The acceleration of a resident run with several ranks without a visit to the host (round 6): force_fine's kernel leaves f of
 * the rank's own cells in a device buffer d_fpack[3][8][ngrid] (octs in the order of igrid), ramses_amd_amrres_take_f_device
 * files it into the resident f, the virtual octs follow with ramses_amd_amrres_halo_* (direction 7: make_virtual_fine_dp on
 * f(:,1:3), /root/reference/poisson/force_fine.f90:137-139).  ramses_amd_amrres_sync_f brings f of the listed octs back into the
 * host array for the two host readers left (backup_poisson, load_balance); ramses_amd_amrres_f_traffic reports the bytes of f
 * that crossed PCIe since the start (out2[0] host -> device, out2[1] device -> host). */
int ramses_amd_amrres_take_f_device(int ngrid, const int *igrid, const double *d_fpack);
int ramses_amd_amrres_sync_f(int ngrid, const int *igrid, double *f);
int ramses_amd_amrres_f_traffic(int64_t *out2);
/* Diagnostic (RAMSES_AMD_F_CHECK=1 in patch/force_fine.f90): the largest |device f - host f| over the listed octs and the
 * number of cells that differ. */
int ramses_amd_amrres_compare_f(int ngrid, const int *igrid, const double *f, double *maxdiff, int64_t *ndiff);
/* rho_fine's deposit of a resident run with several ranks (ramses_amd_amrres_rho_mpi_*): _rho_keep(1) leaves it on the device
 * (ramses_amd_amrres_rho_mpi_finish skips the copy into the host vector), _sync_rho brings the listed octs back (backup_poisson),
 * _rho_to_brick gathers the rank's own cells into a dense brick through an order list on the device
 * (d_brick[d_order[ind * ngrid + g]] = rho of cell ind of oct igrid[g]), _rho_absmax returns max |rho| over them
 * (/root/reference/poisson/force_fine.f90:177-181), _rho_traffic the bytes of rho that went back to the host since the start. */
int ramses_amd_amrres_rho_keep(int on);
int ramses_amd_amrres_sync_rho(int ngrid, const int *igrid, double *rho);
int ramses_amd_amrres_rho_to_brick(int ngrid, const int *igrid, const int *d_order, double *d_brick);
int ramses_amd_amrres_rho_absmax(int ngrid, const int *igrid, double *out);
int64_t ramses_amd_amrres_rho_traffic(void);
int ramses_amd_amrres_has_gravity(void);
int ramses_amd_amrres_sync_density(int ngrid, const int *igrid, double *uold);
int ramses_amd_amrres_synchro(const ramses_amd_hydro_params *p, int ngrid, const int *igrid, double dteff);
int ramses_amd_amrres_set_uold_grav(const ramses_amd_hydro_params *p, int ngrid, const int *igrid, double dt);
/* pressure_fix (hydro/godunov_fine.f90:66-83 set_unew's divu = 0 / enew = e_int, :294-481 add_pdv_source_terms, :203-227 the
 * energy switch of set_uold): divu and enew are device vectors (scratch of one hydro step; the sweep adds its divergence
 * and internal-energy fluxes to them, level and coarser level); dx_loc = 0.5**ilevel * boxlen/nx_loc */
int ramses_amd_amrres_enable_pfix(void);
int ramses_amd_amrres_set_unew_pfix(const ramses_amd_hydro_params *p, int ngrid, const int *igrid);
int ramses_amd_amrres_set_uold_pfix(const ramses_amd_hydro_params *p, int ngrid, const int *igrid, double dt, double dx_loc,
                                    double beta_fix, double hexp);
/* rho_fine(ilevel,icount) on the resident density of an AMR run (single rank, periodic nx=ny=nz=1 box, no particles):
 *   multipole_fine(l) + cic_from_multipole(l) for l = nlevelmax .. ilevel (pm/rho_fine.f90:45-60,666-1142): multipoles of leaf
 *   and split cells, the CIC deposit at every cell's centre of mass added in the reference's order, the four sequential sums
 *   multipole(1:4) over the cells of levelmin.  ramses_amd_amrres_xg: the oct centres xg(1:ngridmax,1:3), once per regrid.
 *   first[0..nlevelmax-ilevel+1] / igrid_all: active(l)%igrid of the levels one after the other; rho (host, ncell) receives the
 *   visited levels' cells; multipole4 is written when levelmin is visited. */
int ramses_amd_amrres_xg(const double *xg);
int ramses_amd_amrres_rho_fine(const ramses_amd_hydro_params *p, int ilevel, int nlevelmax, int levelmin, int nvector,
                               const int *first, const int *igrid_all, double boxlen_over_nx, double *rho, double *multipole4);
/* The same with several ranks (one per GPU), one level at a time from nlevelmax down to ilevel; igrid_all = the rank's n_own own
 * octs of the level followed by its reception octs (n_all in all).  cic_cell loops over the own octs only and deposits into own
 * and virtual cells alike (pm/rho_fine.f90:896-1142); the reference's exchanges stay with the caller, on the device vectors
 * through ramses_amd_amrres_halo_* (dir 6: make_virtual_fine_dp on the four multipoles, :814-817; dir 4 / 5:
 * make_virtual_reverse_dp(rho) / make_virtual_fine_dp(rho), :58-59):
 *   _multipole -> [dir 6] -> _deposit -> [dir 4] -> [dir 5] -> _finish (rho of the level's own + reception cells into the host
 *   vector; multipole4 = the rank's four sequential sums when ilevel == levelmin, before the caller's MPI_ALLREDUCE, :176-183) */
int ramses_amd_amrres_rho_mpi_multipole(const ramses_amd_hydro_params *p, int ilevel, int n_own, int n_all, const int *igrid_all,
                                        double boxlen_over_nx);
int ramses_amd_amrres_rho_mpi_deposit(int ilevel, int nvector, double boxlen_over_nx);
int ramses_amd_amrres_rho_mpi_finish(int ilevel, int levelmin, int nvector, const int *igrid_all, double *rho, double *multipole4);
int ramses_amd_amrres_hydro_flag(const ramses_amd_hydro_params *p, int ngrid, const int *igrid, double err_grad_d, double err_grad_p,
                                 double err_grad_u, double floor_d, double floor_p, double floor_u, int *cells, int *ncells);
int ramses_amd_amrres_godunov(const ramses_amd_hydro_params *p, int ilevel, int ngrid, const int *igrid, double dx, double dt,
                              int nvector, int interpol_var, int interpol_type);
/* The device numbers the octs of a resident run itself (csrc/amr_layout.hpp): the cell vectors keep the reference's formula
 * icell = ncoarse + (ind-1)*ngridmax + igrid (hydro/godunov_fine.f90:600-601), but levels of 64^3 .. 4096^3 cells are stored in
 * TILES of 32 x 4 x 4 octs -- level-contiguous SoA blocks, 256-byte runs along x -- and every list / index that crosses this
 * interface is translated.  godunov_fine of such a level runs the DENSE z-marching sweep in place inside
 * ramses_amd_amrres_godunov: fluxes through the faces of refined cells reset, the update starting from unew
 * (hydro/godunov_fine.f90:661-666,720-790), missing neighbour octs interpolated once per sweep into free tile slots (:563-626),
 * the fluxes owed to the coarser level filed and replayed in the reference's order (:798-908) -- strict arithmetic,
 * bit-identical to the tree-walking sweep and to the reference.  NVAR = 5, muscl, slope types 0/1/2/7/8, every Riemann solver
 * but 'exact', no difmag / pressure_fix, one coarse cell (no physical boundaries); anything else keeps the tree-walking sweep.
 * Switches: RAMSES_AMD_DEVICE_ORDER=0 (the host's numbering on the device, tree-walking sweep everywhere), RAMSES_AMD_TILES=0
 * (Z-order numbering, no tiles), RAMSES_AMD_COVERED_DENSE=0 / RAMSES_AMD_TILE_DENSE=0 (fully refined / partial levels keep the
 * tree-walking sweep), RAMSES_AMD_DEVICE_OCTS=f (the device's index space = f x ngridmax: tiles that are not full cost
 * indices; a level that does not fit is numbered along the Z-order curve and keeps the tree-walking sweep).
 * _covered_sweeps: sweeps of fully refined levels through the dense kernel so far; _tile_sweeps: of any level in tiles;
 * _tree_sweeps: through the tree-walking kernel; _tiled_levels: levels stored in tiles (0: host numbering in force). */
int64_t ramses_amd_amrres_covered_sweeps(void);
int64_t ramses_amd_amrres_tile_sweeps(void);
int64_t ramses_amd_amrres_tree_sweeps(void);
int64_t ramses_amd_amrres_relayouts(void);   /* regrids that moved the kept levels to new device indices (their tiles were in the way) */
int ramses_amd_amrres_tiled_levels(void);
/* make_boundary_hydro(ilevel) on the resident cell vectors (hydro/hydro_boundary.f90:5-269; callers amr/amr_step.f90:70,293,514):
 * the boundary octs of every physical boundary region of a level take the mirrored (boundary_type 1-6) or the copied
 * (11-16, with the no_inflow clamp :176-189) state of their reference cells, region after region in the reference's order,
 * octs of a region in list order, in chunks of nvector octs, cell index by cell index inside a chunk (a region two octs deep
 * reads what the same pass has already written, exactly as the reference's in-place loop :119-262 does).  btype = boundary_type(1:nregion), ngrid[r] = boundary(r,ilevel)%ngrid, igrid = the regions' oct lists
 * one after the other.  Imposed boundaries (21-26): the caller evaluates the reference's boundana (:215-241) for the cells of the
 * region and passes the states, [nvar][8][ngrid[r]] per imposed region in order (host array; NULL when there is none). */
int ramses_amd_amrres_boundary_hydro(int nregion, const int *btype, const int *ngrid, const int *igrid, int no_inflow, double smallr,
                                     int nvector, const double *imposed);
/* Several MPI ranks (one per GPU): the virtual-boundary exchanges of amr_step on the resident cell vectors.
 *   make_virtual_fine_dp(uold(1,ivar),ilevel)     amr/virtual_boundaries.f90:373-528, callers amr/amr_step.f90:61,287,505
 *   make_virtual_reverse_dp(unew(1,ivar),ilevel)  amr/virtual_boundaries.f90:693-983, caller amr/amr_step.f90:397
 *   (and enew / divu with pressure_fix, amr/amr_step.f90:417-418); set_unew's zeroing of the virtual octs
 *   hydro/godunov_fine.f90:92-122.
 * comm_set hands over the communicators build_comm left for a level (emission(icpu,l)%igrid / reception(icpu,l)%igrid,
 * concatenated in icpu order; epoch = the shim's count of build_comm calls, comm_epoch returns the one on the device, -1 if
 * none).  dir 0: forward on uold; 1: reverse on unew; 2 / 3: reverse on enew / divu.  All nvar variables travel in one
 * message per peer (reference: one round per variable); the reverse exchange accumulates peer by peer in icpu order, as the
 * reference does.  Transport: ramses_amd_amrres_halo_rccl (one grouped ncclSend/ncclRecv, myid 1-based) or, with several
 * ranks on one GPU, the caller's own MPI between halo_stage_out (pack; pinned host buffers, addresses as integers, offsets
 * [ncpu+1] in doubles) and halo_stage_in (unpack / accumulate).
 * ramses_amd_which_column: which column (1..ncol, 0 = none) of a host array base(1:ncell,1:ncol) an anonymous xx is. */
int ramses_amd_which_column(const double *xx, const double *base, int64_t ncell, int ncol);
int ramses_amd_amrres_comm_epoch(int ilevel);
int ramses_amd_amrres_comm_set(int ilevel, int epoch, int ncpu, const int *em_n, const int *em_ig, const int *rc_n, const int *rc_ig);
int ramses_amd_amrres_zero_unew_virtual(int ilevel);
int ramses_amd_amrres_halo_rccl(int ilevel, int dir, int myid);
int ramses_amd_amrres_halo_stage_out(int ilevel, int dir, int ncpu, int64_t *h_send_addr, int64_t *h_recv_addr, int64_t *send_off,
                                     int64_t *recv_off);
int ramses_amd_amrres_halo_stage_in(int ilevel, int dir);

/* ---------------------------------------------------------------------------
 * SOLVER=mhd: the constrained-transport MHD Godunov sweep.
 * Replaces: mhd/godunov_fine.f90 godfine1 :538-1459 on a fully refined periodic level (no coarse-fine boundaries),
 *           mhd/umuscl.f90 mag_unsplit :31-238 (ctoprim :2029-2186, uslope :2187-2844, trace3d :750-1307, cmpflxm
 *           :1308-1448, cmp_mag_flx :1453-2028), mhd/godunov_utils.f90 upwind / lax_friedrich / hll / hlld / athena_roe / hydro_acoustic / eigen_cons :313-1517, fused with
 *           set_unew (mhd/godunov_fine.f90:40-110).  NDIM = 3, NVAR = 8, NENER = 0, scheme = 'muscl'.
 * riemann: 0 llf, 1 roe, 2 hll, 3 hlld, 4 upwind (= llf in cmpflxm), 5 hydro (hydro_acoustic);  riemann2d: 0 llf, 1 roe, 2 upwind,
 * 3 hll, 4 hlla, 5 hlld -- every solver of the reference (its iriemann /
 * iriemann2d codes, hydro/read_hydro_params.f90:184-220);  slope_type: 0, 1, 2, 3, 7, 8; slope_mag_type: 0, 1, 2, 7, 8 (slope_mag_type = -1
 * means slope_type, :528-530).  Anything else returns RAMSES_AMD_EUNSUPPORTED.
 * d_uold / d_unew: [11][nz][ny][nx] device doubles -- rho, rho u, rho v, rho w, E, the three left-face fields (uold(:,6:8)),
 * the three right-face fields (uold(:,nvar+1:nvar+3)) -- periodic, distinct buffers; the right-face field of a cell must equal
 * the left-face field of its neighbour bit for bit (it does on any level the scheme has advanced; RAMSES_AMD_EINVAL if not).
 * d_work: ramses_amd_mhd_workspace_bytes(nx,ny,nz) bytes of device scratch.  Bit-identical to the reference.
 * ramses_amd_mhd_godunov_brick_fast: the same sweep in the fast arithmetic (divisions as v_rcp_f64 + Newton steps, contracted
 * multiply-adds; <= 1e-12 relative L-infinity of the reference, div B at rounding all the same); RAMSES_AMD_MHD_FAST=1 in the
 * environment routes ramses_amd_mhd_godunov_brick -- and the drop-in's staged and resident sweeps -- to it. */
typedef struct ramses_amd_mhd_params {
  double gamma, smallr, smallc, slope_theta;
  int32_t slope_type, slope_mag_type, riemann, riemann2d;
} ramses_amd_mhd_params;
int64_t ramses_amd_mhd_workspace_bytes(int nx, int ny, int nz);
int ramses_amd_mhd_godunov_brick(const ramses_amd_mhd_params *p, int nx, int ny, int nz, const double *d_uold, double *d_unew,
                                 double dx, double dt, void *d_work, int64_t work_bytes, void *stream);
int ramses_amd_mhd_godunov_brick_fast(const ramses_amd_mhd_params *p, int nx, int ny, int nz, const double *d_uold, double *d_unew,
                                      double dx, double dt, void *d_work, int64_t work_bytes, void *stream);
/* godunov_fine(ilevel) of a SOLVER=mhd run on the reference's own arrays uold / unew (1:ncell,1:nvar+3), staged through the
 * device: the level's cells go up, unew of the level's cells comes back (ramses_amd/patch_mhd/godunov_fine.f90). */
int ramses_amd_mhd_godunov_fine_f90(const ramses_amd_mhd_params *p, int ilevel, int ngrid, const int *igrid, const double *xg,
                                    int64_t ngridmax, int64_t ncoarse, int nx_loc, const double *uold, double *unew, double dx,
                                    double dt);
/* The level of a SOLVER=mhd run RESIDENT on the device between the routines of amr_step (one rank, one fully refined periodic
 * level, levelmin = nlevelmax, no gravity / pressure_fix / magnetic diffusion): uold(1:ncell,1:nvar+3) goes up once, then
 *   ramses_amd_mhd_resident_courant_f90    courant_fine  mhd/courant_fine.f90:1-160 (cmpdt mhd/godunov_utils.f90:5-115 per cell:
 *                                          out5 = {min(dt_in, CFL step), mass, total, internal, magnetic energy of the level})
 *   ramses_amd_mhd_resident_godunov_f90    set_unew + godunov_fine  mhd/godunov_fine.f90:5-109  (a second brick = uold advanced)
 *   ramses_amd_mhd_resident_set_uold_f90   set_uold  mhd/godunov_fine.f90:185-281 (the bricks swap roles; the host array is stale)
 *   ramses_amd_mhd_resident_sync_host_f90  the level's cells back into uold (backup_hydro; no-op while the host is current)
 * ramses_amd/patch_mhd/{courant_fine,godunov_fine,output_hydro}.f90 bind them; RAMSES_AMD_MHD_RESIDENT=0 keeps the staged sweep. */
int ramses_amd_mhd_resident_active(void);
int ramses_amd_mhd_resident_courant_f90(const ramses_amd_mhd_params *p, int ilevel, int ngrid, const int *igrid, const double *xg,
                                        int64_t ngridmax, int64_t ncoarse, int nx_loc, const double *uold, double dx, double dt_in,
                                        double courant_factor, double *out5);
int ramses_amd_mhd_resident_godunov_f90(const ramses_amd_mhd_params *p, int ilevel, int ngrid, const int *igrid, const double *xg,
                                        int64_t ngridmax, int64_t ncoarse, int nx_loc, const double *uold, double dx, double dt);
int ramses_amd_mhd_resident_set_uold_f90(int ilevel);
int ramses_amd_mhd_resident_sync_host_f90(double *uold);
int ramses_amd_mhd_resident_invalidate(void);

/* SOLVER=mhd on the levels of an AMR tree: godfine1 as the reference writes it (mhd/godunov_fine.f90:538-1459) -- the 6^3
 * stencil of every oct through get3cubefather (amr/nbors_utils.f90:5-194), missing neighbour octs interpolated from the
 * coarser level by interpol_hydro (mhd/interpol_hydro.f90:612-793: limited slopes for the Euler variables, the
 * divergence-free face interpolation interpol_mag :990-1047 with interpol_faces :1052-1241, copy_from_refined_faces
 * :1246-1349, cmp_central_faces :1354-1473, compute_2d_tvd :1478-1527), mag_unsplit (mhd/umuscl.f90:31-238) on the stencil,
 * fluxes and EMFs reset next to refined cells (:760-903), the update of unew with constrained transport (:909-1022) and the
 * flux / EMF corrections of the leaf cells of level ilevel-1 (:1024-1457) added in the reference's order (batches of nvector
 * octs of the list; Euler system per direction and side, then the twelve edges).  Bit-identical to the reference.
 *   ramses_amd_mhd_godfine_amr_device    device arrays: d_uold / d_unew [11][ncell] (uold(1:ncell,1:nvar+3)), d_f [3][ncell]
 *                                        or NULL (gravity: ctoprim's half kick), the tree son [ncell], nbor [6][ngridmax],
 *                                        father [ngridmax], d_igrid the octs of the call; coarse != 0: ilevel > levelmin.
 *   ramses_amd_mhd_godunov_fine_amr_f90  the same on the reference's host arrays (staged: ramses_amd/patch_mhd/godunov_fine.f90)
 *                                        (f is read when use_f != 0: poisson)
 *   ramses_amd_mhd_amr_sweeps / _octs    calls / octs swept so far
 *   ramses_amd_mhd_note_reference_sweep  the drop-in reports a level it hands to the reference's host godunov_fine instead; the
 *                                        library prints both counts, per level, in one line at exit
 * Levels >= 3 of a periodic box; interpol_var 0..1, interpol_type 0..3, interpol_mag_type 0..3 (-1 resolved by the caller). */
int ramses_amd_mhd_godfine_amr_device(const ramses_amd_mhd_params *p, int ilevel, int ngrid, const int *d_igrid, const int *d_son,
                                      const int *d_nbor, const int *d_father, int64_t ngridmax, int64_t ncoarse, const double *d_uold,
                                      double *d_unew, const double *d_f, double dx, double dt, int nvector, int interpol_var,
                                      int interpol_type, int interpol_mag_type, int coarse, void *stream);
int ramses_amd_mhd_godunov_fine_amr_f90(const ramses_amd_mhd_params *p, int ilevel, int levelmin, int ngrid, const int *igrid,
                                        const int *son, const int *nbor, const int *father, int64_t ngridmax, int64_t ncoarse,
                                        const double *uold, double *unew, const double *f, int use_f, double dx, double dt,
                                        int nvector, int interpol_var, int interpol_type, int interpol_mag_type);
int64_t ramses_amd_mhd_amr_sweeps(void);
int64_t ramses_amd_mhd_amr_octs(void);
int ramses_amd_mhd_note_reference_sweep(int ilevel);

#ifdef __cplusplus
}
#endif
#endif /* RAMSES_AMD_H */
