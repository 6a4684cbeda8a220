// capi.hip -- the C ABI of libramses_amd.so (declared in include/ramses_amd.h).
// Plain pointers and PODs only; validates arguments, derives constants,
// dispatches to the gfx950 kernels.  No CPU fallback: anything that cannot run
// on the device returns an error code.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/ramses_amd.h"
#include "amr_args.hpp"
#include "amr_sweep_args.hpp"
#include "cg_amr_args.hpp"
#include "mg_amr_args.hpp"
#include "mg_args.hpp"
#include "misc_args.hpp"
#include "pack_args.hpp"
#include "rho_args.hpp"
#include "sweep_args.hpp"

using namespace ramses_amd;

static thread_local char g_err[512] = "";
static int g_tile_rows = 0;  // 0 = per-variant default
static int g_zchunk = 128;
static int g_mg_fused = 1;

static int fail(int code, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}
static int hipfail(hipError_t e, const char *what) {
  return fail(RAMSES_AMD_EHIP, "%s: %s", what, hipGetErrorString(e));
}

static HydroConst make_const(const ramses_amd_hydro_params *p) {
  HydroConst P;
  P.gamma = p->gamma;
  P.smallr = p->smallr;
  P.smallc = p->smallc;
  P.smallc2 = p->smallc * p->smallc;
  P.smallp = P.smallc2 / p->gamma;                       // smallc**2/gamma
  P.smalle = P.smallc2 / p->gamma / (p->gamma - 1.0);    // smallc**2/gamma/(gamma-one)
  P.entho = 1.0 / (p->gamma - 1.0);
  P.gm1 = p->gamma - 1.0;
  P.gamma6 = (p->gamma + 1.0) / (2.0 * p->gamma);
  P.smallpp = p->smallr * P.smallp;
  P.oneovergamma = 1.0 / p->gamma;
  P.slope_theta = p->slope_theta;
  P.niter_riemann = p->niter_riemann;
  return P;
}

static bool is_pow2(double x) {
  if (!(x > 0.0) || !std::isfinite(x)) return false;
  int e;
  return std::frexp(x, &e) == 0.5;
}

static int check_brick(const ramses_amd_brick *b) {
  if (!b) return fail(RAMSES_AMD_EINVAL, "brick is NULL");
  // a direction of extent 1 (an embedded 1-D/2-D problem) needs ghost layers: the in-kernel wrap assumes n >= 2
  const int nmin = b->ng >= 2 ? 1 : 2;
  if (b->nx < 2 || b->ny < nmin || b->nz < nmin) return fail(RAMSES_AMD_EINVAL, "brick must have >=2 cells per direction, or 1 in y/z with ghost layers (got %d %d %d, ng=%d)", b->nx, b->ny, b->nz, b->ng);
  if (b->ng != 0 && b->ng < 2) return fail(RAMSES_AMD_EINVAL, "ghost width must be 0 or >=2 (got %d)", b->ng);
  if (b->pitch_y < b->nx + 2 * b->ng) return fail(RAMSES_AMD_EINVAL, "pitch_y too small");
  if (b->pitch_z < b->pitch_y * (b->ny + 2 * b->ng)) return fail(RAMSES_AMD_EINVAL, "pitch_z too small");
  if (b->pitch_var < b->pitch_z * (b->nz + 2 * b->ng)) return fail(RAMSES_AMD_EINVAL, "pitch_var too small");
  return 0;
}

extern "C" {

const char *ramses_amd_last_error(void) { return g_err; }
// for the other translation units of the library (capi_mpi.hip)
int ramses_amd_set_error(int code, const char *msg) { return fail(code, "%s", msg ? msg : ""); }

int ramses_amd_abi_check(size_t sizeof_hydro_params, size_t sizeof_brick) {
  if (sizeof_hydro_params != sizeof(ramses_amd_hydro_params) || sizeof_brick != sizeof(ramses_amd_brick))
    return fail(RAMSES_AMD_EINVAL, "ABI mismatch: hydro_params %zu (library %zu), brick %zu (library %zu)",
                sizeof_hydro_params, sizeof(ramses_amd_hydro_params), sizeof_brick, sizeof(ramses_amd_brick));
  return 0;
}

void ramses_amd_brick_dense(ramses_amd_brick *b, int nx, int ny, int nz, int ng) {
  b->nx = nx; b->ny = ny; b->nz = nz; b->ng = ng;
  b->pitch_y = (int64_t)nx + 2 * ng;
  b->pitch_z = b->pitch_y * ((int64_t)ny + 2 * ng);
  b->pitch_var = b->pitch_z * ((int64_t)nz + 2 * ng);
}

// One process per GPU under MPI: pick the device from the launcher's local rank
// (falls back to the world rank); all ranks share device 0 on a 1-GPU box.
int ramses_amd_set_device_auto(int world_rank) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) return hipfail(e, "hipGetDeviceCount");
  if (n <= 0) return fail(RAMSES_AMD_ENODEVICE, "no HIP device visible");
  int local = world_rank;
  const char *vars[] = {"OMPI_COMM_WORLD_LOCAL_RANK", "MPI_LOCALRANKID", "PMI_LOCAL_RANK", "SLURM_LOCALID", "LOCAL_RANK"};
  for (const char *v : vars) {
    const char *x = getenv(v);
    if (x && *x) { local = atoi(x); break; }
  }
  if (local < 0) local = 0;
  e = hipSetDevice(local % n);
  if (e != hipSuccess) return hipfail(e, "hipSetDevice");
  return 0;
}

int ramses_amd_device_info(char *name, size_t name_len, int *n_cu, size_t *hbm_bytes) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n == 0) return fail(RAMSES_AMD_ENODEVICE, "no HIP device: %s", hipGetErrorString(e));
  hipDeviceProp_t prop;
  int dev = 0;
  hipGetDevice(&dev);
  e = hipGetDeviceProperties(&prop, dev);
  if (e != hipSuccess) return hipfail(e, "hipGetDeviceProperties");
  if (name && name_len) { std::strncpy(name, prop.gcnArchName, name_len - 1); name[name_len - 1] = 0; }
  if (n_cu) *n_cu = prop.multiProcessorCount;
  if (hbm_bytes) *hbm_bytes = prop.totalGlobalMem;
  return n;
}

int ramses_amd_godunov_tune(int tile_rows, int zchunk) {
  if (tile_rows != 0 && tile_rows != 8 && tile_rows != 12) return fail(RAMSES_AMD_EINVAL, "tile_rows must be 8 or 12 (got %d)", tile_rows);
  if (zchunk < 0) return fail(RAMSES_AMD_EINVAL, "zchunk must be >=0");
  g_tile_rows = tile_rows;
  g_zchunk = zchunk ? zchunk : 128;
  return 0;
}

// NDIM < 3: the 1-D/2-D problem is embedded in the brick (ny and/or nz = 1, ghost
// layers of the unused directions filled periodically = copies of the cell): the
// transverse slopes and flux differences vanish identically, so the 3-D kernels
// return the 1-D/2-D result of the reference bit for bit; cmpdt takes dble(ndim).
static int check_ndim(const ramses_amd_hydro_params *p, const ramses_amd_brick *b) {
  if (p->ndim < 1 || p->ndim > 3) return fail(RAMSES_AMD_EINVAL, "NDIM must be 1, 2 or 3 (got %d)", p->ndim);
  if (p->ndim < 3 && (b->nz != 1 || b->ng < 2)) return fail(RAMSES_AMD_EINVAL, "NDIM=%d needs a brick with nz=1 and ghost layers (got nz=%d, ng=%d)", p->ndim, b->nz, b->ng);
  if (p->ndim < 2 && b->ny != 1) return fail(RAMSES_AMD_EINVAL, "NDIM=1 needs a brick with ny=nz=1 (got ny=%d)", b->ny);
  return 0;
}

static int godunov_brick_region(const ramses_amd_hydro_params *p, const ramses_amd_brick *b,
                                const double *d_uold, const double *d_grav, double *d_unew,
                                double dx, double dt, int region_first, int region_last, void *stream) {
  if (!p) return fail(RAMSES_AMD_EINVAL, "params is NULL");
  if (int rc = check_brick(b)) return rc;
  if (!d_uold || !d_unew) return fail(RAMSES_AMD_EINVAL, "uold/unew device pointers are NULL");
  if (d_uold == d_unew) return fail(RAMSES_AMD_EINVAL, "uold and unew must be distinct buffers");
  if (int rc = check_ndim(p, b)) return rc;
  if (p->nvar < 5 || p->nvar > 7) return fail(RAMSES_AMD_EUNSUPPORTED, "device sweep implements NVAR=5..7 (up to two passive scalars; got %d)", p->nvar);
  if (p->nvar != 5 && p->scheme != RAMSES_AMD_SCHEME_MUSCL) return fail(RAMSES_AMD_EUNSUPPORTED, "passive scalars with scheme='plmde' are not on the device yet");
  if (p->scheme != RAMSES_AMD_SCHEME_MUSCL && p->scheme != RAMSES_AMD_SCHEME_PLMDE) return fail(RAMSES_AMD_EINVAL, "unknown scheme %d", p->scheme);
  if (p->difmag > 0.0) return fail(RAMSES_AMD_EUNSUPPORTED, "difmag>0 is not implemented on the device yet");
  // slope types: 0,1,2,3,7,8 in every build of the reference; 4,5,6 (superbee, ultrabee, central) exist in its NDIM=1
  // branch only (hydro/umuscl.f90:1030-1090), where type 3 means type 2 (MIN(slope_type,2), :1014-1023)
  int slope_type = p->slope_type;
  if (p->ndim == 1 && slope_type == 3) slope_type = 2;
  const bool st1d = slope_type == 4 || slope_type == 5 || slope_type == 6;
  if (st1d && p->ndim != 1)
    return fail(RAMSES_AMD_EUNSUPPORTED, "slope_type=%d exists in NDIM=1 runs of the reference only (0,1,2,3,7,8 in 2-D/3-D)", slope_type);
  if (st1d && (p->nvar != 5 || p->scheme != RAMSES_AMD_SCHEME_MUSCL || d_grav))
    return fail(RAMSES_AMD_EUNSUPPORTED, "slope_type=%d: hydro variables only, scheme='muscl', no gravity", slope_type);
  if (!(slope_type == 0 || slope_type == 1 || slope_type == 2 || slope_type == 3 || slope_type == 7 || slope_type == 8 || st1d))
    return fail(RAMSES_AMD_EUNSUPPORTED, "slope_type=%d is not a slope type of the reference", slope_type);
  if (p->riemann < 0 || p->riemann > 4) return fail(RAMSES_AMD_EINVAL, "unknown Riemann solver %d", p->riemann);
  if (!(dx > 0.0) || !(dt >= 0.0)) return fail(RAMSES_AMD_EINVAL, "dx must be >0 and dt >=0");

  SweepArgs A;
  A.uold = d_uold; A.unew = d_unew; A.grav = d_grav;
  A.nx = b->nx; A.ny = b->ny; A.nz = b->nz; A.ng = b->ng;
  A.pitch_y = b->pitch_y; A.pitch_z = b->pitch_z; A.pitch_var = b->pitch_var;
  A.zchunk = g_zchunk < b->nz ? g_zchunk : b->nz;
  A.dt = dt; A.dx = dx; A.rdx = 1.0 / dx;
  A.P = make_const(p);
  A.pow2 = is_pow2(dx) ? 1 : 0;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  for (int region = region_first; region <= region_last; region++) {
    A.region = region;
    hipError_t e = p->fast_math
                       ? fastmode::launch_godunov_sweep(A, slope_type, p->riemann, g_tile_rows, p->scheme, p->nvar, d_grav != nullptr, s)
                       : strictmode::launch_godunov_sweep(A, slope_type, p->riemann, g_tile_rows, p->scheme, p->nvar, d_grav != nullptr, s);
    if (e != hipSuccess) return hipfail(e, "godunov sweep launch");
  }
  return 0;
}

int ramses_amd_godunov_brick(const ramses_amd_hydro_params *p, const ramses_amd_brick *b,
                             const double *d_uold, const double *d_grav, double *d_unew,
                             double dx, double dt, void *stream) {
  return godunov_brick_region(p, b, d_uold, d_grav, d_unew, dx, dt, SWEEP_ALL, SWEEP_ALL, stream);
}

int ramses_amd_godunov_brick_shell(const ramses_amd_hydro_params *p, const ramses_amd_brick *b,
                                   const double *d_uold, const double *d_grav, double *d_unew,
                                   double dx, double dt, void *stream) {
  return godunov_brick_region(p, b, d_uold, d_grav, d_unew, dx, dt, SWEEP_SHELL, SWEEP_SHELL, stream);
}

int ramses_amd_godunov_brick_interior(const ramses_amd_hydro_params *p, const ramses_amd_brick *b,
                                      const double *d_uold, const double *d_grav, double *d_unew,
                                      double dx, double dt, void *stream) {
  return godunov_brick_region(p, b, d_uold, d_grav, d_unew, dx, dt, SWEEP_INTERIOR, SWEEP_INTERIOR, stream);
}

int ramses_amd_courant_init(const ramses_amd_hydro_params *p, double dx, double *d_out, void *stream) {
  if (!p || !d_out) return fail(RAMSES_AMD_EINVAL, "NULL argument");
  // cmpdt's starting value: courant_factor*dx/smallc (godunov_utils.f90:113)
  const double dt0 = p->courant_factor * dx / p->smallc;
  hipError_t e = launch_courant_init(d_out, dt0, reinterpret_cast<hipStream_t>(stream));
  if (e != hipSuccess) return hipfail(e, "courant init launch");
  return 0;
}

int ramses_amd_courant_brick(const ramses_amd_hydro_params *p, const ramses_amd_brick *b,
                             const double *d_uold, const double *d_grav, double dx,
                             double *d_out, void *stream) {
  if (!p || !d_uold || !d_out) return fail(RAMSES_AMD_EINVAL, "NULL argument");
  if (int rc = check_brick(b)) return rc;
  if (int rc = check_ndim(p, b)) return rc;
  if (p->nvar < 5) return fail(RAMSES_AMD_EUNSUPPORTED, "device courant needs the 5 hydro variables (passive scalars do not enter cmpdt)");
  CourantArgs A;
  A.uold = d_uold; A.grav = d_grav; A.out = d_out;
  A.nx = b->nx; A.ny = b->ny; A.nz = b->nz; A.ng = b->ng;
  A.pitch_y = b->pitch_y; A.pitch_z = b->pitch_z; A.pitch_var = b->pitch_var;
  A.dx = dx; A.vol = dx * dx * dx;
  A.courant_factor = p->courant_factor;
  A.dt_init = p->courant_factor * dx / p->smallc;
  A.ndimf = (double)p->ndim;
  A.P = make_const(p);
  hipError_t e = launch_courant(A, d_grav != nullptr, reinterpret_cast<hipStream_t>(stream));
  if (e != hipSuccess) return hipfail(e, "courant launch");
  return 0;
}

// make_boundary_hydro (hydro/hydro_boundary.f90:5-269) for one face of a ghost-layer brick
int ramses_amd_make_boundary_hydro(const ramses_amd_hydro_params *p, const ramses_amd_brick *b, double *d_uold,
                                   int face, int bound_type, const double *imposed, int no_inflow, void *stream) {
  if (!p || !d_uold) return fail(RAMSES_AMD_EINVAL, "NULL argument");
  if (int rc = check_brick(b)) return rc;
  if (b->ng < 1) return fail(RAMSES_AMD_EINVAL, "physical boundaries need a brick with ghost layers");
  if (face < 0 || face > 5) return fail(RAMSES_AMD_EINVAL, "face must be 0..5");
  if (p->nvar < 5 || p->nvar > 8) return fail(RAMSES_AMD_EUNSUPPORTED, "NVAR=%d", p->nvar);
  BoundaryArgs A;
  A.u = d_uold; A.nx = b->nx; A.ny = b->ny; A.nz = b->nz; A.ng = b->ng; A.nvar = p->nvar;
  A.pitch_y = b->pitch_y; A.pitch_z = b->pitch_z; A.pitch_var = b->pitch_var;
  A.face = face; A.no_inflow = no_inflow ? 1 : 0; A.smallr = p->smallr;
  // bound_type: the reference's codes 1..6 reflexive, 11..16 outflow, 21..26 imposed (direction = face)
  const int kind = bound_type / 10;
  if (kind < 0 || kind > 2 || bound_type % 10 != face + 1) return fail(RAMSES_AMD_EINVAL, "bound_type %d does not belong to face %d", bound_type, face);
  A.type = kind + 1;
  for (int v = 0; v < 8; v++) A.value[v] = 0.0;
  if (A.type == 3) {
    if (!imposed) return fail(RAMSES_AMD_EINVAL, "imposed boundary needs the conserved state");
    for (int v = 0; v < p->nvar; v++) A.value[v] = imposed[v];
  }
  hipError_t e = launch_boundary(A, reinterpret_cast<hipStream_t>(stream));
  if (e != hipSuccess) return hipfail(e, "boundary launch");
  return 0;
}

// slab geometry for face f of brick b: origin (in allocated coordinates) and
// extents of the interior slab (pack) / ghost slab (unpack)
static void slab_box(const ramses_amd_brick *b, int face, bool ghost, int org[3], int ext[3]) {
  const int n[3] = {b->nx, b->ny, b->nz};
  const int ng = b->ng;
  const int axis = face / 2, hi = face & 1;
  for (int d = 0; d < 3; d++) {
    if (d < axis) { org[d] = 0; ext[d] = n[d] + 2 * ng; }       // already exchanged: full extent
    else if (d > axis) { org[d] = ng; ext[d] = n[d]; }           // not yet exchanged: interior
    else {
      ext[d] = ng;
      if (!ghost) org[d] = hi ? n[d] : ng;                       // interior cells next to the face
      else org[d] = hi ? n[d] + ng : 0;                          // ghost cells beyond the face
    }
  }
}

int64_t ramses_amd_halo_slab_size(const ramses_amd_brick *b, int nvar, int face) {
  if (check_brick(b)) return RAMSES_AMD_EINVAL;
  if (b->ng < 2 || face < 0 || face > 5 || nvar < 1) return fail(RAMSES_AMD_EINVAL, "halo slab needs ng>=2, 0<=face<=5");
  int org[3], ext[3];
  slab_box(b, face, false, org, ext);
  return (int64_t)ext[0] * ext[1] * ext[2] * nvar;
}

static int slab_copy(const ramses_amd_brick *b, const double *src, double *dst, int nvar, int face,
                     bool pack, void *stream) {
  if (int rc = check_brick(b)) return rc;
  if (b->ng < 2 || face < 0 || face > 5 || nvar < 1) return fail(RAMSES_AMD_EINVAL, "halo slab needs ng>=2, 0<=face<=5");
  if (!src || !dst) return fail(RAMSES_AMD_EINVAL, "NULL buffer");
  int org[3], ext[3];
  slab_box(b, face, !pack, org, ext);
  BoxCopyArgs A;
  A.src = src; A.dst = dst;
  A.ex = ext[0]; A.ey = ext[1]; A.ez = ext[2]; A.nvar = nvar;
  const long boff = org[0] + (long)org[1] * b->pitch_y + (long)org[2] * b->pitch_z;
  const long cpy = ext[0], cpz = (long)ext[0] * ext[1], cpv = (long)ext[0] * ext[1] * ext[2];
  if (pack) {
    A.s_off = boff; A.s_py = b->pitch_y; A.s_pz = b->pitch_z; A.s_pv = b->pitch_var;
    A.d_off = 0; A.d_py = cpy; A.d_pz = cpz; A.d_pv = cpv;
  } else {
    A.s_off = 0; A.s_py = cpy; A.s_pz = cpz; A.s_pv = cpv;
    A.d_off = boff; A.d_py = b->pitch_y; A.d_pz = b->pitch_z; A.d_pv = b->pitch_var;
  }
  hipError_t e = launch_box_copy(A, reinterpret_cast<hipStream_t>(stream));
  if (e != hipSuccess) return hipfail(e, "halo slab copy launch");
  return 0;
}

int ramses_amd_halo_pack(const ramses_amd_brick *b, const double *d_u, int nvar, int face,
                         double *d_buf, void *stream) {
  return slab_copy(b, d_u, d_buf, nvar, face, true, stream);
}
int ramses_amd_halo_unpack(const ramses_amd_brick *b, double *d_u, int nvar, int face,
                           const double *d_buf, void *stream) {
  return slab_copy(b, d_buf, d_u, nvar, face, false, stream);
}

// One-shot halo (all faces, edges and corners in one launch).  boxes: nbox x 6 ints
// (org x,y,z in allocated coordinates, ext x,y,z); offsets: nbox positions (in doubles) in d_buf.
int ramses_amd_halo_multi(const ramses_amd_brick *b, double *d_u, int nvar, int nbox, const int *boxes,
                          const int64_t *offsets, double *d_buf, int pack, void *stream) {
  if (int rc = check_brick(b)) return rc;
  if (!d_u || !d_buf || !boxes || !offsets) return fail(RAMSES_AMD_EINVAL, "NULL argument");
  if (nbox < 0 || nbox > 26) return fail(RAMSES_AMD_EINVAL, "at most 26 regions");
  if (nbox == 0) return 0;
  MultiBoxArgs A;
  A.brick = d_u; A.buf = d_buf; A.nbox = nbox; A.nvar = nvar; A.pack = pack ? 1 : 0;
  A.pitch_y = b->pitch_y; A.pitch_z = b->pitch_z; A.pitch_var = b->pitch_var;
  const int full[3] = {b->nx + 2 * b->ng, b->ny + 2 * b->ng, b->nz + 2 * b->ng};
  A.rows_before[0] = 0;
  for (int r = 0; r < nbox; r++) {
    for (int d = 0; d < 3; d++) {
      A.org[r][d] = boxes[6 * r + d];
      A.ext[r][d] = boxes[6 * r + 3 + d];
      if (A.org[r][d] < 0 || A.ext[r][d] < 1 || A.org[r][d] + A.ext[r][d] > full[d])
        return fail(RAMSES_AMD_EINVAL, "region %d leaves the brick", r);
    }
    A.off[r] = offsets[r];
    A.rows_before[r + 1] = A.rows_before[r] + (long)A.ext[r][1] * A.ext[r][2] * nvar;
  }
  hipError_t e = launch_multi_box(A, reinterpret_cast<hipStream_t>(stream));
  if (e != hipSuccess) return hipfail(e, "halo multi-box launch");
  return 0;
}

int ramses_amd_fill_ghosts_periodic(const ramses_amd_brick *b, double *d_u, int nvar, int axes,
                                    void *stream) {
  if (int rc = check_brick(b)) return rc;
  if (b->ng < 2) return fail(RAMSES_AMD_EINVAL, "periodic ghost fill needs ng>=2");
  if (!d_u) return fail(RAMSES_AMD_EINVAL, "NULL buffer");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  for (int axis = 0; axis < 3; axis++) {
    if (!(axes & (1 << axis))) continue;
    for (int hi = 0; hi < 2; hi++) {
      // ghost slab beyond face (axis,hi) <- interior slab next to the opposite face
      int gorg[3], gext[3], sorg[3], sext[3];
      slab_box(b, 2 * axis + hi, true, gorg, gext);
      slab_box(b, 2 * axis + (1 - hi), false, sorg, sext);
      BoxCopyArgs A;
      A.src = d_u; A.dst = d_u;
      A.ex = gext[0]; A.ey = gext[1]; A.ez = gext[2]; A.nvar = nvar;
      A.s_off = sorg[0] + (long)sorg[1] * b->pitch_y + (long)sorg[2] * b->pitch_z;
      A.d_off = gorg[0] + (long)gorg[1] * b->pitch_y + (long)gorg[2] * b->pitch_z;
      A.s_py = A.d_py = b->pitch_y; A.s_pz = A.d_pz = b->pitch_z; A.s_pv = A.d_pv = b->pitch_var;
      hipError_t e = launch_box_copy(A, s);
      if (e != hipSuccess) return hipfail(e, "periodic ghost fill launch");
    }
  }
  return 0;
}

// ---------------------------------------------------------------------------
// multigrid Poisson solver on a fully refined periodic level
// ---------------------------------------------------------------------------
static size_t mg_level_cells(int l) { return (size_t)1 << (3 * l); }
// workspace: for l = 1..level-1: u1, u2, u3 (8^l doubles each); then the
// residual partial sums and two norm scalars
// which: 0 = u1 (correction), 1 = u2 (rhs), 2 = u3 (residual), 3 = ping-pong copy of u1
static size_t mg_hier_offset(int level, int l, int which) {
  size_t off = mg_level_cells(level);           // fine-level ping-pong copy of phi comes first
  for (int m = 1; m < l; m++) off += 4 * mg_level_cells(m);
  return off + (size_t)which * mg_level_cells(l);
}
static size_t mg_hier_size(int level) {
  size_t off = mg_level_cells(level);
  for (int m = 1; m < level; m++) off += 4 * mg_level_cells(m);
  return off;
}
// levels with n >= MG_FUSED_MIN_N use the fused time-skewed smoother: a launch of it costs ~120 us whatever the level
// (its planes are marched one barrier at a time), which only pays from 256^3 up; below, the per-colour kernels on
// cache-resident levels are faster (profiles/r02_vcycle_levels.txt).  RAMSES_AMD_MG_FUSED_MIN=<n> overrides (A/B, >= 64).
static int mg_fused_min_n() {
  static int v = -1;
  if (v < 0) {
    v = 256;
    const char *e = getenv("RAMSES_AMD_MG_FUSED_MIN");
    if (e && atoi(e) >= 64) v = atoi(e);
  }
  return v;
}
#define MG_FUSED_MIN_N mg_fused_min_n()

int64_t ramses_amd_mg_workspace_doubles(int level) {
  if (level < 1 || level > 11) return fail(RAMSES_AMD_EINVAL, "multigrid level must be in [1,11] (got %d)", level);
  return (int64_t)(mg_hier_size(level) + MG_MAX_PARTIALS + 8);
}

#define MGCHK(call, what) do { hipError_t e_ = (call); if (e_ != hipSuccess) return hipfail(e_, what); } while (0)

int ramses_amd_mg_gauss_seidel(double *d_phi, const double *d_rhs, int n, double dx2, int redstep, void *stream) {
  if (!d_phi || !d_rhs || n < 2 || (n & (n - 1))) return fail(RAMSES_AMD_EINVAL, "bad argument (n must be a power of two)");
  MGCHK(mg_launch_gs(d_phi, d_rhs, n, dx2, redstep ? 0 : 1, reinterpret_cast<hipStream_t>(stream)), "mg gs launch");
  return 0;
}
int ramses_amd_mg_residual(const double *d_phi, const double *d_rhs, double *d_res, int n, double dx,
                           double *d_work, double *d_norm2, void *stream) {
  if (!d_phi || !d_rhs || !d_res || n < 2 || (n & (n - 1))) return fail(RAMSES_AMD_EINVAL, "bad argument (n must be a power of two)");
  if (d_norm2 && !d_work) return fail(RAMSES_AMD_EINVAL, "norm needs a workspace of %d doubles", MG_MAX_PARTIALS);
  MGCHK(mg_launch_residual(d_phi, d_rhs, d_res, n, dx, d_work, d_norm2, reinterpret_cast<hipStream_t>(stream)), "mg residual launch");
  return 0;
}
int ramses_amd_mg_restrict(const double *d_res_f, double *d_rhs_c, double *d_u1_c, int nf, void *stream) {
  if (!d_res_f || !d_rhs_c || !d_u1_c || nf < 2 || (nf & (nf - 1))) return fail(RAMSES_AMD_EINVAL, "bad argument (nf must be a power of two)");
  MGCHK(mg_launch_restrict(d_res_f, d_rhs_c, d_u1_c, nf, reinterpret_cast<hipStream_t>(stream)), "mg restrict launch");
  return 0;
}
int ramses_amd_mg_interp_correct(double *d_phi_f, const double *d_corr_c, int nf, void *stream) {
  if (!d_phi_f || !d_corr_c || nf < 2 || (nf & (nf - 1))) return fail(RAMSES_AMD_EINVAL, "bad argument (nf must be a power of two)");
  MGCHK(mg_launch_interp(d_phi_f, d_corr_c, nf, reinterpret_cast<hipStream_t>(stream)), "mg interp launch");
  return 0;
}

// fused: 0 one launch per colour pass, 1 the fused smoother with 4 colour passes per launch (24-row
// tiles, one workgroup per CU), 12 / 16: two launches of 2 colour passes on 12- / 16-row tiles
// (three / two workgroups per CU).  Results do not depend on it.
static int g_mg_split_rows = 32;   // default: 2+2 colour passes on 32-row tiles (measured fastest at 512^3)
int ramses_amd_mg_tune(int fused) {
  g_mg_fused = fused ? 1 : 0;
  if (fused == 1) fused = 32;                       // the default
  g_mg_split_rows = (fused == 12 || fused == 16 || fused == 24 || fused == 32) ? fused : 0;   // 4: one 4-pass launch
  mg_set_smooth_rows(g_mg_split_rows ? g_mg_split_rows : 24);
  return 0;
}
// 4 colour passes (2 red-black sweeps) from *cur, optionally with the residual (+norm) of the
// result; on return *cur points at the result and *other at the scratch copy
// rhs_c / u1_c (optional, with *restricted): the level below; if the smoother of this configuration can restrict its own
// residual it does (the residual is then NOT stored in res) and *restricted is set -- otherwise the caller restricts res
static hipError_t mg_smooth4(double **cur, double **other, const double *rhs, double *res, double *partial,
                             double *norm, int n, double dx, hipStream_t s, double *rhs_c = nullptr, double *u1_c = nullptr,
                             bool *restricted = nullptr, const double *corr_c = nullptr) {
  // corr_c: the correction of the level below, to be prolongated and added to *cur first (interpolate_and_correct_fine) --
  // inside the first smoother launch where that configuration can, in a pass of its own otherwise
  hipError_t e;
  if (restricted) *restricted = false;
  static int fuse_prolong = -1;          // RAMSES_AMD_MG_FUSE_PROLONG=0: the prolongation in a pass of its own (A/B; same bits)
  if (fuse_prolong < 0) { const char *ev = getenv("RAMSES_AMD_MG_FUSE_PROLONG"); fuse_prolong = !(ev && ev[0] == '0'); }
  const bool prol_fused = corr_c && g_mg_split_rows && fuse_prolong && mg_smooth_can_restrict(n, 2);
  if (corr_c && !prol_fused && (e = mg_launch_interp(*cur, corr_c, n, s)) != hipSuccess) return e;
  if (g_mg_split_rows) {
    if ((e = mg_launch_smooth_fused(*cur, *other, rhs, nullptr, nullptr, nullptr, n, dx, 2, s, 0, nullptr, nullptr,
                                    prol_fused ? corr_c : nullptr)) != hipSuccess) return e;
    static int fuse_restrict = -1;       // RAMSES_AMD_MG_FUSE_RESTRICT=0: the restriction in a pass of its own (A/B; same bits)
    if (fuse_restrict < 0) { const char *ev = getenv("RAMSES_AMD_MG_FUSE_RESTRICT"); fuse_restrict = !(ev && ev[0] == '0'); }
    if (rhs_c && u1_c && restricted && fuse_restrict && mg_smooth_can_restrict(n, 2)) {
      *restricted = true;
      return mg_launch_smooth_fused(*other, *cur, rhs, nullptr, partial, norm, n, dx, 2, s, 0, rhs_c, u1_c);
    }
    return mg_launch_smooth_fused(*other, *cur, rhs, res, partial, norm, n, dx, 2, s);
  }
  if ((e = mg_launch_smooth_fused(*cur, *other, rhs, res, partial, norm, n, dx, 4, s)) != hipSuccess) return e;
  double *t = *cur; *cur = *other; *other = t;
  return hipSuccess;
}

int ramses_amd_mg_smooth_fused(const double *d_phi_in, double *d_phi_out, const double *d_rhs, double *d_res,
                               double *d_work, double *d_norm2, int n, double dx, int npass, void *stream) {
  if (!d_phi_in || !d_phi_out || !d_rhs || d_phi_in == d_phi_out || n < 2 || (n & 1)) return fail(RAMSES_AMD_EINVAL, "bad argument");
  if (npass != 2 && npass != 4) return fail(RAMSES_AMD_EINVAL, "npass must be 2 or 4");
  if (n < 64) return fail(RAMSES_AMD_EINVAL, "the fused smoother needs n >= 64 (got %d); use the per-colour kernels", n);
  if ((d_res || d_norm2) && !d_work) return fail(RAMSES_AMD_EINVAL, "residual/norm need a workspace of %d doubles", MG_MAX_PARTIALS);
  MGCHK(mg_launch_smooth_fused(d_phi_in, d_phi_out, d_rhs, d_res, d_work, d_norm2, n, dx, npass, reinterpret_cast<hipStream_t>(stream)), "mg fused smoother launch");
  return 0;
}

int ramses_amd_gradient_phi_brick(int level, const double *d_phi, double *d_f, void *stream) {
  if (level < 1 || level > 11 || !d_phi || !d_f) return fail(RAMSES_AMD_EINVAL, "bad argument");
  const int n = 1 << level;
  const double dx = std::ldexp(1.0, -level);
  const double a = 0.50 * 4.0 / 3.0 / dx;   // force_fine.f90:233-234
  const double b = 0.25 * 1.0 / 3.0 / dx;
  MGCHK(mg_launch_gradient(d_phi, d_f, n, a, b, reinterpret_cast<hipStream_t>(stream)), "gradient_phi launch");
  return 0;
}

// recursive_multigrid_coarse (multigrid_fine_commons.f90:307-390), levelmin_mg = 1
static hipError_t mg_coarse_cycle(double *w, int level, int l, int safe, hipStream_t s) {
  const int ngs_coarse = 2, ncycles_coarse_safe = 1;
  const int n = 1 << l;
  const double dx = std::ldexp(1.0, -l), dx2 = dx * dx;
  double *u1 = w + mg_hier_offset(level, l, 0), *u2 = w + mg_hier_offset(level, l, 1), *u3 = w + mg_hier_offset(level, l, 2);
  hipError_t e;
  // the coarsest levels: the whole rest of the V-cycle (down to level 1 and back up) in one launch of one workgroup
  // (RAMSES_AMD_MG_TAIL=<top level of the tail>, 0: one launch per colour pass everywhere; same bits)
  static int use_tail = -1;
  if (use_tail < 0) {
    const char *env = getenv("RAMSES_AMD_MG_TAIL");
    use_tail = env ? atoi(env) : 4;
    if (use_tail > MG_TAIL_LTOP) use_tail = MG_TAIL_LTOP;
  }
  if (l <= use_tail && l >= 2) {
    MgTailArgs T;
    T.w = w; T.ltop = l;
    for (int k = 1; k <= l; k++) {
      for (int a = 0; a < 3; a++) T.off[k][a] = (long)mg_hier_offset(level, k, a);
      const double dxk = std::ldexp(1.0, -k);
      T.dx2[k] = dxk * dxk;
      T.oneoverdx2[k] = 1.0 / (dxk * dxk);
    }
    return mg_launch_coarse_tail(T, s);
  }
  if (l <= 1) {
    for (int i = 0; i < 2 * ngs_coarse; i++) {
      if ((e = mg_launch_gs(u1, u2, n, dx2, 0, s)) != hipSuccess) return e;
      if ((e = mg_launch_gs(u1, u2, n, dx2, 1, s)) != hipSuccess) return e;
    }
    return hipSuccess;
  }
  const int ncycle = safe ? ncycles_coarse_safe : 1;
  double *partial = w + mg_hier_size(level);
  for (int cyc = 0; cyc < ncycle; cyc++) {
    if (n >= MG_FUSED_MIN_N) {
      // pre-smoothing + residual, correction, post-smoothing; the result ends in u1
      double *cur = u1, *oth = w + mg_hier_offset(level, l, 3);
      bool restricted = false;
      if ((e = mg_smooth4(&cur, &oth, u2, u3, partial, nullptr, n, dx, s, w + mg_hier_offset(level, l - 1, 1), w + mg_hier_offset(level, l - 1, 0),
                          &restricted)) != hipSuccess) return e;
      if (!restricted && (e = mg_launch_restrict(u3, w + mg_hier_offset(level, l - 1, 1), w + mg_hier_offset(level, l - 1, 0), n, s)) != hipSuccess) return e;
      if ((e = mg_coarse_cycle(w, level, l - 1, safe, s)) != hipSuccess) return e;
      if ((e = mg_smooth4(&cur, &oth, u2, nullptr, nullptr, nullptr, n, dx, s, nullptr, nullptr, nullptr, w + mg_hier_offset(level, l - 1, 0))) != hipSuccess) return e;
      continue;
    }
    for (int i = 0; i < ngs_coarse; i++) {
      if ((e = mg_launch_gs(u1, u2, n, dx2, 0, s)) != hipSuccess) return e;
      if ((e = mg_launch_gs(u1, u2, n, dx2, 1, s)) != hipSuccess) return e;
    }
    if ((e = mg_launch_residual(u1, u2, u3, n, dx, nullptr, nullptr, s)) != hipSuccess) return e;
    if ((e = mg_launch_restrict(u3, w + mg_hier_offset(level, l - 1, 1), w + mg_hier_offset(level, l - 1, 0), n, s)) != hipSuccess) return e;
    if ((e = mg_coarse_cycle(w, level, l - 1, safe, s)) != hipSuccess) return e;
    if ((e = mg_launch_interp(u1, w + mg_hier_offset(level, l - 1, 0), n, s)) != hipSuccess) return e;
    for (int i = 0; i < ngs_coarse; i++) {
      if ((e = mg_launch_gs(u1, u2, n, dx2, 0, s)) != hipSuccess) return e;
      if ((e = mg_launch_gs(u1, u2, n, dx2, 1, s)) != hipSuccess) return e;
    }
  }
  return hipSuccess;
}

// ---------------------------------------------------------------------------
// Distributed multigrid (one rank's brick with ghost layers per level; the
// V-cycle driver with its halo exchanges is ramses_amd/poisson_parallel.py).
// ---------------------------------------------------------------------------
static bool brick_extent_ok(int n) { return n >= 2 && !(n & (n - 1)); }
int ramses_amd_mg_smooth_fused_ghost(const double *d_phi_in, double *d_phi_out, const double *d_rhs,
                                     double *d_res, double *d_work, double *d_norm2, int nx, int ny, int nz, int ng,
                                     double dx, int npass, void *stream) {
  if (!d_phi_in || !d_phi_out || !d_rhs || d_phi_in == d_phi_out) return fail(RAMSES_AMD_EINVAL, "bad argument");
  if (!brick_extent_ok(nx) || !brick_extent_ok(ny) || !brick_extent_ok(nz)) return fail(RAMSES_AMD_EINVAL, "brick extents must be powers of two (got %d x %d x %d)", nx, ny, nz);
  if (npass != 2 && npass != 4) return fail(RAMSES_AMD_EINVAL, "npass must be 2 or 4");
  if (nx < 64 || ny < 64 || nz < 64) return fail(RAMSES_AMD_EINVAL, "the fused smoother needs every extent >= 64 (got %d x %d x %d)", nx, ny, nz);
  const int H = (d_res || d_norm2) ? npass + 1 : npass;
  if (ng < H) return fail(RAMSES_AMD_EINVAL, "ghost width %d does not cover the %d-cell dependency cone of %d colour passes", ng, H, npass);
  if ((d_res || d_norm2) && !d_work) return fail(RAMSES_AMD_EINVAL, "residual/norm need a workspace of %d doubles", MG_MAX_PARTIALS);
  MGCHK(mg_launch_smooth_fused(d_phi_in, d_phi_out, d_rhs, d_res, d_work, d_norm2, nx, dx, npass, reinterpret_cast<hipStream_t>(stream), ng,
                               nullptr, nullptr, nullptr, ny, nz), "mg fused smoother launch");
  return 0;
}
// f2 = fourpi*(rho - rho_tot) over N doubles (make_fine_bc_rhs on an unmasked level)
int ramses_amd_mg_rhs(const double *d_rho, double *d_f2, int64_t N, double fourpi, double rho_tot, void *stream) {
  if (!d_rho || !d_f2 || N < 1) return fail(RAMSES_AMD_EINVAL, "bad argument");
  MGCHK(mg_launch_rhs(d_rho, d_f2, (long)N, fourpi, rho_tot, reinterpret_cast<hipStream_t>(stream)), "mg rhs launch");
  return 0;
}
int ramses_amd_mg_restrict_ghost(const double *d_res_f, double *d_rhs_c, int nfx, int nfy, int nfz, int ngf, int ngc, void *stream) {
  if (!d_res_f || !d_rhs_c || !brick_extent_ok(nfx) || !brick_extent_ok(nfy) || !brick_extent_ok(nfz) || ngf < 0 || ngc < 0)
    return fail(RAMSES_AMD_EINVAL, "bad argument (brick extents must be powers of two)");
  MGCHK(mg_launch_restrict_ghost(d_res_f, d_rhs_c, nfx, nfy, nfz, ngf, ngc, reinterpret_cast<hipStream_t>(stream)), "mg restrict launch");
  return 0;
}
int ramses_amd_mg_interp_correct_ghost(double *d_phi_f, int nfx, int nfy, int nfz, int ngf, const double *d_corr_c, int ngc,
                                       int cglob, const int *coarse_origin, void *stream) {
  if (!d_phi_f || !d_corr_c || !brick_extent_ok(nfx) || !brick_extent_ok(nfy) || !brick_extent_ok(nfz) || ngf < 0)
    return fail(RAMSES_AMD_EINVAL, "bad argument (brick extents must be powers of two)");
  if (cglob == 0 && ngc < 1) return fail(RAMSES_AMD_EINVAL, "the local coarse brick needs >= 1 ghost layer");
  if (cglob != 0 && !coarse_origin) return fail(RAMSES_AMD_EINVAL, "replicated coarse level needs the origin of this rank's part");
  const int ox = coarse_origin ? coarse_origin[0] : 0, oy = coarse_origin ? coarse_origin[1] : 0, oz = coarse_origin ? coarse_origin[2] : 0;
  MGCHK(mg_launch_interp_ghost(d_phi_f, nfx, nfy, nfz, ngf, d_corr_c, ngc, cglob, ox, oy, oz, reinterpret_cast<hipStream_t>(stream)), "mg interp launch");
  return 0;
}
int ramses_amd_gradient_phi_ghost(const double *d_phi, double *d_f, int nx, int ny, int nz, int ng, double dx, void *stream) {
  if (!d_phi || !d_f || !brick_extent_ok(nx) || !brick_extent_ok(ny) || !brick_extent_ok(nz) || ng < 2)
    return fail(RAMSES_AMD_EINVAL, "gradient_phi needs a brick of power-of-two extents with 2 ghost layers of phi");
  const double a = 0.50 * 4.0 / 3.0 / dx;   // force_fine.f90:233-234
  const double b = 0.25 * 1.0 / 3.0 / dx;
  MGCHK(mg_launch_gradient_ghost(d_phi, d_f, nx, ny, nz, ng, a, b, reinterpret_cast<hipStream_t>(stream)), "gradient_phi launch");
  return 0;
}
// recursive_multigrid_coarse on a dense periodic level (a replicated coarse level
// of the distributed solver): d_u1 = correction for the right-hand side d_rhs.
// d_work: ramses_amd_mg_workspace_doubles(level + 1) doubles.
int ramses_amd_mg_coarse_solve_dense(int level, const double *d_rhs, double *d_u1, double *d_work, int safe,
                                     void *stream) {
  if (level < 1 || level > 10 || !d_rhs || !d_u1 || !d_work) return fail(RAMSES_AMD_EINVAL, "bad argument");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const size_t N = mg_level_cells(level);
  double *u1 = d_work + mg_hier_offset(level + 1, level, 0), *u2 = d_work + mg_hier_offset(level + 1, level, 1);
  MGCHK(hipMemcpyAsync(u2, d_rhs, sizeof(double) * N, hipMemcpyDeviceToDevice, s), "rhs copy");
  MGCHK(hipMemsetAsync(u1, 0, sizeof(double) * N, s), "memset");
  MGCHK(mg_coarse_cycle(d_work, level + 1, level, safe, s), "mg coarse cycle");
  MGCHK(hipMemcpyAsync(d_u1, u1, sizeof(double) * N, hipMemcpyDeviceToDevice, s), "u1 copy");
  return 0;
}

int ramses_amd_multigrid_fine_brick(int level, const double *d_rho, double rho_tot, double fourpi,
                                    double epsilon, int *safe_mode, double *d_phi, double *d_f1,
                                    double *d_f2, double *d_work, int *iters_out, double *err_out,
                                    void *stream) {
  if (level < 1 || level > 11) return fail(RAMSES_AMD_EINVAL, "multigrid level must be in [1,11] (got %d)", level);
  if (!d_rho || !d_phi || !d_f1 || !d_f2 || !d_work || !safe_mode) return fail(RAMSES_AMD_EINVAL, "NULL argument");
  const int MAXITER = 10, ngs_fine = 2;          // multigrid_fine_commons.f90:34, poisson_parameters.f90
  const double SAFE_FACTOR = 0.5;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int n = 1 << level;
  const long N = (long)n * n * n;
  const double dx = std::ldexp(1.0, -level), dx2 = dx * dx;
  double *partial = d_work + mg_hier_size(level);
  double *d_norm = partial + MG_MAX_PARTIALS;
  MGCHK(mg_launch_rhs(d_rho, d_f2, N, fourpi, rho_tot, s), "mg rhs launch");
  int iter = 0;
  double err = 1.0, last_err, i_res_norm2 = 0.0, res_norm2 = 0.0;
  const bool fused = (n >= MG_FUSED_MIN_N) && g_mg_fused;
  double *d_phi2 = d_work;   // fine-level ping-pong copy
  for (;;) {
    iter++;
    double *cur = d_phi, *oth = d_phi2;
    bool restricted = false;
    if (fused) {
      MGCHK(mg_smooth4(&cur, &oth, d_f2, d_f1, partial, iter == 1 ? d_norm : nullptr, n, dx, s,
                       level > 1 ? d_work + mg_hier_offset(level, level - 1, 1) : nullptr,
                       level > 1 ? d_work + mg_hier_offset(level, level - 1, 0) : nullptr, &restricted), "mg fused smoother launch");
    } else {
      for (int i = 0; i < ngs_fine; i++) {
        MGCHK(mg_launch_gs(d_phi, d_f2, n, dx2, 0, s), "mg gs launch");
        MGCHK(mg_launch_gs(d_phi, d_f2, n, dx2, 1, s), "mg gs launch");
      }
      MGCHK(mg_launch_residual(d_phi, d_f2, d_f1, n, dx, partial, iter == 1 ? d_norm : nullptr, s), "mg residual launch");
    }
    if (iter == 1) {
      MGCHK(hipMemcpyAsync(&i_res_norm2, d_norm, sizeof(double), hipMemcpyDeviceToHost, s), "norm copy");
    }
    if (level > 1) {
      if (!restricted) MGCHK(mg_launch_restrict(d_f1, d_work + mg_hier_offset(level, level - 1, 1), d_work + mg_hier_offset(level, level - 1, 0), n, s), "mg restrict launch");
      MGCHK(mg_coarse_cycle(d_work, level, level - 1, *safe_mode, s), "mg coarse cycle");
      if (!fused) MGCHK(mg_launch_interp(cur, d_work + mg_hier_offset(level, level - 1, 0), n, s), "mg interp launch");
    }
    if (fused) {
      // prolongation + post-smoothing: only the norm of the residual is needed (f(:,1) is scratch in
      // the reference and force_fine overwrites it next): it is not written to HBM
      MGCHK(mg_smooth4(&cur, &oth, d_f2, nullptr, partial, d_norm + 1, n, dx, s, nullptr, nullptr, nullptr,
                       level > 1 ? d_work + mg_hier_offset(level, level - 1, 0) : nullptr), "mg fused smoother launch");
      // the result is back in d_phi (two buffer swaps, or none)
    } else {
      for (int i = 0; i < ngs_fine; i++) {
        MGCHK(mg_launch_gs(d_phi, d_f2, n, dx2, 0, s), "mg gs launch");
        MGCHK(mg_launch_gs(d_phi, d_f2, n, dx2, 1, s), "mg gs launch");
      }
      MGCHK(mg_launch_residual(d_phi, d_f2, d_f1, n, dx, partial, d_norm + 1, s), "mg residual launch");
    }
    MGCHK(hipMemcpyAsync(&res_norm2, d_norm + 1, sizeof(double), hipMemcpyDeviceToHost, s), "norm copy");
    MGCHK(hipStreamSynchronize(s), "stream sync");
    last_err = err;
    err = std::sqrt(res_norm2 / (i_res_norm2 + 1e-20 * (rho_tot * rho_tot)));
    if (err < epsilon || iter >= MAXITER) break;
    if (err > last_err * SAFE_FACTOR && !*safe_mode) *safe_mode = 1;
  }
  if (iters_out) *iters_out = iter;
  if (err_out) *err_out = err;
  return 0;
}

// ---------------------------------------------------------------------------
// Host-array entry points: what the Fortran shims of ramses_amd/patch/ bind.
// They take the reference's own arrays (Fortran-owned, host memory), stage
// them on the device, run the brick kernels and write the results back.
// ---------------------------------------------------------------------------
extern "C++" {
namespace {
struct DevBuf {
  void *p = nullptr;
  size_t cap = 0;
  hipError_t ensure(size_t bytes) {
    if (bytes <= cap) return hipSuccess;
    if (p) { hipFree(p); p = nullptr; cap = 0; }
    hipError_t e = hipMalloc(&p, bytes);
    if (e == hipSuccess) cap = bytes;
    return e;
  }
  template <class T> T *as() { return reinterpret_cast<T *>(p); }
};
struct HostCtx {
  DevBuf uold, unew, fvec, igrid, xg, octorg, bold, bnew, bf, flag, red;
  // Poisson fields of the resident level (rho_fine -> multigrid_fine -> force_fine without PCIe):
  // rho, phi bricks, the multigrid work arrays, the oct-position -> list-index table of rho_fine
  DevBuf brho, bphi, bf1, bf2, mgwork, octidx, diag, cellvec1;
  bool res_rho_valid = false, res_phi_valid = false;      // brho / bphi hold the level's current rho / phi
  bool res_pois_host_stale = false;                       // the host arrays phi, f (and rho) are behind the device
  // device-resident level (ramses_amd_resident_*): the level brick in bold is
  // the current hydro state; the host array is stale until synced
  bool res_valid = false, res_host_stale = false, res_new_ready = false;
  bool res_grav_valid = false;   // bf holds the acceleration f(:,1:3) of the resident level
  int res_level = 0, res_ngrid = 0, res_nvar = 0;
  long res_ncell = 0, res_ncoarse = 0, res_ngridmax = 0;
  const double *res_host_uold = nullptr;
};
HostCtx g_host;
}  // namespace
}  // extern "C++"

// The staged entry points reuse the staging buffers of the resident level.  If that level holds
// the only current copy of the hydro state, dropping it would silently lose a step: refuse, as
// ramses_amd_resident_invalidate does (the caller syncs the host array first).
static int resident_release(const char *who) {
  HostCtx &H = g_host;
  if (H.res_valid && H.res_host_stale)
    return fail(RAMSES_AMD_EINVAL, "%s: level %d is resident on the device and the host array is stale; call ramses_amd_resident_sync_host_f90 first", who, H.res_level);
  H.res_valid = false;
  return 0;
}

int ramses_amd_godunov_fine_host(const ramses_amd_hydro_params *p, int ilevel, int ngrid,
                                 const int *igrid, const double *xg, int64_t ngridmax,
                                 int64_t ncoarse, int nx_loc, const double *uold, double *unew,
                                 const double *f, double dx, double dt) {
  if (!p || !igrid || !xg || !uold || !unew) return fail(RAMSES_AMD_EINVAL, "NULL argument");
  if (p->ndim != 3 || p->nvar < 5 || p->nvar > 7) return fail(RAMSES_AMD_EUNSUPPORTED, "device path implements NDIM=3, NVAR=5..7");
  if (nx_loc != 1) return fail(RAMSES_AMD_EUNSUPPORTED, "device path needs a periodic box with nx=ny=nz=1 (got nx_loc=%d)", nx_loc);
  if (ilevel < 1 || ilevel > 11) return fail(RAMSES_AMD_EINVAL, "level out of range");
  const int n = 1 << ilevel;
  const long ncells_level = (long)n * n * n;
  if ((long)ngrid * 8 != ncells_level)
    return fail(RAMSES_AMD_EUNSUPPORTED,
                "level %d is not fully refined on this rank (ngrid=%d, need %ld): AMR / multi-rank levels are not on the device yet",
                ilevel, ngrid, ncells_level / 8);
  const long ncell = ncoarse + 8 * ngridmax;
  const int nvar = p->nvar;
  if (int rc = resident_release("godunov_fine (staged brick sweep)")) return rc;   // the bricks are reused
  hipStream_t s = nullptr;
  HostCtx &H = g_host;
#define HCHK(call, what) do { hipError_t e_ = (call); if (e_ != hipSuccess) return hipfail(e_, what); } while (0)
  HCHK(H.uold.ensure(sizeof(double) * nvar * ncell), "hipMalloc uold");
  HCHK(H.unew.ensure(sizeof(double) * nvar * ncell), "hipMalloc unew");
  HCHK(H.igrid.ensure(sizeof(int) * ngrid), "hipMalloc igrid");
  HCHK(H.xg.ensure(sizeof(double) * 3 * ngridmax), "hipMalloc xg");
  HCHK(H.octorg.ensure(sizeof(long) * ngrid), "hipMalloc octorg");
  HCHK(H.bold.ensure(sizeof(double) * nvar * ncells_level), "hipMalloc brick");
  HCHK(H.bnew.ensure(sizeof(double) * nvar * ncells_level), "hipMalloc brick");
  HCHK(H.flag.ensure(sizeof(int)), "hipMalloc flag");
  HCHK(hipMemcpyAsync(H.uold.p, uold, sizeof(double) * nvar * ncell, hipMemcpyHostToDevice, s), "H2D uold");
  HCHK(hipMemcpyAsync(H.unew.p, unew, sizeof(double) * nvar * ncell, hipMemcpyHostToDevice, s), "H2D unew");
  HCHK(hipMemcpyAsync(H.igrid.p, igrid, sizeof(int) * ngrid, hipMemcpyHostToDevice, s), "H2D igrid");
  HCHK(hipMemcpyAsync(H.xg.p, xg, sizeof(double) * 3 * ngridmax, hipMemcpyHostToDevice, s), "H2D xg");
  HCHK(hipMemsetAsync(H.flag.p, 0, sizeof(int), s), "memset");
  const double skip[3] = {0.0, 0.0, 0.0};   // icoarse_min = 0 for nx = 1
  HCHK(launch_oct_origin(H.igrid.as<int>(), H.xg.as<double>(), ngridmax, ngrid, n, skip, H.octorg.as<long>(), H.flag.as<int>(), s), "oct origin launch");
  int bad = 0;
  HCHK(hipMemcpyAsync(&bad, H.flag.p, sizeof(int), hipMemcpyDeviceToHost, s), "D2H flag");
  HCHK(hipStreamSynchronize(s), "sync");
  if (bad) return fail(RAMSES_AMD_EINVAL, "%d octs of level %d do not sit on the level-%d lattice (xg inconsistent)", bad, ilevel, ilevel);
  PackArgs A;
  A.igrid = H.igrid.as<int>(); A.octorg = H.octorg.as<long>();
  A.ngrid = ngrid; A.n = n; A.nvar = nvar;
  A.ncoarse = ncoarse; A.ngridmax = ngridmax; A.ncell = ncell; A.pitch_var = ncells_level;
  A.brick = H.bold.as<double>(); A.cellvec = H.uold.as<double>();
  HCHK(launch_oct_copy(A, true, s), "gather launch");
  const double *d_grav = nullptr;
  if (f) {
    HCHK(H.fvec.ensure(sizeof(double) * 3 * ncell), "hipMalloc f");
    HCHK(H.bf.ensure(sizeof(double) * 3 * ncells_level), "hipMalloc f brick");
    HCHK(hipMemcpyAsync(H.fvec.p, f, sizeof(double) * 3 * ncell, hipMemcpyHostToDevice, s), "H2D f");
    PackArgs G = A;
    G.nvar = 3; G.brick = H.bf.as<double>(); G.cellvec = H.fvec.as<double>();
    HCHK(launch_oct_copy(G, true, s), "gather launch");
    d_grav = H.bf.as<double>();
  }
  ramses_amd_brick b;
  ramses_amd_brick_dense(&b, n, n, n, 0);
  if (int rc = ramses_amd_godunov_brick(p, &b, H.bold.as<double>(), d_grav, H.bnew.as<double>(), dx, dt, s)) return rc;
  // The reference adds flux differences to the unew that set_unew prepared
  // (= uold on active cells); the brick kernel returns uold + differences, so
  // scattering it over unew's active cells gives the same array.
  A.brick = H.bnew.as<double>(); A.cellvec = H.unew.as<double>();
  HCHK(launch_oct_copy(A, false, s), "scatter launch");
  HCHK(hipMemcpyAsync(unew, H.unew.p, sizeof(double) * nvar * ncell, hipMemcpyDeviceToHost, s), "D2H unew");
  HCHK(hipStreamSynchronize(s), "sync");
#undef HCHK
  return 0;
}

// godunov_fine(ilevel) on an AMR level: the level is partially refined and/or
// has refined cells (hydro/godunov_fine.f90:486-911, every branch of godfine1:
// interpolated stencil cells, zeroed fluxes at refined interfaces, += onto the
// unew that already holds the finer level's corrections, corrections owed to
// the coarser level).  Works directly on the reference's tree arrays.
// workspace of the device entry point, in bytes
// (coarse-correction records, their targets, oct -> list position, the father-oct groups and their counter; then, 128-byte
//  aligned, the packed oct records of the grouped kernel)
static size_t amr_ws_pack_offset(int ngrid, int64_t ngridmax) {
  const size_t head = sizeof(double) * (size_t)ngrid * 6 * 4 * 9 + sizeof(int) * (size_t)ngrid * 6 + sizeof(int) * (size_t)ngridmax +
                      sizeof(int) * ((size_t)ngrid + 16) + 64;
  return (head + 127) / 128 * 128;
}
static size_t amr_ws_walk_offset(int ngrid, int64_t ngridmax) {
  const size_t o = amr_ws_pack_offset(ngrid, ngridmax) + sizeof(double) * (size_t)AMR_PACK_REC_MAX * (size_t)ngrid;
  return (o + 127) / 128 * 128;
}
int64_t ramses_amd_godunov_fine_amr_workspace(int ngrid, int64_t ngridmax) {
  if (ngrid < 0 || ngridmax < 1) return fail(RAMSES_AMD_EINVAL, "bad argument");
  // coarse-correction records, their targets, oct -> list position, the father-oct groups and their counter; the packed
  // oct records; the father-cell walk table of the groups (192 ints per father oct, at most one group per oct of the list)
  return (int64_t)(amr_ws_walk_offset(ngrid, ngridmax) + sizeof(int) * 192 * (size_t)ngrid);
}

static int amr_check(const ramses_amd_hydro_params *p, int ilevel, int nvector, int interpol_var, int interpol_type) {
  if (p->ndim != 3 || p->nvar < 5 || p->nvar > 7) return fail(RAMSES_AMD_EUNSUPPORTED, "AMR device sweep implements NDIM=3, NVAR=5..7");
  if (p->scheme != 0 && p->scheme != 1) return fail(RAMSES_AMD_EINVAL, "unknown scheme %d", p->scheme);
  if (p->scheme == 1 && p->nvar != 5) return fail(RAMSES_AMD_EUNSUPPORTED, "passive scalars with scheme='plmde' are not on the device yet");
  if (p->difmag < 0.0) return fail(RAMSES_AMD_EINVAL, "difmag must be >= 0");
  if (ilevel < 3) return fail(RAMSES_AMD_EUNSUPPORTED, "AMR device sweep needs ilevel >= 3 (father cells inside octs); got %d", ilevel);
  if (nvector < 1) return fail(RAMSES_AMD_EINVAL, "nvector must be >= 1");
  if (interpol_var < 0 || interpol_var > 2 || interpol_type < 0 || interpol_type > 4) return fail(RAMSES_AMD_EINVAL, "interpol_var/interpol_type out of range");
  return 0;
}

// all arrays resident on the device; d_work: ramses_amd_godunov_fine_amr_workspace bytes.
// d_err (one int, zeroed by the caller) counts tree inconsistencies.
int ramses_amd_godunov_fine_amr_device(const ramses_amd_hydro_params *p, int ilevel, int ngrid,
                                       const int *d_igrid, const int *d_son, const int *d_nbor,
                                       const int *d_father, int64_t ngridmax, int64_t ncoarse,
                                       const double *d_uold, double *d_unew, const double *d_grav,
                                       double *d_divu, double *d_enew, double dx, double dt,
                                       int nvector, int interpol_var, int interpol_type,
                                       void *d_work, int *d_err, void *stream) {
  if ((d_divu == nullptr) != (d_enew == nullptr)) return fail(RAMSES_AMD_EINVAL, "pressure_fix needs both divu and enew");
  if (!p || !d_igrid || !d_son || !d_nbor || !d_father || !d_uold || !d_unew || !d_work || !d_err) return fail(RAMSES_AMD_EINVAL, "NULL argument");
  if (int rc = amr_check(p, ilevel, nvector, interpol_var, interpol_type)) return rc;
  if (ngrid <= 0) return 0;
  AmrSweepArgs A;
  A.uold = d_uold; A.unew = d_unew; A.grav = d_grav; A.divu = d_divu; A.enew = d_enew;
  A.son = d_son; A.nbor = d_nbor; A.father = d_father;
  A.igrid = d_igrid; A.ngrid = ngrid; A.nvar = p->nvar; A.scheme = p->scheme;
  A.ncell = ncoarse + 8 * ngridmax; A.ncoarse = ncoarse; A.ngridmax = ngridmax;
  A.dt = dt; A.dx = dx; A.rdx = 1.0 / dx; A.difmag = p->difmag;
  { int ex; A.pow2 = (std::frexp(dx, &ex) == 0.5) ? 1 : 0; }
  A.interpol_var = interpol_var; A.interpol_type = interpol_type;
  char *w = reinterpret_cast<char *>(d_work);
  A.corr = reinterpret_cast<double *>(w);
  w += sizeof(double) * (size_t)ngrid * 6 * 4 * 9;
  A.corr_tgt = reinterpret_cast<int *>(w);
  w += sizeof(int) * (size_t)ngrid * 6;
  int *posof = reinterpret_cast<int *>(w);
  A.err = d_err;
  A.P = make_const(p);
  double *pack_area = reinterpret_cast<double *>(reinterpret_cast<char *>(d_work) + amr_ws_pack_offset(ngrid, ngridmax));
  int *walk_area = reinterpret_cast<int *>(reinterpret_cast<char *>(d_work) + amr_ws_walk_offset(ngrid, ngridmax));
  hipError_t e = launch_amr_godunov(A, p->slope_type, p->riemann, posof, nvector, reinterpret_cast<hipStream_t>(stream), pack_area, walk_area);
  if (e != hipSuccess) return hipfail(e, "AMR godunov launch");
  return 0;
}

int ramses_amd_godunov_fine_amr_host(const ramses_amd_hydro_params *p, int ilevel, int ngrid,
                                     const int *igrid, const int *son, const int *nbor,
                                     const int *father, int64_t ngridmax, int64_t ncoarse,
                                     const double *uold, double *unew, const double *f,
                                     double *divu, double *enew, double dx, double dt,
                                     int nvector, int interpol_var, int interpol_type) {
  if (!p || !igrid || !son || !nbor || !father || !uold || !unew) return fail(RAMSES_AMD_EINVAL, "NULL argument");
  if ((divu == nullptr) != (enew == nullptr)) return fail(RAMSES_AMD_EINVAL, "pressure_fix needs both divu and enew");
  if (int rc = amr_check(p, ilevel, nvector, interpol_var, interpol_type)) return rc;
  if (ngrid <= 0) return 0;
  if (int rc = resident_release("godunov_fine (tree-walking sweep)")) return rc;   // the staging buffers are reused
  const long ncell = ncoarse + 8 * ngridmax;
  hipStream_t s = nullptr;
  HostCtx &H = g_host;
  static DevBuf dson, dnbor, dfather, dwork, ddivu, denew;
#define HCHK(call, what) do { hipError_t e_ = (call); if (e_ != hipSuccess) return hipfail(e_, what); } while (0)
  const int nvar = p->nvar;
  HCHK(H.uold.ensure(sizeof(double) * nvar * ncell), "hipMalloc uold");
  HCHK(H.unew.ensure(sizeof(double) * nvar * ncell), "hipMalloc unew");
  HCHK(H.igrid.ensure(sizeof(int) * ngrid), "hipMalloc igrid");
  HCHK(dson.ensure(sizeof(int) * ncell), "hipMalloc son");
  HCHK(dnbor.ensure(sizeof(int) * 6 * ngridmax), "hipMalloc nbor");
  HCHK(dfather.ensure(sizeof(int) * ngridmax), "hipMalloc father");
  HCHK(dwork.ensure((size_t)ramses_amd_godunov_fine_amr_workspace(ngrid, ngridmax)), "hipMalloc work");
  HCHK(H.flag.ensure(sizeof(int)), "hipMalloc flag");
  HCHK(hipMemcpyAsync(H.uold.p, uold, sizeof(double) * nvar * ncell, hipMemcpyHostToDevice, s), "H2D uold");
  HCHK(hipMemcpyAsync(H.unew.p, unew, sizeof(double) * nvar * ncell, hipMemcpyHostToDevice, s), "H2D unew");
  HCHK(hipMemcpyAsync(H.igrid.p, igrid, sizeof(int) * ngrid, hipMemcpyHostToDevice, s), "H2D igrid");
  HCHK(hipMemcpyAsync(dson.p, son, sizeof(int) * ncell, hipMemcpyHostToDevice, s), "H2D son");
  HCHK(hipMemcpyAsync(dnbor.p, nbor, sizeof(int) * 6 * ngridmax, hipMemcpyHostToDevice, s), "H2D nbor");
  HCHK(hipMemcpyAsync(dfather.p, father, sizeof(int) * ngridmax, hipMemcpyHostToDevice, s), "H2D father");
  HCHK(hipMemsetAsync(H.flag.p, 0, sizeof(int), s), "memset");
  const double *d_grav = nullptr;
  if (f) {
    HCHK(H.fvec.ensure(sizeof(double) * 3 * ncell), "hipMalloc f");
    HCHK(hipMemcpyAsync(H.fvec.p, f, sizeof(double) * 3 * ncell, hipMemcpyHostToDevice, s), "H2D f");
    d_grav = H.fvec.as<double>();
  }
  double *d_divu = nullptr, *d_enew = nullptr;
  if (divu) {
    HCHK(ddivu.ensure(sizeof(double) * ncell), "hipMalloc divu");
    HCHK(denew.ensure(sizeof(double) * ncell), "hipMalloc enew");
    HCHK(hipMemcpyAsync(ddivu.p, divu, sizeof(double) * ncell, hipMemcpyHostToDevice, s), "H2D divu");
    HCHK(hipMemcpyAsync(denew.p, enew, sizeof(double) * ncell, hipMemcpyHostToDevice, s), "H2D enew");
    d_divu = ddivu.as<double>(); d_enew = denew.as<double>();
  }
  if (int rc = ramses_amd_godunov_fine_amr_device(p, ilevel, ngrid, H.igrid.as<int>(), dson.as<int>(), dnbor.as<int>(),
                                                  dfather.as<int>(), ngridmax, ncoarse, H.uold.as<double>(),
                                                  H.unew.as<double>(), d_grav, d_divu, d_enew, dx, dt, nvector, interpol_var, interpol_type,
                                                  dwork.p, H.flag.as<int>(), s)) return rc;
  int bad = 0;
  HCHK(hipMemcpyAsync(&bad, H.flag.p, sizeof(int), hipMemcpyDeviceToHost, s), "D2H flag");
  HCHK(hipMemcpyAsync(unew, H.unew.p, sizeof(double) * nvar * ncell, hipMemcpyDeviceToHost, s), "D2H unew");
  if (divu) {
    HCHK(hipMemcpyAsync(divu, ddivu.p, sizeof(double) * ncell, hipMemcpyDeviceToHost, s), "D2H divu");
    HCHK(hipMemcpyAsync(enew, denew.p, sizeof(double) * ncell, hipMemcpyDeviceToHost, s), "D2H enew");
  }
  HCHK(hipStreamSynchronize(s), "sync");
#undef HCHK
  if (bad) return fail(RAMSES_AMD_EINVAL, "level %d: %d of the 3^3 father cells of an oct do not exist (tree inconsistent)", ilevel, bad);
  return 0;
}

// Fortran-friendly variant of the AMR entry: f is always a valid array (ignored when has_f==0)
int ramses_amd_godunov_fine_amr_f90(const ramses_amd_hydro_params *p, int ilevel, int ngrid,
                                    const int *igrid, const int *son, const int *nbor,
                                    const int *father, int64_t ngridmax, int64_t ncoarse,
                                    const double *uold, double *unew, const double *f_or_dummy, int has_f,
                                    double *divu_or_dummy, double *enew_or_dummy, int has_pfix,
                                    double dx, double dt, int nvector, int interpol_var, int interpol_type) {
  return ramses_amd_godunov_fine_amr_host(p, ilevel, ngrid, igrid, son, nbor, father, ngridmax, ncoarse, uold, unew,
                                          has_f ? f_or_dummy : nullptr, has_pfix ? divu_or_dummy : nullptr,
                                          has_pfix ? enew_or_dummy : nullptr, dx, dt, nvector, interpol_var, interpol_type);
}

// Fortran-friendly variant: f is always a valid array (ignored when has_f==0)
int ramses_amd_godunov_fine_f90(const ramses_amd_hydro_params *p, int ilevel, int ngrid,
                                const int *igrid, const double *xg, int64_t ngridmax,
                                int64_t ncoarse, int nx_loc, const double *uold, double *unew,
                                const double *f_or_dummy, int has_f, double dx, double dt) {
  return ramses_amd_godunov_fine_host(p, ilevel, ngrid, igrid, xg, ngridmax, ncoarse, nx_loc, uold, unew,
                                      has_f ? f_or_dummy : nullptr, dx, dt);
}

// multigrid_fine(ilevel,icount) on the reference's own arrays (levelmin of a
// periodic single-rank run: first guess phi = 0, every cell unmasked).
int ramses_amd_multigrid_fine_f90(int ilevel, int ngrid, const int *igrid, const double *xg,
                                  int64_t ngridmax, int64_t ncoarse, int nx_loc, const double *rho,
                                  double *phi, double rho_tot, double fourpi, double epsilon,
                                  int *safe_mode, int *iters, double *err) {
  if (!igrid || !xg || !rho || !phi || !safe_mode) return fail(RAMSES_AMD_EINVAL, "NULL argument");
  if (nx_loc != 1) return fail(RAMSES_AMD_EUNSUPPORTED, "device multigrid needs a periodic box with nx=ny=nz=1 (got nx_loc=%d)", nx_loc);
  if (ilevel < 1 || ilevel > 11) return fail(RAMSES_AMD_EINVAL, "level out of range");
  const int n = 1 << ilevel;
  const long N = (long)n * n * n;
  if ((long)ngrid * 8 != N)
    return fail(RAMSES_AMD_EUNSUPPORTED, "level %d is not fully refined on this rank (ngrid=%d): masked/AMR multigrid is not on the device yet", ilevel, ngrid);
  const long ncell = ncoarse + 8 * ngridmax;
  hipStream_t s = nullptr;
  HostCtx &H = g_host;
  // igrid/xg/octorg are shared with the resident level: the same level rewrites them with the same contents
  if (H.res_valid && !(H.res_level == ilevel && H.res_ngrid == ngrid && H.res_ncell == ncell))
    if (int rc = resident_release("multigrid_fine")) return rc;
  static DevBuf rhovec, phivec, brho, bphi, bf1, bf2, work;
#define HCHK(call, what) do { hipError_t e_ = (call); if (e_ != hipSuccess) return hipfail(e_, what); } while (0)
  const int64_t nwork = ramses_amd_mg_workspace_doubles(ilevel);
  HCHK(rhovec.ensure(sizeof(double) * ncell), "hipMalloc");
  HCHK(phivec.ensure(sizeof(double) * ncell), "hipMalloc");
  HCHK(brho.ensure(sizeof(double) * N), "hipMalloc");
  HCHK(bphi.ensure(sizeof(double) * N), "hipMalloc");
  HCHK(bf1.ensure(sizeof(double) * N), "hipMalloc");
  HCHK(bf2.ensure(sizeof(double) * N), "hipMalloc");
  HCHK(work.ensure(sizeof(double) * nwork), "hipMalloc");
  HCHK(H.igrid.ensure(sizeof(int) * ngrid), "hipMalloc igrid");
  HCHK(H.xg.ensure(sizeof(double) * 3 * ngridmax), "hipMalloc xg");
  HCHK(H.octorg.ensure(sizeof(long) * ngrid), "hipMalloc octorg");
  HCHK(H.flag.ensure(sizeof(int)), "hipMalloc flag");
  HCHK(hipMemcpyAsync(rhovec.p, rho, sizeof(double) * ncell, hipMemcpyHostToDevice, s), "H2D rho");
  HCHK(hipMemcpyAsync(phivec.p, phi, sizeof(double) * ncell, hipMemcpyHostToDevice, s), "H2D phi");
  HCHK(hipMemcpyAsync(H.igrid.p, igrid, sizeof(int) * ngrid, hipMemcpyHostToDevice, s), "H2D igrid");
  HCHK(hipMemcpyAsync(H.xg.p, xg, sizeof(double) * 3 * ngridmax, hipMemcpyHostToDevice, s), "H2D xg");
  HCHK(hipMemsetAsync(H.flag.p, 0, sizeof(int), s), "memset");
  const double skip[3] = {0.0, 0.0, 0.0};
  HCHK(launch_oct_origin(H.igrid.as<int>(), H.xg.as<double>(), ngridmax, ngrid, n, skip, H.octorg.as<long>(), H.flag.as<int>(), s), "oct origin launch");
  int bad = 0;
  HCHK(hipMemcpyAsync(&bad, H.flag.p, sizeof(int), hipMemcpyDeviceToHost, s), "D2H flag");
  HCHK(hipStreamSynchronize(s), "sync");
  if (bad) return fail(RAMSES_AMD_EINVAL, "%d octs of level %d do not sit on the level lattice", bad, ilevel);
  PackArgs A;
  A.igrid = H.igrid.as<int>(); A.octorg = H.octorg.as<long>();
  A.ngrid = ngrid; A.n = n; A.nvar = 1;
  A.ncoarse = ncoarse; A.ngridmax = ngridmax; A.ncell = ncell; A.pitch_var = N;
  A.brick = brho.as<double>(); A.cellvec = rhovec.as<double>();
  HCHK(launch_oct_copy(A, true, s), "gather launch");
  HCHK(hipMemsetAsync(bphi.p, 0, sizeof(double) * N, s), "memset phi");   // make_multipole_phi, periodic: phi = 0
  if (int rc = ramses_amd_multigrid_fine_brick(ilevel, brho.as<double>(), rho_tot, fourpi, epsilon, safe_mode,
                                               bphi.as<double>(), bf1.as<double>(), bf2.as<double>(),
                                               work.as<double>(), iters, err, s)) return rc;
  A.brick = bphi.as<double>(); A.cellvec = phivec.as<double>();
  HCHK(launch_oct_copy(A, false, s), "scatter launch");
  HCHK(hipMemcpyAsync(phi, phivec.p, sizeof(double) * ncell, hipMemcpyDeviceToHost, s), "D2H phi");
  HCHK(hipStreamSynchronize(s), "sync");
#undef HCHK
  return 0;
}

// force_fine(ilevel,icount) on the reference's own arrays (fully refined periodic level of a
// single-rank run, gravity_type = 0): f(:,1:3) = gradient_phi of phi (poisson/force_fine.f90:
// 199-324, 5-point differences); the caller keeps the diagnostics of :158-190 (epot, rho_max).
int ramses_amd_force_fine_f90(int ilevel, int ngrid, const int *igrid, const double *xg,
                              int64_t ngridmax, int64_t ncoarse, int nx_loc, const double *phi, double *f,
                              const double *rho, const int *son_or_dummy, int has_son, double fact, double *diag2) {
  if (!igrid || !xg || !phi || !f || !rho || !diag2 || (has_son && !son_or_dummy)) return fail(RAMSES_AMD_EINVAL, "NULL argument");
  if (nx_loc != 1) return fail(RAMSES_AMD_EUNSUPPORTED, "device force_fine needs a periodic box with nx=ny=nz=1 (got nx_loc=%d)", nx_loc);
  if (ilevel < 2 || ilevel > 11) return fail(RAMSES_AMD_EINVAL, "level out of range");
  const int n = 1 << ilevel;
  const long N = (long)n * n * n;
  if ((long)ngrid * 8 != N)
    return fail(RAMSES_AMD_EUNSUPPORTED, "level %d is not fully refined on this rank (ngrid=%d)", ilevel, ngrid);
  const long ncell = ncoarse + 8 * ngridmax;
  hipStream_t s = nullptr;
  HostCtx &H = g_host;
  static DevBuf phivec, fvec3, bphi, bf;
#define HCHK(call, what) do { hipError_t e_ = (call); if (e_ != hipSuccess) return hipfail(e_, what); } while (0)
  HCHK(phivec.ensure(sizeof(double) * ncell), "hipMalloc");
  HCHK(fvec3.ensure(sizeof(double) * 3 * ncell), "hipMalloc");
  HCHK(bphi.ensure(sizeof(double) * N), "hipMalloc");
  HCHK(bf.ensure(sizeof(double) * 3 * N), "hipMalloc");
  HCHK(H.igrid.ensure(sizeof(int) * ngrid), "hipMalloc igrid");
  HCHK(H.xg.ensure(sizeof(double) * 3 * ngridmax), "hipMalloc xg");
  HCHK(H.octorg.ensure(sizeof(long) * ngrid), "hipMalloc octorg");
  HCHK(H.flag.ensure(sizeof(int)), "hipMalloc flag");
  HCHK(hipMemcpyAsync(phivec.p, phi, sizeof(double) * ncell, hipMemcpyHostToDevice, s), "H2D phi");
  HCHK(hipMemcpyAsync(fvec3.p, f, sizeof(double) * 3 * ncell, hipMemcpyHostToDevice, s), "H2D f");   // cells off the level keep their values
  HCHK(hipMemcpyAsync(H.igrid.p, igrid, sizeof(int) * ngrid, hipMemcpyHostToDevice, s), "H2D igrid");
  HCHK(hipMemcpyAsync(H.xg.p, xg, sizeof(double) * 3 * ngridmax, hipMemcpyHostToDevice, s), "H2D xg");
  HCHK(hipMemsetAsync(H.flag.p, 0, sizeof(int), s), "memset");
  const double skip[3] = {0.0, 0.0, 0.0};
  HCHK(launch_oct_origin(H.igrid.as<int>(), H.xg.as<double>(), ngridmax, ngrid, n, skip, H.octorg.as<long>(), H.flag.as<int>(), s), "oct origin launch");
  int bad = 0;
  HCHK(hipMemcpyAsync(&bad, H.flag.p, sizeof(int), hipMemcpyDeviceToHost, s), "D2H flag");
  HCHK(hipStreamSynchronize(s), "sync");
  if (bad) return fail(RAMSES_AMD_EINVAL, "%d octs of level %d do not sit on the level lattice", bad, ilevel);
  // igrid/xg/octorg are shared with the resident level: the same level rewrites them with the
  // same contents, anything else ends the residency
  const bool resident = H.res_valid && H.res_level == ilevel && H.res_ngrid == ngrid && H.res_ncell == ncell;
  if (!resident) if (int rc = resident_release("force_fine")) return rc;
  double *d_f = bf.as<double>();
  if (resident) {
    // the acceleration of the resident level is rewritten in place (synchro_hydro_fine,
    // courant_fine, godunov_fine and set_uold read it there)
    HCHK(H.bf.ensure(sizeof(double) * 3 * N), "hipMalloc f brick");
    d_f = H.bf.as<double>();
  }
  PackArgs A;
  A.igrid = H.igrid.as<int>(); A.octorg = H.octorg.as<long>();
  A.ngrid = ngrid; A.n = n; A.nvar = 1;
  A.ncoarse = ncoarse; A.ngridmax = ngridmax; A.ncell = ncell; A.pitch_var = N;
  A.brick = bphi.as<double>(); A.cellvec = phivec.as<double>();
  HCHK(launch_oct_copy(A, true, s), "gather launch");
  if (int rc = ramses_amd_gradient_phi_brick(ilevel, bphi.as<double>(), d_f, s)) return rc;
  if (resident) H.res_grav_valid = true;
  A.nvar = 3;
  A.brick = d_f; A.cellvec = fvec3.as<double>();
  HCHK(launch_oct_copy(A, false, s), "scatter launch");
  HCHK(hipMemcpyAsync(f, fvec3.p, sizeof(double) * 3 * ncell, hipMemcpyDeviceToHost, s), "D2H f");
  // diagnostics (:158-190): potential energy of the leaf cells and maximum density, reduced on the device
  {
    static DevBuf rhovec, brho, sonvec, bleaf;
    HCHK(rhovec.ensure(sizeof(double) * ncell), "hipMalloc");
    HCHK(brho.ensure(sizeof(double) * N), "hipMalloc");
    HCHK(H.diag.ensure(sizeof(double) * (FORCE_DIAG_SCRATCH + 2)), "hipMalloc");
    HCHK(hipMemcpyAsync(rhovec.p, rho, sizeof(double) * ncell, hipMemcpyHostToDevice, s), "H2D rho");
    A.nvar = 1; A.brick = brho.as<double>(); A.cellvec = rhovec.as<double>();
    HCHK(launch_oct_copy(A, true, s), "gather launch");
    const int *d_leaf = nullptr;
    if (has_son) {
      // son(icell) == 0 marks a leaf: gathered as 8-byte words through the same kernel (son viewed as doubles would
      // need pairs of cells), so a small dedicated pass: leaf[b] = (son[icell] == 0)
      HCHK(sonvec.ensure(sizeof(int) * ncell), "hipMalloc");
      HCHK(bleaf.ensure(sizeof(int) * N), "hipMalloc");
      HCHK(hipMemcpyAsync(sonvec.p, son_or_dummy, sizeof(int) * ncell, hipMemcpyHostToDevice, s), "H2D son");
      HCHK(launch_oct_leaf(A, sonvec.as<int>(), bleaf.as<int>(), s), "leaf launch");
      d_leaf = bleaf.as<int>();
    }
    double *scratch = H.diag.as<double>();
    HCHK(launch_force_diag(d_f, brho.as<double>(), d_leaf, N, fact, scratch, scratch + FORCE_DIAG_SCRATCH, s), "force diagnostics launch");
    HCHK(hipMemcpyAsync(diag2, scratch + FORCE_DIAG_SCRATCH, sizeof(double) * 2, hipMemcpyDeviceToHost, s), "D2H diag");
  }
  HCHK(hipStreamSynchronize(s), "sync");
#undef HCHK
  return 0;
}

// ---------------------------------------------------------------------------
// Device-resident level: courant_fine, set_unew,
// godunov_fine and set_uold of a fully refined periodic level without the
// state crossing PCIe every step.  The Fortran shims call these instead of the
// staging entry points when the run configuration guarantees that no host
// routine touches uold between two hydro steps (ramses_amd_iface.f90:
// ramses_amd_resident()); the host array is refreshed on demand
// (ramses_amd_resident_sync_host_f90, called by the backup_hydro shim).
// ---------------------------------------------------------------------------
#define HCHK(call, what) do { hipError_t e_ = (call); if (e_ != hipSuccess) return hipfail(e_, what); } while (0)
static int resident_ensure(const ramses_amd_hydro_params *p, int ilevel, int ngrid, const int *igrid,
                           const double *xg, int64_t ngridmax, int64_t ncoarse, int nx_loc,
                           const double *uold) {
  if (!p || !igrid || !xg || !uold) return fail(RAMSES_AMD_EINVAL, "NULL argument");
  if (p->ndim != 3 || p->nvar < 5 || p->nvar > 7) return fail(RAMSES_AMD_EUNSUPPORTED, "device path implements NDIM=3, NVAR=5..7");
  if (nx_loc != 1) return fail(RAMSES_AMD_EUNSUPPORTED, "device path needs a periodic box with nx=ny=nz=1 (got nx_loc=%d)", nx_loc);
  if (ilevel < 1 || ilevel > 11) return fail(RAMSES_AMD_EINVAL, "level out of range");
  const int n = 1 << ilevel;
  const long N = (long)n * n * n;
  if ((long)ngrid * 8 != N)
    return fail(RAMSES_AMD_EUNSUPPORTED, "level %d is not fully refined on this rank (ngrid=%d, need %ld)", ilevel, ngrid, N / 8);
  HostCtx &H = g_host;
  const long ncell = ncoarse + 8 * ngridmax;
  const int nvar = p->nvar;
  if (H.res_valid && H.res_level == ilevel && H.res_ngrid == ngrid && H.res_nvar == nvar && H.res_ncell == ncell &&
      H.res_host_uold == uold)
    return 0;
  hipStream_t s = nullptr;
  HCHK(H.uold.ensure(sizeof(double) * nvar * ncell), "hipMalloc uold");
  HCHK(H.igrid.ensure(sizeof(int) * ngrid), "hipMalloc igrid");
  HCHK(H.xg.ensure(sizeof(double) * 3 * ngridmax), "hipMalloc xg");
  HCHK(H.octorg.ensure(sizeof(long) * ngrid), "hipMalloc octorg");
  HCHK(H.bold.ensure(sizeof(double) * nvar * N), "hipMalloc brick");
  HCHK(H.bnew.ensure(sizeof(double) * nvar * N), "hipMalloc brick");
  HCHK(H.flag.ensure(sizeof(int)), "hipMalloc flag");
  HCHK(H.red.ensure(sizeof(double) * 4), "hipMalloc reduction");
  HCHK(hipMemcpyAsync(H.uold.p, uold, sizeof(double) * nvar * ncell, hipMemcpyHostToDevice, s), "H2D uold");
  HCHK(hipMemcpyAsync(H.igrid.p, igrid, sizeof(int) * ngrid, hipMemcpyHostToDevice, s), "H2D igrid");
  HCHK(hipMemcpyAsync(H.xg.p, xg, sizeof(double) * 3 * ngridmax, hipMemcpyHostToDevice, s), "H2D xg");
  HCHK(hipMemsetAsync(H.flag.p, 0, sizeof(int), s), "memset");
  const double skip[3] = {0.0, 0.0, 0.0};
  HCHK(launch_oct_origin(H.igrid.as<int>(), H.xg.as<double>(), ngridmax, ngrid, n, skip, H.octorg.as<long>(), H.flag.as<int>(), s), "oct origin launch");
  int bad = 0;
  HCHK(hipMemcpyAsync(&bad, H.flag.p, sizeof(int), hipMemcpyDeviceToHost, s), "D2H flag");
  HCHK(hipStreamSynchronize(s), "sync");
  if (bad) return fail(RAMSES_AMD_EINVAL, "%d octs of level %d do not sit on the level lattice (xg inconsistent)", bad, ilevel);
  PackArgs A;
  A.igrid = H.igrid.as<int>(); A.octorg = H.octorg.as<long>();
  A.ngrid = ngrid; A.n = n; A.nvar = nvar;
  A.ncoarse = ncoarse; A.ngridmax = ngridmax; A.ncell = ncell; A.pitch_var = N;
  A.brick = H.bold.as<double>(); A.cellvec = H.uold.as<double>();
  HCHK(launch_oct_copy(A, true, s), "gather launch");
  H.res_valid = true; H.res_host_stale = false; H.res_new_ready = false; H.res_grav_valid = false;
  H.res_rho_valid = false; H.res_phi_valid = false; H.res_pois_host_stale = false;
  H.res_level = ilevel; H.res_ngrid = ngrid; H.res_nvar = nvar; H.res_ncell = ncell;
  H.res_ncoarse = ncoarse; H.res_ngridmax = ngridmax; H.res_host_uold = uold;
  return 0;
}

// courant_fine (hydro/courant_fine.f90:1-159) on the resident level:
// out4 = {dt_loc, mass_loc, sum(E*vol) ("ekin_loc"), eint_loc}.  dt is
// bit-identical (min is order independent); the three sums are accumulated in
// a different order than the reference's serial loop (diagnostics only).
int ramses_amd_resident_courant_f90(const ramses_amd_hydro_params *p, int ilevel, int ngrid,
                                    const int *igrid, const double *xg, int64_t ngridmax,
                                    int64_t ncoarse, int nx_loc, const double *uold, double dx,
                                    double dt_in, double *out4) {
  if (!out4) return fail(RAMSES_AMD_EINVAL, "NULL argument");
  if (int rc = resident_ensure(p, ilevel, ngrid, igrid, xg, ngridmax, ncoarse, nx_loc, uold)) return rc;
  HostCtx &H = g_host;
  const int n = 1 << ilevel;
  hipStream_t s = nullptr;
  ramses_amd_brick b;
  ramses_amd_brick_dense(&b, n, n, n, 0);
  if (int rc = ramses_amd_courant_init(p, dx, H.red.as<double>(), s)) return rc;
  if (int rc = ramses_amd_courant_brick(p, &b, H.bold.as<double>(), nullptr, dx, H.red.as<double>(), s)) return rc;
  HCHK(hipMemcpyAsync(out4, H.red.p, sizeof(double) * 4, hipMemcpyDeviceToHost, s), "D2H courant");
  HCHK(hipStreamSynchronize(s), "sync");
  if (dt_in < out4[0]) out4[0] = dt_in;   // dt_loc starts from dtnew(ilevel)
  return 0;
}

// set_unew + godunov_fine on the resident level: bold -> bnew (= uold + flux differences)
int ramses_amd_resident_godunov_f90(const ramses_amd_hydro_params *p, int ilevel, int ngrid,
                                    const int *igrid, const double *xg, int64_t ngridmax,
                                    int64_t ncoarse, int nx_loc, const double *uold, double dx, double dt) {
  if (int rc = resident_ensure(p, ilevel, ngrid, igrid, xg, ngridmax, ncoarse, nx_loc, uold)) return rc;
  HostCtx &H = g_host;
  const int n = 1 << ilevel;
  ramses_amd_brick b;
  ramses_amd_brick_dense(&b, n, n, n, 0);
  if (int rc = ramses_amd_godunov_brick(p, &b, H.bold.as<double>(), nullptr, H.bnew.as<double>(), dx, dt, nullptr)) return rc;
  H.res_new_ready = true;
  return 0;
}

// set_uold on the resident level: the new state becomes the current one
int ramses_amd_resident_set_uold_f90(int ilevel) {
  HostCtx &H = g_host;
  if (!H.res_valid || H.res_level != ilevel) return fail(RAMSES_AMD_EINVAL, "set_uold: level %d is not resident", ilevel);
  if (!H.res_new_ready) return fail(RAMSES_AMD_EINVAL, "set_uold: no godunov_fine result pending on level %d", ilevel);
  DevBuf t = H.bold; H.bold = H.bnew; H.bnew = t;
  H.res_new_ready = false;
  H.res_host_stale = true;
  return 0;
}

// refresh the host array from the resident level (no-op when it is current)
int ramses_amd_resident_sync_host_f90(double *uold) {
  HostCtx &H = g_host;
  if (!H.res_valid || !H.res_host_stale) return 0;
  if (uold != H.res_host_uold) return fail(RAMSES_AMD_EINVAL, "sync_host: not the array the level was loaded from");
  const int n = 1 << H.res_level;
  const long N = (long)n * n * n;
  hipStream_t s = nullptr;
  PackArgs A;
  A.igrid = H.igrid.as<int>(); A.octorg = H.octorg.as<long>();
  A.ngrid = H.res_ngrid; A.n = n; A.nvar = H.res_nvar;
  A.ncoarse = H.res_ncoarse; A.ngridmax = H.res_ngridmax; A.ncell = H.res_ncell; A.pitch_var = N;
  A.brick = H.bold.as<double>(); A.cellvec = H.uold.as<double>();
  HCHK(launch_oct_copy(A, false, s), "scatter launch");
  // cells of other levels come back with the values they were loaded with
  HCHK(hipMemcpyAsync(uold, H.uold.p, sizeof(double) * H.res_nvar * H.res_ncell, hipMemcpyDeviceToHost, s), "D2H uold");
  HCHK(hipStreamSynchronize(s), "sync");
  H.res_host_stale = false;
  return 0;
}

// ---- gravity on the resident level (first part) -------------------------
// The acceleration lives in bf next to the hydro state: loaded from the host array on first use,
// rewritten by ramses_amd_force_fine_f90 every step.
static int resident_ensure_grav(const double *f) {
  HostCtx &H = g_host;
  if (H.res_grav_valid) return 0;
  if (!f) return fail(RAMSES_AMD_EINVAL, "NULL argument");
  const int n = 1 << H.res_level;
  const long N = (long)n * n * n;
  hipStream_t s = nullptr;
  HCHK(H.fvec.ensure(sizeof(double) * 3 * H.res_ncell), "hipMalloc f");
  HCHK(H.bf.ensure(sizeof(double) * 3 * N), "hipMalloc f brick");
  HCHK(hipMemcpyAsync(H.fvec.p, f, sizeof(double) * 3 * H.res_ncell, hipMemcpyHostToDevice, s), "H2D f");
  PackArgs G;
  G.igrid = H.igrid.as<int>(); G.octorg = H.octorg.as<long>();
  G.ngrid = H.res_ngrid; G.n = n; G.nvar = 3;
  G.ncoarse = H.res_ncoarse; G.ngridmax = H.res_ngridmax; G.ncell = H.res_ncell; G.pitch_var = N;
  G.brick = H.bf.as<double>(); G.cellvec = H.fvec.as<double>();
  HCHK(launch_oct_copy(G, true, s), "gather launch");
  H.res_grav_valid = true;
  return 0;
}

// synchro_hydro_fine(ilevel,dteff,1) (hydro/synchro_hydro_fine.f90:5-136) on the resident level
int ramses_amd_resident_synchro_f90(const ramses_amd_hydro_params *p, int ilevel, int ngrid,
                                    const int *igrid, const double *xg, int64_t ngridmax,
                                    int64_t ncoarse, int nx_loc, const double *uold, const double *f, double dteff) {
  if (int rc = resident_ensure(p, ilevel, ngrid, igrid, xg, ngridmax, ncoarse, nx_loc, uold)) return rc;
  if (int rc = resident_ensure_grav(f)) return rc;
  HostCtx &H = g_host;
  const int n = 1 << ilevel;
  const long N = (long)n * n * n;
  if (H.res_new_ready) return fail(RAMSES_AMD_EINVAL, "synchro_hydro_fine between godunov_fine and set_uold");
  hipError_t e = launch_synchro_hydro(H.bold.as<double>(), H.bf.as<double>(), N, dteff, p->smallr, nullptr);
  if (e != hipSuccess) return hipfail(e, "synchro_hydro launch");
  H.res_host_stale = true;
  return 0;
}

// courant_fine with the gravity term of cmpdt (hydro/courant_fine.f90:77-85)
int ramses_amd_resident_courant_grav_f90(const ramses_amd_hydro_params *p, int ilevel, int ngrid,
                                         const int *igrid, const double *xg, int64_t ngridmax,
                                         int64_t ncoarse, int nx_loc, const double *uold, const double *f,
                                         double dx, double dt_in, double *out4) {
  if (!out4) return fail(RAMSES_AMD_EINVAL, "NULL argument");
  if (int rc = resident_ensure(p, ilevel, ngrid, igrid, xg, ngridmax, ncoarse, nx_loc, uold)) return rc;
  if (int rc = resident_ensure_grav(f)) return rc;
  HostCtx &H = g_host;
  const int n = 1 << ilevel;
  hipStream_t s = nullptr;
  ramses_amd_brick b;
  ramses_amd_brick_dense(&b, n, n, n, 0);
  if (int rc = ramses_amd_courant_init(p, dx, H.red.as<double>(), s)) return rc;
  if (int rc = ramses_amd_courant_brick(p, &b, H.bold.as<double>(), H.bf.as<double>(), dx, H.red.as<double>(), s)) return rc;
  HCHK(hipMemcpyAsync(out4, H.red.p, sizeof(double) * 4, hipMemcpyDeviceToHost, s), "D2H courant");
  HCHK(hipStreamSynchronize(s), "sync");
  if (dt_in < out4[0]) out4[0] = dt_in;
  return 0;
}

// set_unew + godunov_fine with the gravity predictor (godfine1 :637-647, ctoprim)
int ramses_amd_resident_godunov_grav_f90(const ramses_amd_hydro_params *p, int ilevel, int ngrid,
                                         const int *igrid, const double *xg, int64_t ngridmax,
                                         int64_t ncoarse, int nx_loc, const double *uold, const double *f,
                                         double dx, double dt) {
  if (int rc = resident_ensure(p, ilevel, ngrid, igrid, xg, ngridmax, ncoarse, nx_loc, uold)) return rc;
  if (int rc = resident_ensure_grav(f)) return rc;
  HostCtx &H = g_host;
  const int n = 1 << ilevel;
  ramses_amd_brick b;
  ramses_amd_brick_dense(&b, n, n, n, 0);
  if (int rc = ramses_amd_godunov_brick(p, &b, H.bold.as<double>(), H.bf.as<double>(), H.bnew.as<double>(), dx, dt, nullptr)) return rc;
  H.res_new_ready = true;
  return 0;
}

// set_uold with add_gravity_source_terms (hydro/godunov_fine.f90:160-162,237-289) before the swap
int ramses_amd_resident_set_uold_grav_f90(const ramses_amd_hydro_params *p, int ilevel, double dt) {
  HostCtx &H = g_host;
  if (!p) return fail(RAMSES_AMD_EINVAL, "NULL argument");
  if (!H.res_valid || H.res_level != ilevel) return fail(RAMSES_AMD_EINVAL, "set_uold: level %d is not resident", ilevel);
  if (!H.res_new_ready) return fail(RAMSES_AMD_EINVAL, "set_uold: no godunov_fine result pending on level %d", ilevel);
  if (!H.res_grav_valid) return fail(RAMSES_AMD_EINVAL, "set_uold: no acceleration on the device for level %d", ilevel);
  const int n = 1 << ilevel;
  const long N = (long)n * n * n;
  hipError_t e = launch_add_gravity_source(H.bnew.as<double>(), H.bold.as<double>(), H.bf.as<double>(), N, dt, p->smallr, nullptr);
  if (e != hipSuccess) return hipfail(e, "add_gravity_source launch");
  return ramses_amd_resident_set_uold_f90(ilevel);
}

// the density of the resident level back into uold(:,1) (rho_fine reads nothing else of uold)
int ramses_amd_resident_sync_density_f90(double *uold) {
  HostCtx &H = g_host;
  if (!H.res_valid || !H.res_host_stale) return 0;
  if (uold != H.res_host_uold) return fail(RAMSES_AMD_EINVAL, "sync_density: not the array the level was loaded from");
  const int n = 1 << H.res_level;
  const long N = (long)n * n * n;
  hipStream_t s = nullptr;
  PackArgs A;
  A.igrid = H.igrid.as<int>(); A.octorg = H.octorg.as<long>();
  A.ngrid = H.res_ngrid; A.n = n; A.nvar = 1;
  A.ncoarse = H.res_ncoarse; A.ngridmax = H.res_ngridmax; A.ncell = H.res_ncell; A.pitch_var = N;
  A.brick = H.bold.as<double>(); A.cellvec = H.uold.as<double>();
  HCHK(launch_oct_copy(A, false, s), "scatter launch");
  HCHK(hipMemcpyAsync(uold, H.uold.p, sizeof(double) * H.res_ncell, hipMemcpyDeviceToHost, s), "D2H density");
  HCHK(hipStreamSynchronize(s), "sync");
  return 0;   // the other variables of the host array stay stale
}

// ---- the Poisson branch of amr_step on the resident level (second part) ----------
// rho_fine's hydro deposit, multigrid_fine and force_fine read and write device bricks only; the host
// arrays rho, phi, f are refreshed on demand (backup_poisson shim -> ramses_amd_resident_sync_poisson_f90).

// rho_fine (pm/rho_fine.f90:5-226) for a hydro-only source on the resident level: rho = CIC deposit of the cell
// masses at their centres of mass (multipole_fine + cic_from_multipole), multipole(1:4) summed in the
// reference's order.  The caller sets rho_tot = multipole(1)/scale**ndim (:179).
int ramses_amd_resident_rho_fine_f90(const ramses_amd_hydro_params *p, int ilevel, int ngrid,
                                     const int *igrid, const double *xg, int64_t ngridmax,
                                     int64_t ncoarse, int nx_loc, const double *uold, double boxlen,
                                     int nvector, double *multipole4) {
  if (!multipole4) return fail(RAMSES_AMD_EINVAL, "NULL argument");
  if (nvector < 1) return fail(RAMSES_AMD_EINVAL, "nvector must be >= 1");
  if (int rc = resident_ensure(p, ilevel, ngrid, igrid, xg, ngridmax, ncoarse, nx_loc, uold)) return rc;
  HostCtx &H = g_host;
  if (H.res_new_ready) return fail(RAMSES_AMD_EINVAL, "rho_fine between godunov_fine and set_uold");
  const int n = 1 << ilevel;
  const long N = (long)n * n * n;
  hipStream_t s = nullptr;
  HCHK(H.brho.ensure(sizeof(double) * N), "hipMalloc rho brick");
  HCHK(H.octidx.ensure(sizeof(int) * (size_t)(N / 8)), "hipMalloc oct index");
  HCHK(H.diag.ensure(sizeof(double) * (FORCE_DIAG_SCRATCH + 8)), "hipMalloc");
  HCHK(launch_oct_index(H.octorg.as<long>(), ngrid, n, H.octidx.as<int>(), s), "oct index launch");
  RhoArgs A;
  A.dens = H.bold.as<double>();          // variable 1 of the resident state
  A.rho = H.brho.as<double>();
  A.octorg = H.octorg.as<long>(); A.octidx = H.octidx.as<int>();
  A.n = n; A.ngrid = ngrid; A.nvector = nvector;
  A.dx = std::ldexp(1.0, -ilevel);
  A.scale = boxlen / (double)nx_loc;
  const double dx_loc = A.dx * A.scale;
  A.vol_loc = dx_loc * dx_loc * dx_loc;
  A.smallr = p->smallr;
  HCHK(launch_rho_deposit(A, s), "rho deposit launch");
  double *d_mp = H.diag.as<double>() + FORCE_DIAG_SCRATCH + 2;
  static DevBuf mpscratch;
  HCHK(mpscratch.ensure(multipole_scratch_bytes((long)ngrid * 8)), "hipMalloc multipole scratch");
  HCHK(launch_multipole(A, d_mp, mpscratch.p, s), "multipole launch");
  HCHK(hipMemcpyAsync(multipole4, d_mp, sizeof(double) * 4, hipMemcpyDeviceToHost, s), "D2H multipole");
  HCHK(hipStreamSynchronize(s), "sync");
  H.res_rho_valid = true;
  H.res_pois_host_stale = true;
  return 0;
}

// multigrid_fine(ilevel,icount) on the resident level: source = the deposit left by ramses_amd_resident_rho_fine_f90
int ramses_amd_resident_multigrid_f90(int ilevel, double rho_tot, double fourpi, double epsilon, int *safe_mode,
                                      int *iters, double *err) {
  HostCtx &H = g_host;
  if (!safe_mode) return fail(RAMSES_AMD_EINVAL, "NULL argument");
  if (!H.res_valid || H.res_level != ilevel) return fail(RAMSES_AMD_EINVAL, "multigrid_fine: level %d is not resident", ilevel);
  if (!H.res_rho_valid) return fail(RAMSES_AMD_EINVAL, "multigrid_fine: no density deposit on the device (rho_fine)");
  const int n = 1 << ilevel;
  const long N = (long)n * n * n;
  hipStream_t s = nullptr;
  const int64_t nwork = ramses_amd_mg_workspace_doubles(ilevel);
  if (nwork < 0) return (int)nwork;
  HCHK(H.bphi.ensure(sizeof(double) * N), "hipMalloc"); HCHK(H.bf1.ensure(sizeof(double) * N), "hipMalloc");
  HCHK(H.bf2.ensure(sizeof(double) * N), "hipMalloc"); HCHK(H.mgwork.ensure(sizeof(double) * nwork), "hipMalloc");
  HCHK(hipMemsetAsync(H.bphi.p, 0, sizeof(double) * N, s), "memset phi");   // make_multipole_phi, periodic: phi = 0
  if (int rc = ramses_amd_multigrid_fine_brick(ilevel, H.brho.as<double>(), rho_tot, fourpi, epsilon, safe_mode,
                                               H.bphi.as<double>(), H.bf1.as<double>(), H.bf2.as<double>(),
                                               H.mgwork.as<double>(), iters, err, s)) return rc;
  H.res_phi_valid = true;
  H.res_pois_host_stale = true;
  return 0;
}

// force_fine(ilevel,icount) on the resident level: f = gradient_phi(phi) into the acceleration brick the hydro
// routines read; diag2 = {sum over cells and directions of fact*f**2, max |rho|} (poisson/force_fine.f90:158-190)
int ramses_amd_resident_force_fine_f90(int ilevel, double fact, double *diag2) {
  HostCtx &H = g_host;
  if (!diag2) return fail(RAMSES_AMD_EINVAL, "NULL argument");
  if (!H.res_valid || H.res_level != ilevel) return fail(RAMSES_AMD_EINVAL, "force_fine: level %d is not resident", ilevel);
  if (!H.res_phi_valid || !H.res_rho_valid) return fail(RAMSES_AMD_EINVAL, "force_fine: no potential on the device (multigrid_fine)");
  const int n = 1 << ilevel;
  const long N = (long)n * n * n;
  hipStream_t s = nullptr;
  HCHK(H.bf.ensure(sizeof(double) * 3 * N), "hipMalloc f brick");
  HCHK(H.diag.ensure(sizeof(double) * (FORCE_DIAG_SCRATCH + 8)), "hipMalloc");
  if (int rc = ramses_amd_gradient_phi_brick(ilevel, H.bphi.as<double>(), H.bf.as<double>(), s)) return rc;
  H.res_grav_valid = true;
  double *scratch = H.diag.as<double>();
  HCHK(launch_force_diag(H.bf.as<double>(), H.brho.as<double>(), nullptr, N, fact, scratch, scratch + FORCE_DIAG_SCRATCH, s), "force diagnostics launch");
  HCHK(hipMemcpyAsync(diag2, scratch + FORCE_DIAG_SCRATCH, sizeof(double) * 2, hipMemcpyDeviceToHost, s), "D2H diag");
  HCHK(hipStreamSynchronize(s), "sync");
  H.res_pois_host_stale = true;
  return 0;
}

// phi, f(1:ncell,1:3) and rho of the resident level back into the host arrays (backup_poisson)
int ramses_amd_resident_sync_poisson_f90(double *phi, double *f, double *rho) {
  HostCtx &H = g_host;
  if (!H.res_valid || !H.res_pois_host_stale) return 0;
  if (!phi || !f || !rho) return fail(RAMSES_AMD_EINVAL, "NULL argument");
  const int n = 1 << H.res_level;
  const long N = (long)n * n * n;
  const long ncell = H.res_ncell;
  hipStream_t s = nullptr;
  HCHK(H.cellvec1.ensure(sizeof(double) * 3 * ncell), "hipMalloc");
  PackArgs A;
  A.igrid = H.igrid.as<int>(); A.octorg = H.octorg.as<long>();
  A.ngrid = H.res_ngrid; A.n = n;
  A.ncoarse = H.res_ncoarse; A.ngridmax = H.res_ngridmax; A.ncell = ncell; A.pitch_var = N;
  struct { bool ok; double *host; double *brick; int nvar; } col[3] = {
      {H.res_phi_valid, phi, H.bphi.as<double>(), 1}, {H.res_grav_valid, f, H.bf.as<double>(), 3}, {H.res_rho_valid, rho, H.brho.as<double>(), 1}};
  for (auto &c : col) {
    if (!c.ok) continue;
    // cells of other levels keep their host values: scatter into a device copy of the host vector
    HCHK(hipMemcpyAsync(H.cellvec1.p, c.host, sizeof(double) * c.nvar * ncell, hipMemcpyHostToDevice, s), "H2D");
    A.nvar = c.nvar; A.brick = c.brick; A.cellvec = H.cellvec1.as<double>();
    HCHK(launch_oct_copy(A, false, s), "scatter launch");
    HCHK(hipMemcpyAsync(c.host, H.cellvec1.p, sizeof(double) * c.nvar * ncell, hipMemcpyDeviceToHost, s), "D2H");
    HCHK(hipStreamSynchronize(s), "sync");
  }
  H.res_pois_host_stale = false;
  return 0;
}

// ---------------------------------------------------------------------------
// Page-lock a host array of the caller for the staged paths (the Fortran module
// arrays are allocated once with fixed ngridmax and never reallocated, so their
// addresses are stable for the run): H2D/D2H of pinned memory runs at DMA speed
// instead of through the pageable bounce buffers.  Not fatal if the driver
// refuses (the copies then take the pageable path).  Opt-in (RAMSES_AMD_PIN=1):
// at the sizes measured so far (128^3 uniform, 570 k-cell AMR run) the staged
// calls gain 5-10 % and the one-time registration costs ~0.2 s.
// ---------------------------------------------------------------------------
int ramses_amd_host_register(void *p, int64_t bytes) {
  struct Range { char *lo, *hi; };
  static Range done[64];
  static int ndone = 0;
  static int enabled = -1;
  if (enabled < 0) {
    const char *e = getenv("RAMSES_AMD_PIN");
    enabled = e && e[0] == '1';
  }
  if (!enabled || !p || bytes <= 0) return 0;
  char *lo = static_cast<char *>(p), *hi = lo + bytes;
  for (int i = 0; i < ndone; i++)
    if (lo >= done[i].lo && hi <= done[i].hi) return 0;
  if (ndone >= 64) return 0;
  hipError_t e = hipHostRegister(p, (size_t)bytes, hipHostRegisterDefault);
  if (e != hipSuccess) (void)hipGetLastError();   // pageable copies still work
  done[ndone].lo = lo; done[ndone].hi = hi; ndone++;   // (also remembers refusals: asked once)
  return 0;
}

// forget the resident level (the host array was modified behind our back)
int ramses_amd_resident_invalidate(void) {
  HostCtx &H = g_host;
  if (H.res_valid && H.res_host_stale) return fail(RAMSES_AMD_EINVAL, "invalidate: the host array is stale; sync first");
  H.res_valid = false;
  return 0;
}
#undef HCHK

// ---------------------------------------------------------------------------
// Multigrid on AMR levels.  The reference's own driver (multigrid_fine and
// recursive_multigrid_coarse, poisson/multigrid_fine_commons.f90:25-390) and its
// per-solve setup (initial guess, masks, build_parent_comms_mg, scan flags) stay
// the reference's host code; the compute routines it calls are shadowed by the
// patch directory and run here.  begin() registers the tree and the fine level,
// add_level() the multigrid levels below it; the state of all levels then stays
// on the device until end() writes phi back (host resets of u(:,1:2) between the
// routines are folded into the restriction, which zeroes both).
// RAMSES_AMD_MG_SYNC=1: every routine reloads its inputs from the host arrays and
// writes its outputs back (debugging aid: any routine can then be switched to the
// reference individually).
// ---------------------------------------------------------------------------
extern "C++" {
namespace {
// a multigrid level = the rank's own octs (block 0) followed, under MPI, by the reception octs of the other ranks
// (active_mg(icpu,l) for icpu /= myid): one layout of ngrid = sum of the blocks' octs, of which the first nact are updated
struct MgAmrBlock {
  int ngrid, off;             // octs, position of the first one in the level's layout
  double *h_u;                // the block's host array u(1:ngrid*8, 1:4)
  const int *h_f;             // f(1:ngrid*8, 1)
};
// the virtual boundaries of one level of the solve (several ranks): emission = positions in the rank's own part of the
// layout, per peer; reception = the peer's block of the layout (rc_off octs in, rc_n octs long)
struct MgAmrComm {
  bool set = false;
  int ncpu = 0;
  std::vector<int> em_first, rc_off, rc_n;
  DevBuf em_pos;
};
struct MgAmrDev {
  int level = 0, ngrid = 0, nact = 0, filled = 0;
  DevBuf igrid, u1, u2, u3, u4, scan;
  MgAmrComm comm;
  std::vector<MgAmrBlock> blocks;   // (coarse levels; host arrays only valid during the solve)
  MgAmrLevel view() {
    MgAmrLevel L;
    L.ngrid = ngrid; L.nact = nact; L.igrid = igrid.as<int>();
    L.u1 = u1.as<double>(); L.u2 = u2.as<double>(); L.u3 = u3.as<double>(); L.u4 = u4.as<double>();
    L.scan = scan.as<int>();
    return L;
  }
};
struct MgAmrCtx {
  bool open = false, sync = false;
  int ilevel = 0;
  long ncoarse = 0, ngridmax = 0, ncell = 0;
  DevBuf son, nbor, father, lookup, vec, ivec, partial, norm;
  // halo exchanges of the solve: device message buffers, pinned host twins (host-MPI transport), the open exchange
  DevBuf sendbuf, recvbuf, tmpidx;
  void *h_send = nullptr, *h_recv = nullptr;
  size_t h_send_cap = 0, h_recv_cap = 0;
  std::vector<int64_t> send_off, recv_off;
  int halo_level = 0, halo_comp = 0, halo_dir = -1;
  // what crossed PCIe: [0] bytes of level arrays moved by the routines AFTER the first one of the solve uploaded them,
  // [1] number of such copies, [2] bytes of halo messages (host-MPI transport), [3] halo exchanges
  long long stats[4] = {0, 0, 0, 0};
  bool uploaded = false;
  MgAmrDev lev[32];
  // host arrays of the fine level
  double *h_phi = nullptr, *h_f = nullptr;   // f(1:ncell,1:3)
  const int *h_flag2 = nullptr;
  MgAmrTree tree() {
    MgAmrTree T;
    T.son = son.as<int>(); T.nbor = nbor.as<int>(); T.father = father.as<int>(); T.lookup = lookup.as<int>();
    T.ncoarse = ncoarse; T.ngridmax = ngridmax;
    return T;
  }
};
MgAmrCtx g_mg;
bool g_mg_force_sync = false;     // several MPI ranks: every routine exchanges its arrays with the host
}  // namespace
}  // extern "C++"

#define HCHK(call, what) do { hipError_t e_ = (call); if (e_ != hipSuccess) return hipfail(e_, what); } while (0)

// (re)load the fine level from the host arrays: phi -> u1, f(:,2) -> u2, f(:,3) -> u4, flag2 -> scan
static void mgamr_count(size_t bytes) {
  MgAmrCtx &M = g_mg;
  if (M.uploaded) { M.stats[0] += (long long)bytes; M.stats[1] += 1; }
}
static int mgamr_load_fine(bool with_residual) {
  MgAmrCtx &M = g_mg;
  MgAmrDev &D = M.lev[M.ilevel];
  mgamr_count(sizeof(double) * (size_t)M.ncell * (with_residual ? 4 : 3) + sizeof(int) * (size_t)M.ncell);
  hipStream_t s = nullptr;
  const long ncell = M.ncell;
  HCHK(M.vec.ensure(sizeof(double) * ncell), "hipMalloc");
  HCHK(M.ivec.ensure(sizeof(int) * ncell), "hipMalloc");
  struct { const double *src; double *dst; } cols[4] = {{M.h_phi, D.u1.as<double>()}, {M.h_f + ncell, D.u2.as<double>()},
                                                       {M.h_f + 2 * ncell, D.u4.as<double>()}, {M.h_f, D.u3.as<double>()}};
  for (int k = 0; k < (with_residual ? 4 : 3); k++) {
    HCHK(hipMemcpyAsync(M.vec.p, cols[k].src, sizeof(double) * ncell, hipMemcpyHostToDevice, s), "H2D");
    HCHK(mgamr_launch_gather(M.vec.as<double>(), cols[k].dst, D.igrid.as<int>(), D.ngrid, M.ncoarse, M.ngridmax, s), "gather");
  }
  HCHK(hipMemcpyAsync(M.ivec.p, M.h_flag2, sizeof(int) * ncell, hipMemcpyHostToDevice, s), "H2D flag2");
  HCHK(mgamr_launch_gather_scan(M.ivec.as<int>(), D.scan.as<int>(), D.igrid.as<int>(), D.ngrid, M.ncoarse, M.ngridmax, s), "gather");
  return 0;
}
// write one array of the fine level back into its host cell vector (other cells untouched)
static int mgamr_store_fine(double *h_vec, const double *d_col, bool whole_layout = false) {
  MgAmrCtx &M = g_mg;
  MgAmrDev &D = M.lev[M.ilevel];
  hipStream_t s = nullptr;
  if (M.open) mgamr_count(2 * sizeof(double) * (size_t)M.ncell);
  HCHK(hipMemcpyAsync(M.vec.p, h_vec, sizeof(double) * M.ncell, hipMemcpyHostToDevice, s), "H2D");
  HCHK(mgamr_launch_scatter(M.vec.as<double>(), d_col, D.igrid.as<int>(), whole_layout ? D.ngrid : D.nact, D.ngrid, M.ncoarse, M.ngridmax, s), "scatter");
  HCHK(hipMemcpyAsync(h_vec, M.vec.p, sizeof(double) * M.ncell, hipMemcpyDeviceToHost, s), "D2H");
  HCHK(hipStreamSynchronize(s), "sync");
  return 0;
}
// one component (8*ngrid_b values, octant-major) of every block between the blocks' host arrays and the level's layout
static int mgamr_copy_comp(MgAmrDev &D, DevBuf &dev, int k, bool to_device, bool mine_only) {   // k = 1..4
  hipStream_t s = nullptr;
  for (size_t b = 0; b < D.blocks.size(); b++) {
    const MgAmrBlock &B = D.blocks[b];
    if (B.ngrid == 0 || (mine_only && b > 0)) continue;
    double *host = B.h_u + (size_t)(k - 1) * 8 * B.ngrid;
    double *devp = dev.as<double>() + B.off;
    mgamr_count(sizeof(double) * 8 * (size_t)B.ngrid);
    if (to_device) HCHK(hipMemcpy2DAsync(devp, sizeof(double) * D.ngrid, host, sizeof(double) * B.ngrid, sizeof(double) * B.ngrid, 8, hipMemcpyHostToDevice, s), "H2D level");
    else HCHK(hipMemcpy2DAsync(host, sizeof(double) * B.ngrid, devp, sizeof(double) * D.ngrid, sizeof(double) * B.ngrid, 8, hipMemcpyDeviceToHost, s), "D2H level");
  }
  return 0;
}
static int mgamr_load_coarse(MgAmrDev &D, bool all) {
  if (D.ngrid == 0) return 0;
  if (all) {
    if (int rc = mgamr_copy_comp(D, D.u1, 1, true, false)) return rc;
    if (int rc = mgamr_copy_comp(D, D.u2, 2, true, false)) return rc;
    if (int rc = mgamr_copy_comp(D, D.u3, 3, true, false)) return rc;
  }
  return mgamr_copy_comp(D, D.u4, 4, true, false);
}
// k = 1..3; all_blocks: the reception blocks too (the restriction adds into cells other ranks own)
static int mgamr_store_coarse(MgAmrDev &D, int k, bool all_blocks = false) {
  if (D.ngrid == 0) return 0;
  DevBuf *b[3] = {&D.u1, &D.u2, &D.u3};
  if (int rc = mgamr_copy_comp(D, *b[k - 1], k, false, !all_blocks)) return rc;
  HCHK(hipStreamSynchronize(nullptr), "sync");
  return 0;
}
__global__ void mgamr_scan_bit_kernel(const int *f, int *scan, long n) {
  for (long c = (long)blockIdx.x * blockDim.x + threadIdx.x; c < n; c += (long)gridDim.x * blockDim.x) scan[c] = f[c] & 1;
}

int ramses_amd_mgamr_begin(int ilevel, int64_t ngridmax, int64_t ncoarse, const int *son, const int *nbor,
                           const int *father, const int *lookup_mg, const int *flag2, double *phi, double *f,
                           int ngrid, const int *igrid) {
  if (!son || !nbor || !father || !lookup_mg || !flag2 || !phi || !f || (!igrid && ngrid > 0)) return fail(RAMSES_AMD_EINVAL, "NULL argument");
  if (ilevel < 2 || ilevel > 30) return fail(RAMSES_AMD_EUNSUPPORTED, "AMR multigrid on the device needs 2 <= ilevel <= 30 (got %d)", ilevel);
  if (int rc = resident_release("multigrid_fine (AMR level)")) return rc;
  MgAmrCtx &M = g_mg;
  hipStream_t s = nullptr;
  const char *e = getenv("RAMSES_AMD_MG_SYNC");
  M.sync = (e && e[0] == '1') || g_mg_force_sync;
  M.open = true; M.ilevel = ilevel; M.ncoarse = ncoarse; M.ngridmax = ngridmax; M.ncell = ncoarse + 8 * ngridmax;
  M.uploaded = false; M.halo_level = 0; M.halo_dir = -1;
  for (int k = 0; k < 4; k++) M.stats[k] = 0;
  M.h_phi = phi; M.h_f = f; M.h_flag2 = flag2;
  for (int l = 0; l < 32; l++) { M.lev[l].ngrid = 0; M.lev[l].nact = 0; M.lev[l].filled = 0; M.lev[l].level = l; M.lev[l].blocks.clear(); M.lev[l].comm.set = false; }
  HCHK(M.son.ensure(sizeof(int) * M.ncell), "hipMalloc son");
  HCHK(M.nbor.ensure(sizeof(int) * 6 * ngridmax), "hipMalloc nbor");
  HCHK(M.father.ensure(sizeof(int) * ngridmax), "hipMalloc father");
  HCHK(M.lookup.ensure(sizeof(int) * ngridmax), "hipMalloc lookup");
  HCHK(M.partial.ensure(sizeof(double) * 1024), "hipMalloc partial");
  HCHK(M.norm.ensure(sizeof(double)), "hipMalloc norm");
  HCHK(hipMemcpyAsync(M.son.p, son, sizeof(int) * M.ncell, hipMemcpyHostToDevice, s), "H2D son");
  HCHK(hipMemcpyAsync(M.nbor.p, nbor, sizeof(int) * 6 * ngridmax, hipMemcpyHostToDevice, s), "H2D nbor");
  HCHK(hipMemcpyAsync(M.father.p, father, sizeof(int) * ngridmax, hipMemcpyHostToDevice, s), "H2D father");
  // (the oct -> position table is built here from the levels' lists: under MPI the reference's lookup_mg counts inside each
  //  rank's buffer, the device layout is the concatenation of the buffers)
  (void)lookup_mg;
  HCHK(hipMemsetAsync(M.lookup.p, 0, sizeof(int) * ngridmax, s), "memset lookup");
  MgAmrDev &D = M.lev[ilevel];
  D.ngrid = ngrid; D.nact = ngrid; D.blocks.clear();
  const size_t n = sizeof(double) * 8 * (size_t)(ngrid > 0 ? ngrid : 1);
  HCHK(D.igrid.ensure(sizeof(int) * (size_t)(ngrid > 0 ? ngrid : 1)), "hipMalloc");
  HCHK(D.u1.ensure(n), "hipMalloc"); HCHK(D.u2.ensure(n), "hipMalloc"); HCHK(D.u3.ensure(n), "hipMalloc"); HCHK(D.u4.ensure(n), "hipMalloc");
  HCHK(D.scan.ensure(sizeof(int) * 8 * (size_t)(ngrid > 0 ? ngrid : 1)), "hipMalloc");
  if (ngrid > 0) {
    HCHK(hipMemcpyAsync(D.igrid.p, igrid, sizeof(int) * ngrid, hipMemcpyHostToDevice, s), "H2D igrid");
    // octs of the fine level are found through the same lookup table (their lookup_mg entries are unused)
    HCHK(mgamr_launch_lookup(D.igrid.as<int>(), ngrid, M.lookup.as<int>(), s), "lookup");
  }
  if (int rc = mgamr_load_fine(false)) return rc;
  return 0;
}

// a multigrid level in blocks: level_begin(total octs), then level_block() per rank buffer with octs -- the calling rank's
// own first --, the last block completes the level (lists, masks, scan flags and the other arrays go to the device)
int ramses_amd_mgamr_level_begin(int level, int ngrid_total) {
  MgAmrCtx &M = g_mg;
  if (!M.open) return fail(RAMSES_AMD_EINVAL, "mgamr_level_begin outside begin/end");
  if (level < 1 || level >= M.ilevel) return fail(RAMSES_AMD_EINVAL, "multigrid level %d out of range", level);
  if (ngrid_total < 0) return fail(RAMSES_AMD_EINVAL, "bad oct count");
  MgAmrDev &D = M.lev[level];
  D.ngrid = ngrid_total; D.nact = 0; D.filled = 0; D.blocks.clear();
  const size_t nn = 8 * (size_t)(ngrid_total > 0 ? ngrid_total : 1);
  HCHK(D.igrid.ensure(sizeof(int) * (size_t)(ngrid_total > 0 ? ngrid_total : 1)), "hipMalloc");
  HCHK(D.u1.ensure(sizeof(double) * nn), "hipMalloc"); HCHK(D.u2.ensure(sizeof(double) * nn), "hipMalloc");
  HCHK(D.u3.ensure(sizeof(double) * nn), "hipMalloc"); HCHK(D.u4.ensure(sizeof(double) * nn), "hipMalloc");
  HCHK(D.scan.ensure(sizeof(int) * nn), "hipMalloc");
  return 0;
}
int ramses_amd_mgamr_level_block(int level, int ngrid, const int *igrid, double *u, const int *fscan) {
  MgAmrCtx &M = g_mg;
  if (!M.open) return fail(RAMSES_AMD_EINVAL, "mgamr_level_block outside begin/end");
  if (level < 1 || level >= M.ilevel) return fail(RAMSES_AMD_EINVAL, "multigrid level %d out of range", level);
  if (ngrid < 0 || (ngrid > 0 && (!igrid || !u || !fscan))) return fail(RAMSES_AMD_EINVAL, "bad block");
  hipStream_t s = nullptr;
  MgAmrDev &D = M.lev[level];
  if (D.filled + ngrid > D.ngrid) return fail(RAMSES_AMD_EINVAL, "level %d: blocks exceed the announced %d octs", level, D.ngrid);
  MgAmrBlock B = {ngrid, D.filled, u, fscan};
  if (D.blocks.empty()) D.nact = ngrid;          // the first block is the caller's own
  D.blocks.push_back(B);
  if (ngrid > 0) {
    HCHK(hipMemcpyAsync(D.igrid.as<int>() + B.off, igrid, sizeof(int) * ngrid, hipMemcpyHostToDevice, s), "H2D igrid");
    const size_t nn = 8 * (size_t)D.ngrid;
    HCHK(M.ivec.ensure(sizeof(int) * nn > sizeof(int) * M.ncell ? sizeof(int) * nn : sizeof(int) * M.ncell), "hipMalloc");
    HCHK(hipMemcpy2DAsync(M.ivec.as<int>() + B.off, sizeof(int) * D.ngrid, fscan, sizeof(int) * ngrid, sizeof(int) * ngrid, 8, hipMemcpyHostToDevice, s), "H2D scan");
  }
  D.filled += ngrid;
  if (D.filled < D.ngrid) { HCHK(hipStreamSynchronize(s), "sync"); return 0; }
  // complete: positions, scan bits, arrays
  if (D.ngrid > 0) {
    HCHK(mgamr_launch_lookup(D.igrid.as<int>(), D.ngrid, M.lookup.as<int>(), s), "lookup");
    hipLaunchKernelGGL(mgamr_scan_bit_kernel, dim3(64), dim3(256), 0, s, M.ivec.as<int>(), D.scan.as<int>(), (long)(8 * (size_t)D.ngrid));
    HCHK(hipGetLastError(), "scan launch");
    if (int rc = mgamr_load_coarse(D, true)) return rc;
  }
  HCHK(hipStreamSynchronize(s), "sync");   // host buffers of the caller may be temporaries
  return 0;
}
int ramses_amd_mgamr_force_sync(int on) { g_mg_force_sync = on != 0; return 0; }
// single rank: the level is one block
int ramses_amd_mgamr_add_level(int level, int ngrid, const int *igrid, double *u, const int *fscan) {
  if (ngrid > 0 && (!igrid || !u || !fscan)) return fail(RAMSES_AMD_EINVAL, "NULL argument");
  if (int rc = ramses_amd_mgamr_level_begin(level, ngrid)) return rc;
  return ramses_amd_mgamr_level_block(level, ngrid, igrid, u, fscan);
}
// the fine level's list passed to begin() holds nact active octs followed by reception octs (MPI): only the former are updated
int ramses_amd_mgamr_fine_active(int nact) {
  MgAmrCtx &M = g_mg;
  if (!M.open) return fail(RAMSES_AMD_EINVAL, "mgamr_fine_active outside begin/end");
  MgAmrDev &D = M.lev[M.ilevel];
  if (nact < 0 || nact > D.ngrid) return fail(RAMSES_AMD_EINVAL, "bad active count %d of %d", nact, D.ngrid);
  D.nact = nact;
  return 0;
}

static int mgamr_level(int level, MgAmrDev **out) {
  MgAmrCtx &M = g_mg;
  if (!M.open) return fail(RAMSES_AMD_EINVAL, "AMR multigrid routine called outside begin/end");
  if (level < 1 || level > M.ilevel) return fail(RAMSES_AMD_EINVAL, "level %d is not part of the solve", level);
  *out = &M.lev[level];
  M.uploaded = true;         // a compute routine runs: the levels are on the device, what moves from here on is counted
  return 0;
}
static int mgamr_sync_in(int level, bool with_residual) {
  MgAmrCtx &M = g_mg;
  if (!M.sync) return 0;
  if (level == M.ilevel) return mgamr_load_fine(with_residual);
  return mgamr_load_coarse(M.lev[level], true);
}

int ramses_amd_mgamr_gauss_seidel(int level, int redstep, int safe) {
  MgAmrDev *D;
  if (int rc = mgamr_level(level, &D)) return rc;
  if (int rc = mgamr_sync_in(level, false)) return rc;
  const double dx = std::ldexp(1.0, -level);
  HCHK(mgamr_launch_gs(D->view(), g_mg.tree(), redstep ? 0 : 1, safe, dx * dx, nullptr), "gs launch");
  if (g_mg.sync) return level == g_mg.ilevel ? mgamr_store_fine(g_mg.h_phi, D->u1.as<double>()) : mgamr_store_coarse(*D, 1);
  return 0;
}
int ramses_amd_mgamr_residual(int level) {
  MgAmrDev *D;
  if (int rc = mgamr_level(level, &D)) return rc;
  if (int rc = mgamr_sync_in(level, false)) return rc;
  const double dx = std::ldexp(1.0, -level);
  HCHK(mgamr_launch_residual(D->view(), g_mg.tree(), 1.0 / (dx * dx), nullptr), "residual launch");
  if (g_mg.sync) return level == g_mg.ilevel ? mgamr_store_fine(g_mg.h_f, D->u3.as<double>()) : mgamr_store_coarse(*D, 3);
  return 0;
}
int ramses_amd_mgamr_norm2(int level, double *norm2) {
  MgAmrDev *D;
  if (!norm2) return fail(RAMSES_AMD_EINVAL, "NULL argument");
  if (int rc = mgamr_level(level, &D)) return rc;
  if (int rc = mgamr_sync_in(level, true)) return rc;
  const double dx = std::ldexp(1.0, -level);
  HCHK(mgamr_launch_norm(D->view(), dx * dx * dx, g_mg.partial.as<double>(), g_mg.norm.as<double>(), nullptr), "norm launch");
  HCHK(hipMemcpy(norm2, g_mg.norm.p, sizeof(double), hipMemcpyDeviceToHost), "D2H norm");
  return 0;
}
// restriction of the residual of `finelevel` into the rhs of finelevel-1; also zeroes that level's correction
int ramses_amd_mgamr_restrict(int finelevel) {
  MgAmrDev *F, *C;
  if (int rc = mgamr_level(finelevel, &F)) return rc;
  if (int rc = mgamr_level(finelevel - 1, &C)) return rc;
  if (int rc = mgamr_sync_in(finelevel, true)) return rc;
  if (g_mg.sync) if (int rc = mgamr_load_coarse(*C, true)) return rc;
  HCHK(mgamr_launch_restrict(F->view(), C->view(), g_mg.tree(), nullptr), "restrict launch");
  if (g_mg.sync) {
    if (int rc = mgamr_store_coarse(*C, 2, true)) return rc;
    // the correction is reset by the reference's driver itself; do not touch the host copy
  }
  return 0;
}
int ramses_amd_mgamr_interpolate(int finelevel) {
  MgAmrDev *F, *C;
  if (int rc = mgamr_level(finelevel, &F)) return rc;
  if (int rc = mgamr_level(finelevel - 1, &C)) return rc;
  if (int rc = mgamr_sync_in(finelevel, false)) return rc;
  if (g_mg.sync) if (int rc = mgamr_load_coarse(*C, true)) return rc;
  HCHK(mgamr_launch_interp(F->view(), C->view(), g_mg.tree(), nullptr), "interp launch");
  if (g_mg.sync) return finelevel == g_mg.ilevel ? mgamr_store_fine(g_mg.h_phi, F->u1.as<double>()) : mgamr_store_coarse(*F, 1);
  return 0;
}
// end of the solve: phi of the fine level goes back to the host array
int ramses_amd_mgamr_end(void) {
  MgAmrCtx &M = g_mg;
  if (!M.open) return 0;
  int rc = 0;
  M.uploaded = false;        // (the solution's way home is part of the solve, not of a routine)
  // several ranks: the reception octs' phi is current on the device as well (the last make_virtual_fine_dp ran there)
  if (!M.sync) rc = mgamr_store_fine(M.h_phi, M.lev[M.ilevel].u1.as<double>(), true);
  M.open = false;
  const char *e = getenv("RAMSES_AMD_MG_STATS");
  if (e && e[0] == '1') {
    printf(" ramses_amd: multigrid level %d: level arrays across PCIe after the upload: %lld bytes in %lld copies; halo: %lld bytes in %lld exchanges\n",
           M.ilevel, M.stats[0], M.stats[1], M.stats[2], M.stats[3]);
    fflush(stdout);
  }
  return rc;
}
int ramses_amd_mgamr_stats(int64_t *out4) {
  if (!out4) return fail(RAMSES_AMD_EINVAL, "NULL argument");
  for (int k = 0; k < 4; k++) out4[k] = (int64_t)g_mg.stats[k];
  return 0;
}

// ---- virtual boundaries of the levels of the solve (several ranks), on the device -----------------------------------
// make_virtual_fine_dp(phi / f(:,1)) of the solved level, make_virtual_mg_dp / make_reverse_mg_dp of the multigrid levels
// (poisson/multigrid_fine_commons.f90:1172-1290,1378-1475): the level's layout is the rank's own octs followed by every
// peer's reception block, so a forward exchange gathers the emission cells (positions in the own part) into one message per
// peer and drops what arrives into the peer's block; a reverse exchange sends the blocks and ADDS what arrives to the
// emission cells, peer by peer in icpu order like the reference (:1443-1457; floating-point addition is not associative).
// Message layout = the reference's: u(i + (ind-1)*n).
__global__ void mgamr_pos_from_octs_kernel(const int *octs, int n, const int *lookup, int *pos, int *bad) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int j = lookup[octs[i] - 1];
  if (j <= 0) atomicAdd(bad, 1);
  pos[i] = j - 1;
}
__global__ void mgamr_pos_shift_kernel(int *pos, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) pos[i] -= 1;
}
// buf[ind*n + i] <-> comp[ind*ngrid + pos[i]]  (pos == nullptr: the block of n octs starting at off)
extern "C++" {
template <int MODE>   // 0 gather into buf, 1 scatter from buf, 2 add buf
__global__ void mgamr_halo_kernel(double *__restrict__ comp, int ngrid, const int *__restrict__ pos, int off, int n, double *__restrict__ buf) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= 8L * n) return;
  const int i = (int)(t % n), ind = (int)(t / n);
  const long c = (long)ind * ngrid + (pos ? pos[i] : off + i);
  if (MODE == 0) buf[t] = comp[c];
  else if (MODE == 1) comp[c] = buf[t];
  else comp[c] = comp[c] + buf[t];
}
}  // extern "C++"

int ramses_amd_mgamr_comm_set(int level, int ncpu, int myid, const int *em_n, const int *em_list, int list_is_octs, const int *rc_n) {
  MgAmrDev *D;
  if (int rc = mgamr_level(level, &D)) return rc;
  g_mg.uploaded = false;     // (setup, not a routine)
  if (ncpu < 1 || myid < 1 || myid > ncpu || !em_n || !rc_n) return fail(RAMSES_AMD_EINVAL, "mgamr_comm_set: bad argument");
  MgAmrComm &Cm = D->comm;
  Cm.set = false; Cm.ncpu = ncpu;
  Cm.em_first.assign((size_t)ncpu + 1, 0); Cm.rc_off.assign((size_t)ncpu, 0); Cm.rc_n.assign((size_t)ncpu, 0);
  int off = D->nact;
  for (int c = 0; c < ncpu; c++) {
    if (em_n[c] < 0 || rc_n[c] < 0) return fail(RAMSES_AMD_EINVAL, "mgamr_comm_set: negative list length");
    Cm.em_first[c + 1] = Cm.em_first[c] + em_n[c];
    const int n = c == myid - 1 ? 0 : rc_n[c];
    Cm.rc_off[c] = off; Cm.rc_n[c] = n;
    off += n;
  }
  if (off != D->ngrid) return fail(RAMSES_AMD_EINVAL, "mgamr_comm_set(level %d): own %d + reception octs = %d, the layout has %d", level, D->nact, off, D->ngrid);
  const int nem = Cm.em_first[ncpu];
  if (nem > 0 && !em_list) return fail(RAMSES_AMD_EINVAL, "mgamr_comm_set: NULL list");
  HCHK(Cm.em_pos.ensure(sizeof(int) * (size_t)(nem > 0 ? nem : 1)), "hipMalloc");
  if (nem > 0) {
    hipStream_t s = nullptr;
    if (list_is_octs) {
      HCHK(g_mg.tmpidx.ensure(sizeof(int) * ((size_t)nem + 1)), "hipMalloc");
      int *d_octs = g_mg.tmpidx.as<int>(), *d_bad = d_octs + nem;
      HCHK(hipMemcpyAsync(d_octs, em_list, sizeof(int) * (size_t)nem, hipMemcpyHostToDevice, s), "H2D emission list");
      HCHK(hipMemsetAsync(d_bad, 0, sizeof(int), s), "memset");
      hipLaunchKernelGGL(mgamr_pos_from_octs_kernel, dim3((nem + 255) / 256), dim3(256), 0, s, d_octs, nem, g_mg.lookup.as<int>(), Cm.em_pos.as<int>(), d_bad);
      int bad = 0;
      HCHK(hipMemcpy(&bad, d_bad, sizeof(int), hipMemcpyDeviceToHost), "D2H");
      if (bad) return fail(RAMSES_AMD_EINVAL, "mgamr_comm_set(level %d): %d emission octs are not octs of the level", level, bad);
    } else {
      HCHK(hipMemcpyAsync(Cm.em_pos.p, em_list, sizeof(int) * (size_t)nem, hipMemcpyHostToDevice, s), "H2D emission list");
      hipLaunchKernelGGL(mgamr_pos_shift_kernel, dim3((nem + 255) / 256), dim3(256), 0, s, Cm.em_pos.as<int>(), nem);
      HCHK(hipStreamSynchronize(s), "sync");
      for (int k = 0; k < nem; k++) if (em_list[k] < 1 || em_list[k] > D->nact) return fail(RAMSES_AMD_EINVAL, "mgamr_comm_set(level %d): emission position %d outside 1..%d", level, em_list[k], D->nact);
    }
  }
  Cm.set = true;
  return 0;
}

namespace {
int mgamr_halo_args(int level, int comp, int dir, MgAmrDev **D, double **vec) {
  if (int rc = mgamr_level(level, D)) return rc;
  if (!(*D)->comm.set) return fail(RAMSES_AMD_EINVAL, "level %d: no communicators on the device (ramses_amd_mgamr_comm_set)", level);
  if (comp < 1 || comp > 4 || dir < 0 || dir > 1) return fail(RAMSES_AMD_EINVAL, "mgamr halo: bad component %d / direction %d", comp, dir);
  DevBuf *b[4] = {&(*D)->u1, &(*D)->u2, &(*D)->u3, &(*D)->u4};
  *vec = b[comp - 1]->as<double>();
  return 0;
}
// messages of every peer into sendbuf; offsets in doubles
int mgamr_halo_pack(MgAmrDev &D, double *vec, int dir) {
  MgAmrCtx &M = g_mg;
  MgAmrComm &Cm = D.comm;
  M.send_off.assign((size_t)Cm.ncpu + 1, 0); M.recv_off.assign((size_t)Cm.ncpu + 1, 0);
  for (int c = 0; c < Cm.ncpu; c++) {
    const int64_t ne = 8 * (int64_t)(Cm.em_first[c + 1] - Cm.em_first[c]), nr = 8 * (int64_t)Cm.rc_n[c];
    M.send_off[c + 1] = M.send_off[c] + (dir == 0 ? ne : nr);
    M.recv_off[c + 1] = M.recv_off[c] + (dir == 0 ? nr : ne);
  }
  const size_t ns = (size_t)M.send_off[Cm.ncpu], nr = (size_t)M.recv_off[Cm.ncpu];
  HCHK(M.sendbuf.ensure(sizeof(double) * (ns > 0 ? ns : 1)), "hipMalloc sendbuf");
  HCHK(M.recvbuf.ensure(sizeof(double) * (nr > 0 ? nr : 1)), "hipMalloc recvbuf");
  for (int c = 0; c < Cm.ncpu; c++) {
    const int n = (int)((M.send_off[c + 1] - M.send_off[c]) / 8);
    if (n <= 0) continue;
    double *buf = M.sendbuf.as<double>() + M.send_off[c];
    const dim3 g((unsigned)((8L * n + 255) / 256)), b(256);
    if (dir == 0) hipLaunchKernelGGL(mgamr_halo_kernel<0>, g, b, 0, nullptr, vec, D.ngrid, Cm.em_pos.as<int>() + Cm.em_first[c], 0, n, buf);
    else hipLaunchKernelGGL(mgamr_halo_kernel<0>, g, b, 0, nullptr, vec, D.ngrid, (const int *)nullptr, Cm.rc_off[c], n, buf);
  }
  HCHK(hipGetLastError(), "mgamr halo pack launch");
  return 0;
}
int mgamr_halo_unpack(MgAmrDev &D, double *vec, int dir) {
  MgAmrCtx &M = g_mg;
  MgAmrComm &Cm = D.comm;
  for (int c = 0; c < Cm.ncpu; c++) {      // icpu order: the reverse exchange adds peer by peer
    const int n = (int)((M.recv_off[c + 1] - M.recv_off[c]) / 8);
    if (n <= 0) continue;
    double *buf = M.recvbuf.as<double>() + M.recv_off[c];
    const dim3 g((unsigned)((8L * n + 255) / 256)), b(256);
    if (dir == 0) hipLaunchKernelGGL(mgamr_halo_kernel<1>, g, b, 0, nullptr, vec, D.ngrid, (const int *)nullptr, Cm.rc_off[c], n, buf);
    else hipLaunchKernelGGL(mgamr_halo_kernel<2>, g, b, 0, nullptr, vec, D.ngrid, Cm.em_pos.as<int>() + Cm.em_first[c], 0, n, buf);
  }
  HCHK(hipGetLastError(), "mgamr halo unpack launch");
  return 0;
}
int pin_ensure(void *&p, size_t &cap, size_t bytes) {
  if (bytes <= cap) return 0;
  if (p) { hipHostFree(p); p = nullptr; cap = 0; }
  const size_t want = bytes + bytes / 2;
  HCHK(hipHostMalloc(&p, want, hipHostMallocDefault), "hipHostMalloc");
  cap = want;
  return 0;
}
}  // namespace

// host-MPI transport (several ranks on one GPU, or no RCCL): stage_out packs on the device and hands pinned host buffers over
// -- the message for peer icpu at h_send + send_off[icpu-1], likewise h_recv / recv_off -- stage_in applies what arrived
int ramses_amd_mgamr_halo_stage_out(int level, int comp, int dir, int ncpu, int64_t *h_send_addr, int64_t *h_recv_addr, int64_t *send_off,
                                    int64_t *recv_off) {
  MgAmrCtx &M = g_mg;
  MgAmrDev *D;
  double *vec;
  if (!h_send_addr || !h_recv_addr || !send_off || !recv_off) return fail(RAMSES_AMD_EINVAL, "NULL argument");
  if (int rc = mgamr_halo_args(level, comp, dir, &D, &vec)) return rc;
  if (ncpu != D->comm.ncpu) return fail(RAMSES_AMD_EINVAL, "ncpu mismatch");
  if (int rc = mgamr_halo_pack(*D, vec, dir)) return rc;
  const size_t ns = (size_t)M.send_off[ncpu], nr = (size_t)M.recv_off[ncpu];
  if (int rc = pin_ensure(M.h_send, M.h_send_cap, sizeof(double) * (ns > 0 ? ns : 1))) return rc;
  if (int rc = pin_ensure(M.h_recv, M.h_recv_cap, sizeof(double) * (nr > 0 ? nr : 1))) return rc;
  if (ns > 0) HCHK(hipMemcpyAsync(M.h_send, M.sendbuf.p, sizeof(double) * ns, hipMemcpyDeviceToHost, nullptr), "D2H halo");
  HCHK(hipStreamSynchronize(nullptr), "sync");
  *h_send_addr = (int64_t)(intptr_t)M.h_send; *h_recv_addr = (int64_t)(intptr_t)M.h_recv;
  for (int c = 0; c <= ncpu; c++) { send_off[c] = M.send_off[c]; recv_off[c] = M.recv_off[c]; }
  M.halo_level = level; M.halo_comp = comp; M.halo_dir = dir;
  M.stats[2] += (long long)(sizeof(double) * (ns + nr)); M.stats[3] += 1;
  return 0;
}
int ramses_amd_mgamr_halo_stage_in(int level, int comp, int dir) {
  MgAmrCtx &M = g_mg;
  MgAmrDev *D;
  double *vec;
  if (int rc = mgamr_halo_args(level, comp, dir, &D, &vec)) return rc;
  if (M.halo_level != level || M.halo_comp != comp || M.halo_dir != dir)
    return fail(RAMSES_AMD_EINVAL, "mgamr_halo_stage_in(level %d, component %d, dir %d) does not close the exchange stage_out opened (%d, %d, %d)",
                level, comp, dir, M.halo_level, M.halo_comp, M.halo_dir);
  M.halo_level = 0; M.halo_dir = -1;
  const size_t nr = (size_t)M.recv_off[D->comm.ncpu];
  if (nr > 0) HCHK(hipMemcpyAsync(M.recvbuf.p, M.h_recv, sizeof(double) * nr, hipMemcpyHostToDevice, nullptr), "H2D halo");
  return mgamr_halo_unpack(*D, vec, dir);
}
// the same exchange over RCCL (every rank on its own GPU): one grouped send/recv, nothing crosses PCIe
extern "C" int ramses_amd_rccl_exchange(int npeer, const int *peer, const double *d_send, const int64_t *send_off, const int64_t *send_cnt,
                                        double *d_recv, const int64_t *recv_off, const int64_t *recv_cnt, void *stream);
int ramses_amd_mgamr_halo_rccl(int level, int comp, int dir) {
  MgAmrCtx &M = g_mg;
  MgAmrDev *D;
  double *vec;
  if (int rc = mgamr_halo_args(level, comp, dir, &D, &vec)) return rc;
  if (int rc = mgamr_halo_pack(*D, vec, dir)) return rc;
  std::vector<int> peer;
  std::vector<int64_t> so, sc, ro, rcn;
  for (int c = 0; c < D->comm.ncpu; c++) {
    const int64_t ns = M.send_off[c + 1] - M.send_off[c], nr = M.recv_off[c + 1] - M.recv_off[c];
    if (ns == 0 && nr == 0) continue;
    peer.push_back(c); so.push_back(M.send_off[c]); sc.push_back(ns); ro.push_back(M.recv_off[c]); rcn.push_back(nr);
  }
  if (int rc = ramses_amd_rccl_exchange((int)peer.size(), peer.data(), M.sendbuf.as<double>(), so.data(), sc.data(), M.recvbuf.as<double>(),
                                        ro.data(), rcn.data(), nullptr)) return rc;
  M.stats[3] += 1;
  return mgamr_halo_unpack(*D, vec, dir);
}

// ---------------------------------------------------------------------------
// Conjugate-gradient Poisson solver on one AMR level (phi_fine_cg,
// poisson/phi_fine_cg.f90:88-187; kernels in cg_amr.hip).  The caller has run the
// reference's pre-loop steps (initial guess, boundaries, cmp_residual_cg): phi and
// f(:,1) = f(:,2) = r hold the state the loop starts from.  The loop is pipelined:
// iteration k+1 is queued while r2 of iteration k travels to the host, which needs
// it only to decide about iteration k+2 (the reference tests the error of the
// previous iteration).
// ---------------------------------------------------------------------------
extern "C++" {
namespace {
struct CgCtx {
  DevBuf son, nbor, igrid, nb, x, r, p, z, rho, scal, partial, prod, scan;
  double *pin = nullptr;      // pinned, device-visible: r2 of each iteration (ring of 4, written by the kernels), rhs norm
  double *pin_dev = nullptr;
  hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
};
CgCtx g_cg;
// How the solver's dot products are summed.  Argument: 1 the reference's order by the parallel parity scan (bit-identical;
// the default), 2 the same order by a one-lane chain (slow: the scan's check), 0 a fixed parallel tree (fastest; equal to
// rounding only), < 0 as RAMSES_AMD_CG_ORDERED says ("0", "1", "chain"; unset: 1).
int cg_sum_mode(int ordered) {
  if (ordered >= 0) return ordered > 2 ? 1 : ordered;
  const char *e = getenv("RAMSES_AMD_CG_ORDERED");
  if (!e || !e[0]) return 1;
  if (e[0] == '0') return 0;
  if (e[0] == 'c') return 2;
  return 1;
}
}  // namespace
}  // extern "C++"

int ramses_amd_cg_solve_host(int ilevel, int ngrid, const int *igrid, const int *son, const int *nbor,
                             int64_t ngridmax, int64_t ncoarse, double *phi, double *f, const double *rho_or_null,
                             double rho_tot, double fact, double ncell_level, double epsilon, int itermax,
                             int ordered, int *iter_out, double *err_out) {
  if (!igrid || !son || !nbor || !phi || !f || !iter_out || !err_out) return fail(RAMSES_AMD_EINVAL, "NULL argument");
  if (ngrid <= 0 || ngridmax < ngrid || ncoarse < 1) return fail(RAMSES_AMD_EINVAL, "bad level sizes (ngrid=%d)", ngrid);
  if (ilevel < 1 || ilevel > 30) return fail(RAMSES_AMD_EINVAL, "bad level %d", ilevel);
  if (!(ncell_level > 0) || itermax < 1) return fail(RAMSES_AMD_EINVAL, "bad ncell_level/itermax");
  if (int rc = resident_release("phi_fine_cg")) return rc;
  CgCtx &G = g_cg;
  hipStream_t s = nullptr;
  ordered = cg_sum_mode(ordered);
  const long ncell = ncoarse + 8 * ngridmax;
  const size_t vb = sizeof(double) * ncell;
  HCHK(G.son.ensure(sizeof(int) * ncell), "hipMalloc son");
  HCHK(G.nbor.ensure(sizeof(int) * 6 * ngridmax), "hipMalloc nbor");
  HCHK(G.igrid.ensure(sizeof(int) * ngrid), "hipMalloc igrid");
  HCHK(G.nb.ensure(sizeof(int) * 6 * (size_t)ngrid), "hipMalloc nb");
  HCHK(G.x.ensure(vb), "hipMalloc x"); HCHK(G.r.ensure(vb), "hipMalloc r");
  HCHK(G.p.ensure(vb), "hipMalloc p"); HCHK(G.z.ensure(vb), "hipMalloc z");
  HCHK(G.scal.ensure(sizeof(double) * 8), "hipMalloc"); HCHK(G.partial.ensure(sizeof(double) * CG_MAX_BLOCKS), "hipMalloc");
  if (ordered) HCHK(G.prod.ensure(sizeof(double) * 8 * (size_t)ngrid), "hipMalloc prod");
  if (ordered == 1) HCHK(G.scan.ensure(cg_scan_bytes(ngrid)), "hipMalloc scan");
  if (!G.pin) {
    HCHK(hipHostMalloc(reinterpret_cast<void **>(&G.pin), sizeof(double) * 8, hipHostMallocMapped | hipHostMallocCoherent), "hipHostMalloc");
    HCHK(hipHostGetDevicePointer(reinterpret_cast<void **>(&G.pin_dev), G.pin, 0), "hipHostGetDevicePointer");
    for (int k = 0; k < 4; k++) HCHK(hipEventCreateWithFlags(&G.ev[k], hipEventDisableTiming), "hipEventCreate");
  }
  HCHK(hipMemcpyAsync(G.son.p, son, sizeof(int) * ncell, hipMemcpyHostToDevice, s), "H2D son");
  HCHK(hipMemcpyAsync(G.nbor.p, nbor, sizeof(int) * 6 * ngridmax, hipMemcpyHostToDevice, s), "H2D nbor");
  HCHK(hipMemcpyAsync(G.igrid.p, igrid, sizeof(int) * ngrid, hipMemcpyHostToDevice, s), "H2D igrid");
  HCHK(hipMemcpyAsync(G.x.p, phi, vb, hipMemcpyHostToDevice, s), "H2D phi");
  HCHK(hipMemcpyAsync(G.r.p, f, vb, hipMemcpyHostToDevice, s), "H2D r");
  HCHK(hipMemcpyAsync(G.p.p, f + ncell, vb, hipMemcpyHostToDevice, s), "H2D p");
  HCHK(hipMemcpyAsync(G.z.p, f + 2 * ncell, vb, hipMemcpyHostToDevice, s), "H2D z");
  HCHK(hipMemsetAsync(G.scal.p, 0, sizeof(double) * 8, s), "memset");
  HCHK(cg_launch_setup(G.igrid.as<int>(), ngrid, G.son.as<int>(), G.nbor.as<int>(), ngridmax, G.nb.as<int>(), s), "cg setup");
  CgLevel L;
  L.ngrid = ngrid; L.igrid = G.igrid.as<int>(); L.nb = G.nb.as<int>(); L.ncoarse = ncoarse; L.ngridmax = ngridmax;
  L.x = G.x.as<double>(); L.r = G.r.as<double>(); L.p = G.p.as<double>(); L.z = G.z.as<double>();
  L.host_r2 = G.pin_dev;
  L.scal = G.scal.as<double>(); L.partial = G.partial.as<double>(); L.prod = ordered ? G.prod.as<double>() : nullptr;
  L.scan = ordered == 1 ? G.scan.p : nullptr;
  double rhs_norm = 0.0;
  if (rho_or_null) {
    HCHK(G.rho.ensure(vb), "hipMalloc rho");
    HCHK(hipMemcpyAsync(G.rho.p, rho_or_null, vb, hipMemcpyHostToDevice, s), "H2D rho");
    HCHK(cg_launch_rhs_norm(L, G.rho.as<double>(), rho_tot, fact * fact, s), "cg rhs norm");
    HCHK(hipMemcpyAsync(G.pin + 4, L.scal + CG_RHS, sizeof(double), hipMemcpyDeviceToHost, s), "D2H rhs");
  }
  // r2 of iteration k is stored into pin[k & 3] by the kernel that forms it, signalled by ev[k & 3]
  HCHK(cg_launch_dot_rr(L, 1, s), "cg dot");
  HCHK(hipEventRecord(G.ev[1], s), "event");
  int iter = 0;
  double error = 1.0, error_ini = 1.0;
  while (error > epsilon * error_ini && iter < itermax) {
    iter++;
    HCHK(cg_launch_iteration(L, iter, (iter + 1) & 3, s), "cg iteration");
    HCHK(hipEventRecord(G.ev[(iter + 1) & 3], s), "event");
    HCHK(hipEventSynchronize(G.ev[iter & 3]), "event sync");
    error = std::sqrt(G.pin[iter & 3] / ncell_level);     // :186
    if (iter == 1) error_ini = error;
  }
  HCHK(hipMemcpyAsync(phi, G.x.p, vb, hipMemcpyDeviceToHost, s), "D2H phi");
  HCHK(hipMemcpyAsync(f, G.r.p, vb, hipMemcpyDeviceToHost, s), "D2H r");
  HCHK(hipMemcpyAsync(f + ncell, G.p.p, vb, hipMemcpyDeviceToHost, s), "D2H p");
  HCHK(hipMemcpyAsync(f + 2 * ncell, G.z.p, vb, hipMemcpyDeviceToHost, s), "D2H z");
  HCHK(hipStreamSynchronize(s), "sync");
  if (rho_or_null) rhs_norm = std::sqrt(G.pin[4] / ncell_level);   // :78
  *iter_out = iter;
  err_out[0] = error; err_out[1] = error_ini; err_out[2] = rhs_norm;
  return 0;
}

size_t ramses_amd_ordered_sum_scratch(int64_t n) { return ordered_sum_bytes((long)n); }
int ramses_amd_ordered_sum_device(const double *d_x, int64_t n, double *d_out, void *d_scratch, void *stream) {
  if (n < 0 || (n > 0 && !d_x) || !d_out || !d_scratch) return fail(RAMSES_AMD_EINVAL, "ordered_sum: bad argument");
  HCHK(ordered_sum_launch(d_x, (long)n, d_out, d_scratch, static_cast<hipStream_t>(stream)), "ordered sum");
  return 0;
}

// ---------------------------------------------------------------------------
// The same solver with several MPI ranks: the reference's loop stays in the caller (the Fortran shim), which owns
// the two MPI_ALLREDUCEs per iteration (poisson/phi_fine_cg.f90:108,154) and the halo exchange of p (:134); every
// loop body is a device routine on the rank's octs.  Local sums land in the device scalars; the caller reads them
// (cgmpi_get), reduces them over the ranks and writes the global value back (cgmpi_set) before the next routine
// uses it -- alpha and beta are formed on the device from those scalars as in the single-rank loop.  The virtual
// cells of p travel through the host array f(:,2) around the reference's own make_virtual_fine_dp.
// ---------------------------------------------------------------------------
extern "C++" {
namespace {
struct CgMpi {
  bool open = false;
  CgLevel L;
  long ncell = 0;
  double *h_f = nullptr;
  DevBuf list, pack;
  std::vector<double> hpack;
  // the level's communicators on the device (round 3): p's virtual cells are exchanged from the device vector
  bool comm = false;
  int ncpu = 0;
  std::vector<int> em_first, rc_first;
  DevBuf em_ig, rc_ig, sendbuf, recvbuf;
  void *h_send = nullptr, *h_recv = nullptr;
  size_t h_send_cap = 0, h_recv_cap = 0;
  bool halo_open = false;
};
CgMpi g_cgm;

__global__ void cg_cells_kernel(double *vec, double *buf, const int *igrid, int n, long ncoarse, long ngridmax, int gather) {
  const long total = (long)n * 8;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    const long c = ncoarse + (t / n) * ngridmax + igrid[t % n] - 1;
    if (gather) buf[t] = vec[c]; else vec[c] = buf[t];
  }
}
}  // namespace
}  // extern "C++"

// upload the state the loop starts from (as ramses_amd_cg_solve_host); out2 = {local rhs norm^2 (0 without rho), local r.r}
int ramses_amd_cgmpi_begin(int ilevel, int ngrid, const int *igrid, const int *son, const int *nbor, int64_t ngridmax, int64_t ncoarse,
                           const double *phi, double *f, const double *rho_or_null, double rho_tot, double fact, int ordered,
                           double *out2) {
  if (!son || !nbor || !phi || !f || !out2 || (ngrid > 0 && !igrid)) return fail(RAMSES_AMD_EINVAL, "NULL argument");
  if (ngrid < 0 || ngridmax < ngrid || ncoarse < 1) return fail(RAMSES_AMD_EINVAL, "bad level sizes (ngrid=%d)", ngrid);
  if (ilevel < 1 || ilevel > 30) return fail(RAMSES_AMD_EINVAL, "bad level %d", ilevel);
  if (int rc = resident_release("phi_fine_cg")) return rc;
  CgCtx &G = g_cg;
  CgMpi &M = g_cgm;
  hipStream_t s = nullptr;
  ordered = cg_sum_mode(ordered);
  const long ncell = ncoarse + 8 * ngridmax;
  const size_t vb = sizeof(double) * ncell;
  const int ng1 = ngrid > 0 ? ngrid : 1;
  HCHK(G.son.ensure(sizeof(int) * ncell), "hipMalloc son");
  HCHK(G.nbor.ensure(sizeof(int) * 6 * ngridmax), "hipMalloc nbor");
  HCHK(G.igrid.ensure(sizeof(int) * ng1), "hipMalloc igrid");
  HCHK(G.nb.ensure(sizeof(int) * 6 * (size_t)ng1), "hipMalloc nb");
  HCHK(G.x.ensure(vb), "hipMalloc x"); HCHK(G.r.ensure(vb), "hipMalloc r");
  HCHK(G.p.ensure(vb), "hipMalloc p"); HCHK(G.z.ensure(vb), "hipMalloc z");
  HCHK(G.scal.ensure(sizeof(double) * 8), "hipMalloc"); HCHK(G.partial.ensure(sizeof(double) * CG_MAX_BLOCKS), "hipMalloc");
  if (ordered) HCHK(G.prod.ensure(sizeof(double) * 8 * (size_t)ng1), "hipMalloc prod");
  if (ordered == 1) HCHK(G.scan.ensure(cg_scan_bytes(ng1)), "hipMalloc scan");
  if (!G.pin) {
    HCHK(hipHostMalloc(reinterpret_cast<void **>(&G.pin), sizeof(double) * 8, hipHostMallocMapped | hipHostMallocCoherent), "hipHostMalloc");
    HCHK(hipHostGetDevicePointer(reinterpret_cast<void **>(&G.pin_dev), G.pin, 0), "hipHostGetDevicePointer");
    for (int k = 0; k < 4; k++) HCHK(hipEventCreateWithFlags(&G.ev[k], hipEventDisableTiming), "hipEventCreate");
  }
  HCHK(hipMemcpyAsync(G.son.p, son, sizeof(int) * ncell, hipMemcpyHostToDevice, s), "H2D son");
  HCHK(hipMemcpyAsync(G.nbor.p, nbor, sizeof(int) * 6 * ngridmax, hipMemcpyHostToDevice, s), "H2D nbor");
  if (ngrid > 0) HCHK(hipMemcpyAsync(G.igrid.p, igrid, sizeof(int) * ngrid, hipMemcpyHostToDevice, s), "H2D igrid");
  HCHK(hipMemcpyAsync(G.x.p, phi, vb, hipMemcpyHostToDevice, s), "H2D phi");
  HCHK(hipMemcpyAsync(G.r.p, f, vb, hipMemcpyHostToDevice, s), "H2D r");
  HCHK(hipMemcpyAsync(G.p.p, f + ncell, vb, hipMemcpyHostToDevice, s), "H2D p");
  HCHK(hipMemcpyAsync(G.z.p, f + 2 * ncell, vb, hipMemcpyHostToDevice, s), "H2D z");
  HCHK(hipMemsetAsync(G.scal.p, 0, sizeof(double) * 8, s), "memset");
  HCHK(cg_launch_setup(G.igrid.as<int>(), ngrid, G.son.as<int>(), G.nbor.as<int>(), ngridmax, G.nb.as<int>(), s), "cg setup");
  CgLevel &L = M.L;
  L.ngrid = ngrid; L.igrid = G.igrid.as<int>(); L.nb = G.nb.as<int>(); L.ncoarse = ncoarse; L.ngridmax = ngridmax;
  L.x = G.x.as<double>(); L.r = G.r.as<double>(); L.p = G.p.as<double>(); L.z = G.z.as<double>();
  L.host_r2 = G.pin_dev;
  L.scal = G.scal.as<double>(); L.partial = G.partial.as<double>(); L.prod = ordered ? G.prod.as<double>() : nullptr;
  L.scan = ordered == 1 ? G.scan.p : nullptr;
  M.ncell = ncell; M.h_f = f;
  out2[0] = 0.0; out2[1] = 0.0;
  if (rho_or_null) {
    HCHK(G.rho.ensure(vb), "hipMalloc rho");
    HCHK(hipMemcpyAsync(G.rho.p, rho_or_null, vb, hipMemcpyHostToDevice, s), "H2D rho");
    HCHK(cg_launch_rhs_norm(L, G.rho.as<double>(), rho_tot, fact * fact, s), "cg rhs norm");
    HCHK(hipMemcpyAsync(&out2[0], L.scal + CG_RHS, sizeof(double), hipMemcpyDeviceToHost, s), "D2H rhs");
  }
  HCHK(cg_launch_dot_rr(L, 1, s), "cg dot");
  HCHK(hipMemcpyAsync(&out2[1], L.scal + CG_R2, sizeof(double), hipMemcpyDeviceToHost, s), "D2H r2");
  HCHK(hipStreamSynchronize(s), "sync");
  M.open = true;
  return 0;
}
#define CGM_OPEN(what) do { if (!g_cgm.open) return fail(RAMSES_AMD_EINVAL, "%s: no CG solve is open (ramses_amd_cgmpi_begin)", what); } while (0)
// device scalars: slot 0 r.r, 1 r.r of the previous iteration, 2 p.Ap
int ramses_amd_cgmpi_get(int slot, double *val) {
  CGM_OPEN("cgmpi_get");
  if (slot < 0 || slot > 3 || !val) return fail(RAMSES_AMD_EINVAL, "bad argument");
  HCHK(hipMemcpy(val, g_cgm.L.scal + slot, sizeof(double), hipMemcpyDeviceToHost), "D2H scalar");
  return 0;
}
int ramses_amd_cgmpi_set(int slot, double val) {
  CGM_OPEN("cgmpi_set");
  if (slot < 0 || slot > 3) return fail(RAMSES_AMD_EINVAL, "bad argument");
  HCHK(hipMemcpy(g_cgm.L.scal + slot, &val, sizeof(double), hipMemcpyHostToDevice), "H2D scalar");
  return 0;
}
// step 0: p = r + beta p (:116-133); 1: z = A p and the local p.z (:139-153); 2: x += alpha p, r -= alpha z and the local r.r
// of the next iteration (:160-183, :98-105)
int ramses_amd_cgmpi_step(int step, int iter) {
  CGM_OPEN("cgmpi_step");
  hipError_t e;
  switch (step) {
    case 0: e = cg_launch_update_p(g_cgm.L, iter, nullptr); break;
    case 1: e = cg_launch_ap(g_cgm.L, nullptr); break;
    case 2: e = cg_launch_update_xr(g_cgm.L, nullptr); break;
    default: return fail(RAMSES_AMD_EINVAL, "bad step %d", step);
  }
  HCHK(e, "cg step");
  return 0;
}
// cells of the listed octs of p: device -> host array f(:,2) (to_host != 0: the emission octs before the exchange) or
// host -> device (the reception octs after it)
int ramses_amd_cgmpi_p_cells(int n, const int *igrid, int to_host) {
  CGM_OPEN("cgmpi_p_cells");
  CgMpi &M = g_cgm;
  if (n < 0 || (n > 0 && !igrid)) return fail(RAMSES_AMD_EINVAL, "bad oct list");
  if (n == 0) return 0;
  const long tot = (long)n * 8;
  double *hp = M.h_f + M.ncell;      // f(:,2)
  HCHK(M.list.ensure(sizeof(int) * (size_t)n), "hipMalloc"); HCHK(M.pack.ensure(sizeof(double) * (size_t)tot), "hipMalloc");
  HCHK(hipMemcpy(M.list.p, igrid, sizeof(int) * (size_t)n, hipMemcpyHostToDevice), "H2D list");
  M.hpack.resize((size_t)tot);
  int nb = (int)((tot + 255) / 256);
  if (nb > 4096) nb = 4096;
  if (to_host) {
    hipLaunchKernelGGL(cg_cells_kernel, dim3(nb), dim3(256), 0, nullptr, M.L.p, M.pack.as<double>(), M.list.as<int>(), n, M.L.ncoarse, M.L.ngridmax, 1);
    HCHK(hipGetLastError(), "gather launch");
    HCHK(hipMemcpy(M.hpack.data(), M.pack.p, sizeof(double) * (size_t)tot, hipMemcpyDeviceToHost), "D2H p");
    for (int ind = 0; ind < 8; ind++)
      for (int i = 0; i < n; i++) hp[M.L.ncoarse + (long)ind * M.L.ngridmax + igrid[i] - 1] = M.hpack[(size_t)ind * n + i];
  } else {
    for (int ind = 0; ind < 8; ind++)
      for (int i = 0; i < n; i++) M.hpack[(size_t)ind * n + i] = hp[M.L.ncoarse + (long)ind * M.L.ngridmax + igrid[i] - 1];
    HCHK(hipMemcpy(M.pack.p, M.hpack.data(), sizeof(double) * (size_t)tot, hipMemcpyHostToDevice), "H2D p");
    hipLaunchKernelGGL(cg_cells_kernel, dim3(nb), dim3(256), 0, nullptr, M.L.p, M.pack.as<double>(), M.list.as<int>(), n, M.L.ncoarse, M.L.ngridmax, 0);
    HCHK(hipGetLastError(), "scatter launch");
  }
  return 0;
}
// make_virtual_fine_dp(f(1,2),ilevel) of the loop (poisson/phi_fine_cg.f90:134) on the DEVICE vector p: comm_set sends the level's
// emission / reception oct lists once per solve; one message per peer in the reference's layout (u(i + (ind-1)*n)); RCCL
// (p_halo_rccl) or the caller's own MPI on pinned host buffers between p_halo_stage_out and p_halo_stage_in
int ramses_amd_cgmpi_comm_set(int ncpu, const int *em_n, const int *em_ig, const int *rc_n, const int *rc_ig) {
  CGM_OPEN("cgmpi_comm_set");
  CgMpi &M = g_cgm;
  if (ncpu < 1 || !em_n || !rc_n) return fail(RAMSES_AMD_EINVAL, "cgmpi_comm_set: bad argument");
  M.comm = false; M.ncpu = ncpu;
  M.em_first.assign((size_t)ncpu + 1, 0); M.rc_first.assign((size_t)ncpu + 1, 0);
  for (int c = 0; c < ncpu; c++) {
    if (em_n[c] < 0 || rc_n[c] < 0) return fail(RAMSES_AMD_EINVAL, "cgmpi_comm_set: negative list length");
    M.em_first[c + 1] = M.em_first[c] + em_n[c];
    M.rc_first[c + 1] = M.rc_first[c] + rc_n[c];
  }
  const int nem = M.em_first[ncpu], nrc = M.rc_first[ncpu];
  if ((nem > 0 && !em_ig) || (nrc > 0 && !rc_ig)) return fail(RAMSES_AMD_EINVAL, "cgmpi_comm_set: NULL list");
  HCHK(M.em_ig.ensure(sizeof(int) * (size_t)(nem > 0 ? nem : 1)), "hipMalloc"); HCHK(M.rc_ig.ensure(sizeof(int) * (size_t)(nrc > 0 ? nrc : 1)), "hipMalloc");
  if (nem > 0) HCHK(hipMemcpy(M.em_ig.p, em_ig, sizeof(int) * (size_t)nem, hipMemcpyHostToDevice), "H2D emission list");
  if (nrc > 0) HCHK(hipMemcpy(M.rc_ig.p, rc_ig, sizeof(int) * (size_t)nrc, hipMemcpyHostToDevice), "H2D reception list");
  HCHK(M.sendbuf.ensure(sizeof(double) * 8 * (size_t)(nem > 0 ? nem : 1)), "hipMalloc sendbuf");
  HCHK(M.recvbuf.ensure(sizeof(double) * 8 * (size_t)(nrc > 0 ? nrc : 1)), "hipMalloc recvbuf");
  M.comm = true;
  return 0;
}
namespace {
int cgmpi_pack(CgMpi &M) {
  for (int c = 0; c < M.ncpu; c++) {
    const int n = M.em_first[c + 1] - M.em_first[c];
    if (n <= 0) continue;
    int nb = (int)((8L * n + 255) / 256);
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(cg_cells_kernel, dim3(nb), dim3(256), 0, nullptr, M.L.p, M.sendbuf.as<double>() + 8L * M.em_first[c],
                       M.em_ig.as<int>() + M.em_first[c], n, M.L.ncoarse, M.L.ngridmax, 1);
  }
  HCHK(hipGetLastError(), "cg halo pack launch");
  return 0;
}
int cgmpi_unpack(CgMpi &M) {
  for (int c = 0; c < M.ncpu; c++) {
    const int n = M.rc_first[c + 1] - M.rc_first[c];
    if (n <= 0) continue;
    int nb = (int)((8L * n + 255) / 256);
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(cg_cells_kernel, dim3(nb), dim3(256), 0, nullptr, M.L.p, M.recvbuf.as<double>() + 8L * M.rc_first[c],
                       M.rc_ig.as<int>() + M.rc_first[c], n, M.L.ncoarse, M.L.ngridmax, 0);
  }
  HCHK(hipGetLastError(), "cg halo unpack launch");
  return 0;
}
}  // namespace
int ramses_amd_cgmpi_p_halo_stage_out(int ncpu, int64_t *h_send_addr, int64_t *h_recv_addr, int64_t *send_off, int64_t *recv_off) {
  CGM_OPEN("cgmpi_p_halo_stage_out");
  CgMpi &M = g_cgm;
  if (!M.comm || ncpu != M.ncpu) return fail(RAMSES_AMD_EINVAL, "cgmpi_p_halo_stage_out: no communicators (ramses_amd_cgmpi_comm_set) / ncpu mismatch");
  if (!h_send_addr || !h_recv_addr || !send_off || !recv_off) return fail(RAMSES_AMD_EINVAL, "NULL argument");
  if (int rc = cgmpi_pack(M)) return rc;
  const size_t ns = 8 * (size_t)M.em_first[ncpu], nr = 8 * (size_t)M.rc_first[ncpu];
  if (int rc = pin_ensure(M.h_send, M.h_send_cap, sizeof(double) * (ns > 0 ? ns : 1))) return rc;
  if (int rc = pin_ensure(M.h_recv, M.h_recv_cap, sizeof(double) * (nr > 0 ? nr : 1))) return rc;
  if (ns > 0) HCHK(hipMemcpyAsync(M.h_send, M.sendbuf.p, sizeof(double) * ns, hipMemcpyDeviceToHost, nullptr), "D2H halo");
  HCHK(hipStreamSynchronize(nullptr), "sync");
  *h_send_addr = (int64_t)(intptr_t)M.h_send; *h_recv_addr = (int64_t)(intptr_t)M.h_recv;
  for (int c = 0; c <= ncpu; c++) { send_off[c] = 8 * (int64_t)M.em_first[c]; recv_off[c] = 8 * (int64_t)M.rc_first[c]; }
  M.halo_open = true;
  return 0;
}
int ramses_amd_cgmpi_p_halo_stage_in(void) {
  CGM_OPEN("cgmpi_p_halo_stage_in");
  CgMpi &M = g_cgm;
  if (!M.halo_open) return fail(RAMSES_AMD_EINVAL, "cgmpi_p_halo_stage_in without cgmpi_p_halo_stage_out");
  M.halo_open = false;
  const size_t nr = 8 * (size_t)M.rc_first[M.ncpu];
  if (nr > 0) HCHK(hipMemcpyAsync(M.recvbuf.p, M.h_recv, sizeof(double) * nr, hipMemcpyHostToDevice, nullptr), "H2D halo");
  return cgmpi_unpack(M);
}
int ramses_amd_cgmpi_p_halo_rccl(void) {
  CGM_OPEN("cgmpi_p_halo_rccl");
  CgMpi &M = g_cgm;
  if (!M.comm) return fail(RAMSES_AMD_EINVAL, "cgmpi_p_halo_rccl: no communicators (ramses_amd_cgmpi_comm_set)");
  if (int rc = cgmpi_pack(M)) return rc;
  std::vector<int> peer;
  std::vector<int64_t> so, sc, ro, rcn;
  for (int c = 0; c < M.ncpu; c++) {
    const int64_t ns = 8 * (int64_t)(M.em_first[c + 1] - M.em_first[c]), nr = 8 * (int64_t)(M.rc_first[c + 1] - M.rc_first[c]);
    if (ns == 0 && nr == 0) continue;
    peer.push_back(c); so.push_back(8 * (int64_t)M.em_first[c]); sc.push_back(ns); ro.push_back(8 * (int64_t)M.rc_first[c]); rcn.push_back(nr);
  }
  if (int rc = ramses_amd_rccl_exchange((int)peer.size(), peer.data(), M.sendbuf.as<double>(), so.data(), sc.data(), M.recvbuf.as<double>(),
                                        ro.data(), rcn.data(), nullptr)) return rc;
  return cgmpi_unpack(M);
}
// phi and f = (r, p, A p) back into the host arrays (what the reference's loop leaves)
int ramses_amd_cgmpi_end(double *phi, double *f) {
  CGM_OPEN("cgmpi_end");
  if (!phi || !f) return fail(RAMSES_AMD_EINVAL, "NULL argument");
  CgCtx &G = g_cg;
  const size_t vb = sizeof(double) * g_cgm.ncell;
  HCHK(hipMemcpy(phi, G.x.p, vb, hipMemcpyDeviceToHost), "D2H phi");
  HCHK(hipMemcpy(f, G.r.p, vb, hipMemcpyDeviceToHost), "D2H r");
  HCHK(hipMemcpy(f + g_cgm.ncell, G.p.p, vb, hipMemcpyDeviceToHost), "D2H p");
  HCHK(hipMemcpy(f + 2 * g_cgm.ncell, G.z.p, vb, hipMemcpyDeviceToHost), "D2H z");
  g_cgm.open = false;
  g_cgm.comm = false; g_cgm.halo_open = false;
  return 0;
}
#undef CGM_OPEN
#undef HCHK

// ---------------------------------------------------------------------------
// coarse <-> fine hydro operators on a periodic coarse brick and its fully
// refined child brick
// ---------------------------------------------------------------------------
static int amr_op(bool prolong, int nc, int nvar, int interpol_var, int interpol_type, double smallr,
                  double *d_coarse, double *d_fine, void *stream) {
  if (!d_coarse || !d_fine) return fail(RAMSES_AMD_EINVAL, "NULL argument");
  if (nc < 2) return fail(RAMSES_AMD_EINVAL, "coarse brick must have >=2 cells per direction");
  if (nvar != 5) return fail(RAMSES_AMD_EUNSUPPORTED, "coarse<->fine operators implement NVAR=5 (got %d)", nvar);
  if (interpol_var < 0 || interpol_var > 2) return fail(RAMSES_AMD_EINVAL, "interpol_var must be 0, 1 or 2");
  if (prolong && (interpol_type < 1 || interpol_type > 4)) return fail(RAMSES_AMD_EINVAL, "interpol_type must be 1..4");
  if (prolong && interpol_type == 4 && interpol_var != 2) return fail(RAMSES_AMD_EINVAL, "interpol_type=4 is designed for interpol_var=2");
  AmrOpArgs A;
  A.coarse = d_coarse; A.fine = d_fine; A.nc = nc; A.nvar = nvar;
  A.interpol_var = interpol_var; A.interpol_type = interpol_type; A.smallr = smallr;
  hipError_t e = launch_amr_op(A, prolong, reinterpret_cast<hipStream_t>(stream));
  if (e != hipSuccess) return hipfail(e, prolong ? "interpol_hydro launch" : "upload_fine launch");
  return 0;
}

int ramses_amd_interpol_hydro_brick(int nc, int nvar, int interpol_var, int interpol_type, double smallr,
                                    const double *d_coarse, double *d_fine, void *stream) {
  return amr_op(true, nc, nvar, interpol_var, interpol_type, smallr, const_cast<double *>(d_coarse), d_fine, stream);
}

int ramses_amd_upload_fine_brick(int nc, int nvar, int interpol_var, double smallr, const double *d_fine,
                                 double *d_coarse, void *stream) {
  return amr_op(false, nc, nvar, interpol_var, 1, smallr, d_coarse, const_cast<double *>(d_fine), stream);
}

}  // extern "C"

// ---------------------------------------------------------------------------
// Load the device code of every translation unit now (see warm.hpp) and create the HIP context: called once by
// the Fortran side from the first shim the program reaches, so that the one-time costs (~0.2 s) fall into the
// reference's initialisation phase and not into the first multigrid_fine / godunov_fine of its timed loop.
// ---------------------------------------------------------------------------
extern "C" int ramses_amd_warm_hydro_misc(void);
extern "C" int ramses_amd_warm_mg_kernels(void);
extern "C" int ramses_amd_warm_octree_pack(void);
extern "C" int ramses_amd_warm_amr_ops(void);
extern "C" int ramses_amd_warm_amr_sweep(void);
extern "C" int ramses_amd_warm_mg_amr(void);
extern "C" int ramses_amd_warm_cg_amr(void);
extern "C" int ramses_amd_warm_rho_fine(void);
extern "C" int ramses_amd_warm_capi_mpi(void);
extern "C" int ramses_amd_warm_capi_amr(void);
extern "C" int ramses_amd_warm_pois_amr(void);
extern "C" int ramses_amd_warm_capi(void);
extern "C" int ramses_amd_warm_hydro_sweep_fast(void);
extern "C" int ramses_amd_warm_hydro_sweep_strict(void);
extern "C" int ramses_amd_warm_mhd_sweep(void);

extern "C" int ramses_amd_warmup(void) {
  static bool done = false;
  if (done) return 0;
  done = true;
  if (hipFree(nullptr) != hipSuccess) { (void)hipGetLastError(); return fail(RAMSES_AMD_ENODEVICE, "no usable HIP device"); }
  int bad = 0;
  bad += ramses_amd_warm_hydro_misc();
  bad += ramses_amd_warm_mg_kernels();
  bad += ramses_amd_warm_octree_pack();
  bad += ramses_amd_warm_amr_ops();
  bad += ramses_amd_warm_amr_sweep();
  bad += ramses_amd_warm_mg_amr();
  bad += ramses_amd_warm_cg_amr();
  bad += ramses_amd_warm_rho_fine();
  bad += ramses_amd_warm_capi_mpi();
  bad += ramses_amd_warm_capi_amr();
  bad += ramses_amd_warm_pois_amr();
  bad += ramses_amd_warm_capi();
  bad += ramses_amd_warm_hydro_sweep_fast();
  bad += ramses_amd_warm_hydro_sweep_strict();
  bad += ramses_amd_warm_mhd_sweep();
  if (hipDeviceSynchronize() != hipSuccess || bad) return fail(RAMSES_AMD_EHIP, "warm-up launches failed (%d)", bad);
  return 0;
}

#include "warm.hpp"
RAMSES_AMD_TU_WARM(capi)
