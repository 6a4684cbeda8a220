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
#include "capi_shared.hpp"

using namespace ramses_amd;

static thread_local char g_err[512] = "";
static int g_tile_rows = 0;  // 0 = per-variant default
static int g_zchunk = 128;
static int g_mg_fused = 1;

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
int ramses_amd_set_error(int code, const char *msg) {
  snprintf(g_err, sizeof(g_err), "%s", msg ? msg : "");
  return code;
}

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
  // The lanes of the sweep address a plane with a 32-bit byte offset and the planes of a variable with a 32-bit scalar one
  // (csrc/hydro_sweep.hip: plane_load / plane_store): a brick beyond that is REFUSED here, by name -- never swept wrongly,
  // never routed elsewhere without a word (the largest cubic brick: 812^3 cells per variable)
  if ((unsigned long)b->pitch_z * 8ul >= (1ul << 31) || (unsigned long)b->pitch_var * 8ul >= (1ul << 32))
    return fail(RAMSES_AMD_EUNSUPPORTED,
                "brick of %d x %d x %d cells (+%d ghost layers): a plane of %ld bytes or a variable of %ld bytes is beyond the 32-bit offsets of the "
                "sweep kernel (planes < 2 GiB, variables < 4 GiB): split the level into more bricks (ranks)",
                b->nx, b->ny, b->nz, b->ng, (long)b->pitch_z * 8, (long)b->pitch_var * 8);
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
extern "C" int ramses_amd_warm_capi_host(void);
extern "C" int ramses_amd_warm_capi_tree_poisson(void);
extern "C" int ramses_amd_warm_hydro_sweep_fast(void);
extern "C" int ramses_amd_warm_hydro_sweep_strict(void);
extern "C" int ramses_amd_warm_mhd_sweep(void);
extern "C" int ramses_amd_warm_amr_sweep_st0(void);
extern "C" int ramses_amd_warm_amr_sweep_st1(void);
extern "C" int ramses_amd_warm_amr_sweep_st2(void);
extern "C" int ramses_amd_warm_amr_sweep_st3(void);
extern "C" int ramses_amd_warm_amr_sweep_st7(void);
extern "C" int ramses_amd_warm_amr_sweep_st8(void);

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
  bad += ramses_amd_warm_amr_sweep_st0(); bad += ramses_amd_warm_amr_sweep_st1(); bad += ramses_amd_warm_amr_sweep_st2();
  bad += ramses_amd_warm_amr_sweep_st3(); bad += ramses_amd_warm_amr_sweep_st7(); bad += ramses_amd_warm_amr_sweep_st8();
  bad += ramses_amd_warm_mg_amr();
  bad += ramses_amd_warm_cg_amr();
  bad += ramses_amd_warm_rho_fine();
  bad += ramses_amd_warm_capi_mpi();
  bad += ramses_amd_warm_capi_amr();
  bad += ramses_amd_warm_pois_amr();
  bad += ramses_amd_warm_capi();
  bad += ramses_amd_warm_capi_host();
  bad += ramses_amd_warm_capi_tree_poisson();
  bad += ramses_amd_warm_hydro_sweep_fast();
  bad += ramses_amd_warm_hydro_sweep_strict();
  bad += ramses_amd_warm_mhd_sweep();
  if (hipDeviceSynchronize() != hipSuccess || bad) return fail(RAMSES_AMD_EHIP, "warm-up launches failed (%d)", bad);
  return 0;
}

#include "warm.hpp"
RAMSES_AMD_TU_WARM(capi)
