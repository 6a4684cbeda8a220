// capi.hip -- the C ABI of libramses_amd.so (declared in include/ramses_amd.h).
// Plain pointers and PODs only; validates arguments, derives constants,
// dispatches to the gfx950 kernels.  No CPU fallback: anything that cannot run
// on the device returns an error code.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>

#include "../../include/ramses_amd.h"
#include "misc_args.hpp"
#include "sweep_args.hpp"

using namespace ramses_amd;

static thread_local char g_err[512] = "";
static int g_tile_rows = 0;  // 0 = per-variant default
static int g_zchunk = 128;

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
  if (b->nx < 2 || b->ny < 2 || b->nz < 2) return fail(RAMSES_AMD_EINVAL, "brick must have >=2 cells per direction (got %d %d %d)", b->nx, b->ny, b->nz);
  if (b->ng != 0 && b->ng < 2) return fail(RAMSES_AMD_EINVAL, "ghost width must be 0 or >=2 (got %d)", b->ng);
  if (b->pitch_y < b->nx + 2 * b->ng) return fail(RAMSES_AMD_EINVAL, "pitch_y too small");
  if (b->pitch_z < b->pitch_y * (b->ny + 2 * b->ng)) return fail(RAMSES_AMD_EINVAL, "pitch_z too small");
  if (b->pitch_var < b->pitch_z * (b->nz + 2 * b->ng)) return fail(RAMSES_AMD_EINVAL, "pitch_var too small");
  return 0;
}

extern "C" {

const char *ramses_amd_last_error(void) { return g_err; }

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

int ramses_amd_godunov_brick(const ramses_amd_hydro_params *p, const ramses_amd_brick *b,
                             const double *d_uold, const double *d_grav, double *d_unew,
                             double dx, double dt, void *stream) {
  if (!p) return fail(RAMSES_AMD_EINVAL, "params is NULL");
  if (int rc = check_brick(b)) return rc;
  if (!d_uold || !d_unew) return fail(RAMSES_AMD_EINVAL, "uold/unew device pointers are NULL");
  if (d_uold == d_unew) return fail(RAMSES_AMD_EINVAL, "uold and unew must be distinct buffers");
  if (p->ndim != 3) return fail(RAMSES_AMD_EUNSUPPORTED, "device sweep implements NDIM=3 (got %d)", p->ndim);
  if (p->nvar != 5) return fail(RAMSES_AMD_EUNSUPPORTED, "device sweep implements NVAR=5 (got %d)", p->nvar);
  if (p->scheme != RAMSES_AMD_SCHEME_MUSCL) return fail(RAMSES_AMD_EUNSUPPORTED, "scheme='plmde' is not implemented on the device yet");
  if (p->difmag > 0.0) return fail(RAMSES_AMD_EUNSUPPORTED, "difmag>0 is not implemented on the device yet");
  if (!(p->slope_type == 0 || p->slope_type == 1 || p->slope_type == 2 || p->slope_type == 7 || p->slope_type == 8))
    return fail(RAMSES_AMD_EUNSUPPORTED, "slope_type=%d is not implemented on the device (0,1,2,7,8 are)", p->slope_type);
  if (p->riemann < 0 || p->riemann > 4) return fail(RAMSES_AMD_EINVAL, "unknown Riemann solver %d", p->riemann);
  if (!(dx > 0.0) || !(dt >= 0.0)) return fail(RAMSES_AMD_EINVAL, "dx must be >0 and dt >=0");

  SweepArgs A;
  A.uold = d_uold; A.unew = d_unew; A.grav = d_grav;
  A.nx = b->nx; A.ny = b->ny; A.nz = b->nz; A.ng = b->ng;
  A.pitch_y = b->pitch_y; A.pitch_z = b->pitch_z; A.pitch_var = b->pitch_var;
  A.zchunk = g_zchunk < b->nz ? g_zchunk : b->nz;
  A.dt = dt; A.dx = dx; A.rdx = 1.0 / dx;
  A.P = make_const(p);
  const bool pow2 = is_pow2(dx);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  hipError_t e = p->fast_math
                     ? fastmode::launch_godunov_sweep(A, p->slope_type, p->riemann, g_tile_rows, d_grav != nullptr, pow2, s)
                     : strictmode::launch_godunov_sweep(A, p->slope_type, p->riemann, g_tile_rows, d_grav != nullptr, pow2, s);
  if (e != hipSuccess) return hipfail(e, "godunov sweep launch");
  return 0;
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
  if (p->ndim != 3 || p->nvar != 5) return fail(RAMSES_AMD_EUNSUPPORTED, "device courant implements NDIM=3, NVAR=5");
  CourantArgs A;
  A.uold = d_uold; A.grav = d_grav; A.out = d_out;
  A.nx = b->nx; A.ny = b->ny; A.nz = b->nz; A.ng = b->ng;
  A.pitch_y = b->pitch_y; A.pitch_z = b->pitch_z; A.pitch_var = b->pitch_var;
  A.dx = dx; A.vol = dx * dx * dx;
  A.courant_factor = p->courant_factor;
  A.dt_init = p->courant_factor * dx / p->smallc;
  A.P = make_const(p);
  hipError_t e = launch_courant(A, d_grav != nullptr, reinterpret_cast<hipStream_t>(stream));
  if (e != hipSuccess) return hipfail(e, "courant launch");
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

}  // extern "C"
