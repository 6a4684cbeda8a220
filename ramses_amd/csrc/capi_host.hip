// capi_host.hip -- the entry points the Fortran shims of ramses_amd/patch/ bind on the reference's own HOST arrays
// (declared in include/ramses_amd.h): the staged paths (arrays up, brick kernels, results back) and the uniform level
// resident on the device between the routines of amr_step (ramses_amd_resident_*).  Split from capi.hip in round 4.
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

// ---------------------------------------------------------------------------
// Host-array entry points: what the Fortran shims of ramses_amd/patch/ bind.
// They take the reference's own arrays (Fortran-owned, host memory), stage
// them on the device, run the brick kernels and write the results back.
// ---------------------------------------------------------------------------
namespace {
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

// The staged entry points reuse the staging buffers of the resident level.  If that level holds
// the only current copy of the hydro state, dropping it would silently lose a step: refuse, as
// ramses_amd_resident_invalidate does (the caller syncs the host array first).
int ramses_amd::capi_resident_release(const char *who) {
  HostCtx &H = g_host;
  if (H.res_valid && H.res_host_stale)
    return fail(RAMSES_AMD_EINVAL, "%s: level %d is resident on the device and the host array is stale; call ramses_amd_resident_sync_host_f90 first", who, H.res_level);
  H.res_valid = false;
  return 0;
}
static int resident_release(const char *who) { return ramses_amd::capi_resident_release(who); }

extern "C" {

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

// godunov_fine(ilevel) of an NDIM = 1 or 2 build of the reference on its own arrays (BASELINE config C1: sedov1d.nml on one
// uniform level): a fully refined level without finer octs, one rank, hydro only.  The level -- its active octs and, where
// the box has physical boundaries, the octs of the boundary regions next to it (make_boundary_hydro has just filled them,
// amr/amr_step.f90:293) -- is assembled by position into a brick with two ghost layers, embedded in three dimensions (ny and /
// or nz = 1, the missing momentum components zero: the transverse slopes and flux differences vanish identically and the
// dense sweep returns the 1-D / 2-D result of the reference bit for bit, tests/test_embedded_ndim_gpu.py); directions without
// boundary octs are periodic.  The cell vectors are uold(1:ncell,1:ndim+2) with ncell = ncoarse + 2^ndim ngridmax; xg is
// xg(1:ngridmax,1:ndim) in coarse-cell units, skip = (icoarse_min, jcoarse_min), nloc = interior coarse cells per direction.
// The level is a few thousand cells: the brick is put together on the host.
// what godunov_fine of an NDIM<3 build did, level by level: sweeps on the device / sweeps the drop-in handed to the reference's host
// routine (ramses_amd_lowdim_note_reference).  One line at exit whenever anything was counted: a run that regrids away from the
// uniform level is a CPU run from then on and must not look like a device run.
static long g_lowdim_dev[32] = {0}, g_lowdim_ref[32] = {0};
static void lowdim_report(void) {
  long nd = 0, nr = 0;
  for (int l = 0; l < 32; l++) { nd += g_lowdim_dev[l]; nr += g_lowdim_ref[l]; }
  if (nd + nr == 0) return;
  printf(" ramses_amd: NDIM<3 godunov_fine: %ld sweeps on the device, %ld through the reference's host routine; per level (device/reference):", nd, nr);
  for (int l = 0; l < 32; l++) if (g_lowdim_dev[l] + g_lowdim_ref[l]) printf(" %d:%ld/%ld", l, g_lowdim_dev[l], g_lowdim_ref[l]);
  printf("\n");
  fflush(stdout);
}
static void lowdim_register(void) {
  static bool registered = false;
  if (!registered) { registered = true; atexit(lowdim_report); }
}
int ramses_amd_lowdim_note_reference(int ilevel) {
  lowdim_register();
  if (ilevel >= 0 && ilevel < 32) g_lowdim_ref[ilevel]++;
  return 0;
}
int64_t ramses_amd_lowdim_device_sweeps(void) { long n = 0; for (int l = 0; l < 32; l++) n += g_lowdim_dev[l]; return n; }
int64_t ramses_amd_lowdim_reference_sweeps(void) { long n = 0; for (int l = 0; l < 32; l++) n += g_lowdim_ref[l]; return n; }

int ramses_amd_godunov_fine_lowdim_f90(const ramses_amd_hydro_params *p, int ilevel, int ngrid, const int *igrid, int nbound,
                                       const int *igrid_bound, const double *xg, int64_t ngridmax, int64_t ncoarse, const int *skip,
                                       const int *nloc, const double *uold, double *unew, double dx, double dt) {
  if (!p || !igrid || !xg || !skip || !nloc || !uold || !unew || (nbound > 0 && !igrid_bound)) return fail(RAMSES_AMD_EINVAL, "NULL argument");
  const int ndim = p->ndim;
  if (ndim != 1 && ndim != 2) return fail(RAMSES_AMD_EINVAL, "ramses_amd_godunov_fine_lowdim_f90 is the entry of NDIM=1 and NDIM=2 builds (got NDIM=%d)", ndim);
  if (p->nvar != ndim + 2) return fail(RAMSES_AMD_EUNSUPPORTED, "NDIM=%d device sweep: hydro variables only (NVAR=%d, got %d)", ndim, ndim + 2, p->nvar);
  if (ilevel < 1 || ilevel > 20 || ngrid < 1 || nbound < 0) return fail(RAMSES_AMD_EINVAL, "bad level / oct count");
  const int twotondim = 1 << ndim, nvh = ndim + 2;
  const long ncell = ncoarse + (long)twotondim * ngridmax;
  const long nx = (long)nloc[0] << ilevel, ny = ndim > 1 ? (long)nloc[1] << ilevel : 1;
  if ((long)ngrid * twotondim != nx * ny)
    return fail(RAMSES_AMD_EUNSUPPORTED, "level %d is not fully refined (%d octs, %ld x %ld cells): AMR levels of NDIM<3 runs stay the reference's", ilevel,
                ngrid, nx, ny);
  if (nx > 65536 || ny > 65536 || nx * ny > (1L << 26)) return fail(RAMSES_AMD_EUNSUPPORTED, "level too large for the host-assembled brick");
  ramses_amd_brick b;
  ramses_amd_brick_dense(&b, (int)nx, (int)ny, 1, 2);
  const int ng = 2;
  const long NX = nx + 2 * ng, NY = ny + 2 * ng, NZ = 1 + 2 * ng;
  if (b.pitch_y != NX || b.pitch_z != NX * NY || b.pitch_var != NX * NY * NZ) return fail(RAMSES_AMD_EINVAL, "unexpected brick pitches");
  static std::vector<double> hb;
  static std::vector<unsigned char> got;
  hb.assign((size_t)5 * b.pitch_var, 0.0);
  got.assign((size_t)NX * NY, 0);
  // the embedded variables: rho, rho u [, rho v], E  ->  rho, rho u, rho v, rho w, E
  int vmap[4];
  vmap[0] = 0; vmap[1] = 1;
  if (ndim == 2) { vmap[2] = 2; vmap[3] = 4; } else { vmap[2] = 4; vmap[3] = -1; }
  const double scale_l = (double)(1L << ilevel);
  long nghost_filled[2] = {0, 0};
  auto place = [&](int g, bool active) -> int {
    long o[2] = {0, 0};
    for (int d = 0; d < ndim; d++) {
      const double xc = (xg[(long)d * ngridmax + g - 1] - (double)skip[d]) * scale_l - 1.0;     // left cell of the oct
      const long r = std::lround(xc);
      if (std::fabs(xc - (double)r) > 1e-6) return fail(RAMSES_AMD_EINVAL, "oct %d of level %d does not sit on the level lattice", g, ilevel);
      o[d] = r;
    }
    for (int ind = 0; ind < twotondim; ind++) {
      const long ci = o[0] + (ind & 1), cj = ndim > 1 ? o[1] + ((ind >> 1) & 1) : 0;
      const bool inside = ci >= 0 && ci < nx && cj >= 0 && cj < ny;
      if (active && !inside) return fail(RAMSES_AMD_EINVAL, "active oct %d lies outside the box", g);
      if (ci < -ng || ci >= nx + ng || cj < -(ndim > 1 ? ng : 0) || cj >= ny + (ndim > 1 ? ng : 0)) continue;   // deeper boundary layers: not read
      const long at = (ci + ng) + NX * ((ndim > 1 ? cj + ng : ng) + NY * (long)ng);
      const long icell = ncoarse + (long)ind * ngridmax + g - 1;
      for (int v = 0; v < nvh; v++) hb[(size_t)vmap[v] * b.pitch_var + at] = uold[(size_t)v * ncell + icell];
      got[(size_t)((ci + ng) + NX * (ndim > 1 ? cj + ng : ng))] = 1;
      if (!inside) { if (ci < 0 || ci >= nx) nghost_filled[0]++; else nghost_filled[1]++; }
    }
    return 0;
  };
  for (int i = 0; i < ngrid; i++) if (int rc = place(igrid[i], true)) return rc;
  for (int i = 0; i < nbound; i++) if (int rc = place(igrid_bound[i], false)) return rc;
  // a direction is bounded when boundary octs filled its ghost cells (then all of them must be there), periodic otherwise
  bool bounded[2] = {nghost_filled[0] > 0, ndim > 1 && nghost_filled[1] > 0};
  int periodic_axes = 4;                                  // z is always a copy of the plane
  if (!bounded[0]) periodic_axes |= 1;
  if (ndim == 1 || !bounded[1]) periodic_axes |= 2;
  for (long j = 0; j < (ndim > 1 ? NY : 1); j++)
    for (long i = 0; i < NX; i++) {
      const bool gx = i < ng || i >= nx + ng, gy = ndim > 1 && (j < ng || j >= ny + ng);
      if (!gx && !gy) continue;
      // ghost cells of a bounded direction come from the boundary octs; those of a periodic one are filled on the device
      const bool need = (gx && bounded[0] && !(gy && !bounded[1])) || (gy && bounded[1] && !(gx && !bounded[0]));
      if (need && !got[(size_t)(i + NX * (ndim > 1 ? j : ng))])
        return fail(RAMSES_AMD_EUNSUPPORTED, "level %d: the boundary regions do not cover ghost cell (%ld,%ld) of the level's brick", ilevel, i - ng, j - ng);
    }
  {
    static bool said = false;
    if (!said) {
      said = true;
      printf(" ramses_amd: NDIM=%d: level %d (%ld x %ld cells) is swept on the device as a brick embedded in 3-D; x %s, y %s\n", ndim, ilevel, nx, ny,
             bounded[0] ? "between boundary octs" : "periodic", ndim == 1 ? "-" : (bounded[1] ? "between boundary octs" : "periodic"));
      fflush(stdout);
    }
  }
  if (int rc = resident_release("godunov_fine (NDIM<3 brick sweep)")) return rc;
  hipStream_t s = nullptr;
  HostCtx &H = g_host;
#define HCHK(call, what) do { hipError_t e_ = (call); if (e_ != hipSuccess) return hipfail(e_, what); } while (0)
  const size_t bytes = sizeof(double) * 5 * (size_t)b.pitch_var;
  HCHK(H.bold.ensure(bytes), "hipMalloc brick"); HCHK(H.bnew.ensure(bytes), "hipMalloc brick");
  HCHK(hipMemcpyAsync(H.bold.p, hb.data(), bytes, hipMemcpyHostToDevice, s), "H2D brick");
  // periodic directions, x first: a later direction copies the ghosts the earlier ones filled (corners)
  for (int axis = 0; axis < 3; axis++)
    if (periodic_axes & (1 << axis))
      if (int rc = ramses_amd_fill_ghosts_periodic(&b, H.bold.as<double>(), 5, 1 << axis, s)) return rc;
  ramses_amd_hydro_params q = *p;
  q.nvar = 5;
  if (int rc = ramses_amd_godunov_brick(&q, &b, H.bold.as<double>(), nullptr, H.bnew.as<double>(), dx, dt, s)) return rc;
  HCHK(hipMemcpyAsync(hb.data(), H.bnew.p, bytes, hipMemcpyDeviceToHost, s), "D2H brick");
  HCHK(hipStreamSynchronize(s), "sync");
#undef HCHK
  // unew(active cells) = uold + flux differences, as after the reference's set_unew + godunov_fine
  for (int i = 0; i < ngrid; i++) {
    const int g = igrid[i];
    long o[2] = {0, 0};
    for (int d = 0; d < ndim; d++) o[d] = std::lround((xg[(long)d * ngridmax + g - 1] - (double)skip[d]) * scale_l - 1.0);
    for (int ind = 0; ind < twotondim; ind++) {
      const long ci = o[0] + (ind & 1), cj = ndim > 1 ? o[1] + ((ind >> 1) & 1) : 0;
      const long at = (ci + ng) + NX * ((ndim > 1 ? cj + ng : ng) + NY * (long)ng);
      const long icell = ncoarse + (long)ind * ngridmax + g - 1;
      for (int v = 0; v < nvh; v++) unew[(size_t)v * ncell + icell] = hb[(size_t)vmap[v] * b.pitch_var + at];
    }
  }
  lowdim_register();
  if (ilevel < 32) g_lowdim_dev[ilevel]++;
  return 0;
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

}  // extern "C"

#include "warm.hpp"
RAMSES_AMD_TU_WARM(capi_host)
