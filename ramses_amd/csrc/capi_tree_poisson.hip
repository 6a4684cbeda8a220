// capi_tree_poisson.hip -- the Poisson solvers of AMR levels behind the C ABI (declared in include/ramses_amd.h): the
// multigrid routines the reference's own driver calls (ramses_amd_mgamr_*), the conjugate-gradient solve on one rank
// (ramses_amd_cg_solve_host) and under MPI (ramses_amd_cgmpi_*).  Split from capi.hip in round 4.
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

static int resident_release(const char *who) { return ramses_amd::capi_resident_release(who); }

extern "C" {

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

}  // extern "C"

#include "warm.hpp"
RAMSES_AMD_TU_WARM(capi_tree_poisson)
