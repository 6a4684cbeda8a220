// pois_amr.hip -- multigrid_fine on a partially refined (AMR) level, driver and per-solve setup on the device.
//
// The reference rebuilds, for every solve, a stack of multigrid levels under the AMR level on the host
// (poisson/multigrid_fine_commons.f90:25-296): first guess, mask, boundary-modified right-hand side, the
// lists of coarse octs (build_parent_comms_mg), the restricted masks, the scan flags -- and only then
// iterates.  Here all of it is device work on a tree that stays on the GPU between regrids:
//   make_initial_phi / interpol_phi   poisson/phi_fine_cg.f90:452-521, poisson/interpol_phi.f90:1-81   init_phi_kernel
//   make_fine_mask                    multigrid_fine_commons.f90:982-1035                              (mask = 1: periodic, one rank)
//   make_fine_bc_rhs                  multigrid_fine_commons.f90:1058-1159                             bc_rhs_kernel
//   build_parent_comms_mg             multigrid_fine_commons.f90:400-894 (single rank: stage 1)        hier_mark_kernel + lookup
//   restrict_mask_fine/coarse_reverse multigrid_fine_fine.f90:88-141, multigrid_fine_coarse.f90:105-160 mask_restrict_kernel
//   2*u4-1, allmasked                 multigrid_fine_commons.f90:103-170                               mask_convert_kernel
//   set_scan_flag_fine / _coarse      multigrid_fine_fine.f90:705-771, multigrid_fine_coarse.f90:892-985 scan_flag_kernel
//   multigrid_fine's iteration loop, recursive_multigrid_coarse  :176-282, :307-390                    host loop below, operators of mg_amr.hip
// The order of octs inside a coarse level's list is not the reference's (it is whatever order the octs
// are discovered in); no result depends on it: cells of one colour never read each other, a coarse cell
// receives its restriction from exactly one fine oct (octants added in the reference's order), the
// interpolation gathers.  The only ordered sum of a solve, the residual norm, runs over the fine level,
// whose list is the caller's.  Arithmetic and operation order are the reference's (-ffp-contract=off).
//
// The host passes phi / phi_old / rho as the reference's cell vectors; only the cells the solve reads
// (rho of the level, phi and phi_old of the level above) and writes (phi of the level) travel.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/ramses_amd.h"
#include "mg_amr_args.hpp"
#include "misc_args.hpp"

using namespace ramses_amd;

extern "C" int ramses_amd_set_error(int code, const char *msg);
static int failf(int code, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  return ramses_amd_set_error(code, buf);
}
#define HCHK(call, what) do { hipError_t e_ = (call); if (e_ != hipSuccess) return failf(RAMSES_AMD_EHIP, "%s: %s", what, hipGetErrorString(e_)); } while (0)

namespace {

struct Buf {
  void *p = nullptr;
  size_t cap = 0;
  hipError_t ensure(size_t bytes) {
    if (bytes <= cap && p) return hipSuccess;
    if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
    if (bytes == 0) bytes = 8;
    hipError_t e = hipMalloc(&p, bytes);
    if (e == hipSuccess) cap = bytes;
    return e;
  }
  template <class T> T *as() { return reinterpret_cast<T *>(p); }
};
struct PinBuf {     // page-locked staging area
  void *p = nullptr;
  size_t cap = 0;
  hipError_t ensure(size_t bytes) {
    if (bytes <= cap && p) return hipSuccess;
    if (p) { (void)hipHostFree(p); p = nullptr; cap = 0; }
    if (bytes == 0) bytes = 8;
    hipError_t e = hipHostMalloc(&p, bytes, hipHostMallocDefault);
    if (e == hipSuccess) cap = bytes;
    return e;
  }
  template <class T> T *as() { return reinterpret_cast<T *>(p); }
};

// same-level neighbour of cell c (1-based AMR index) in direction dir (-x,+x,-y,+y,-z,+z); 0 if its oct does
// not exist.  The single coarse cell of a periodic box is its own neighbour.
__device__ __forceinline__ int nbor_cell(int c, int dir, const MgAmrTree &T) {
  if (c <= T.ncoarse) return c;
  const int pos = (int)((c - T.ncoarse - 1) / T.ngridmax);
  const int g = (int)(c - T.ncoarse - (long)pos * T.ngridmax);
  const int axis = dir >> 1, up = dir & 1;
  const int bit = (pos >> axis) & 1;
  if (bit != up) return c + (up ? 1 : -1) * (int)((1 << axis) * T.ngridmax);
  const int nb = T.nbor[(long)dir * T.ngridmax + g - 1];
  const int g2 = T.son[nb - 1];
  if (g2 == 0) return 0;
  return (int)(T.ncoarse + (long)(pos ^ (1 << axis)) * T.ngridmax + g2);
}

// the cell at offset (d0,d1,d2), each in -1..1, from cell c, the way get3cubefather / get3cubepos find it
// (amr/nbors_utils.f90:5-194, 199-300): from c's oct to the neighbouring oct of the same level through
// son(nbor(oct, dir)), z first, then y, then x; 0 if an oct on that path does not exist
__device__ __forceinline__ int cell_at(int c, int d0, int d1, int d2, const MgAmrTree &T) {
  if (c <= T.ncoarse) return c;
  const int pos = (int)((c - T.ncoarse - 1) / T.ngridmax);
  int g = (int)(c - T.ncoarse - (long)pos * T.ngridmax);
  const int d[3] = {d0, d1, d2};
  int npos = pos;
#pragma unroll
  for (int a = 2; a >= 0; a--) {
    if (d[a] == 0) continue;
    const int b = (pos >> a) & 1;
    npos ^= 1 << a;
    int o = 0;
    if (d[a] < 0 && b == 0) o = -1;
    if (d[a] > 0 && b == 1) o = 1;
    if (o != 0 && g > 0) g = T.son[T.nbor[(long)(2 * a + (o > 0 ? 1 : 0)) * T.ngridmax + g - 1] - 1];
  }
  if (g <= 0) return 0;
  return (int)(T.ncoarse + (long)npos * T.ngridmax + g);
}

// ccc(ind_average, ind): which of the 27 father cells (1-based, x fastest) feed child octant ind
__device__ __constant__ int c_ccc[8][8] = {{1, 2, 4, 5, 10, 11, 13, 14},     {3, 2, 6, 5, 12, 11, 15, 14},
                                           {7, 8, 4, 5, 16, 17, 13, 14},     {9, 8, 6, 5, 18, 17, 15, 14},
                                           {19, 20, 22, 23, 10, 11, 13, 14}, {21, 20, 24, 23, 12, 11, 15, 14},
                                           {25, 26, 22, 23, 16, 17, 13, 14}, {27, 26, 24, 23, 18, 17, 15, 14}};

// interpol_phi for ONE child octant of father cell fc: CIC in space over 8 of the 27 cells around fc, linear
// extrapolation in time (tfrac); a cell of the cube whose oct does not exist is replaced by fc itself
__device__ __forceinline__ double interpol_phi_child(int fc, int child, const double *phi, const double *phi_old, double tfrac,
                                                     const MgAmrTree &T) {
  const double aa = 1.0 / 64.0, bb = 3 * aa, cc = 9 * aa, dd = 27 * aa;
  const double bbbb[8] = {aa, bb, bb, cc, bb, cc, cc, dd};
  double acc = 0.0;
#pragma unroll 1
  for (int av = 0; av < 8; av++) {
    const int t = c_ccc[child][av] - 1;
    int idx = cell_at(fc, t % 3 - 1, (t / 3) % 3 - 1, t / 9 - 1, T);
    if (idx == 0) idx = fc;
    const double p = phi[idx - 1], po = phi_old[idx - 1];
    const double add = bbbb[av] * (p + (p - po) * tfrac);
    acc = acc + add;
  }
  return acc;
}

struct FineArgs {
  MgAmrLevel L;
  MgAmrTree T;
  const double *phi, *phi_old, *rho;   // AMR cell vectors on the device
  double tfrac, fourpi, rho_tot, oneoverdx2;
  int interp;                          // 1: first guess interpolated from the level above, 0: zero
};

// first guess (make_initial_phi / make_multipole_phi in a periodic box) and the mask of an all-active level
__global__ __launch_bounds__(256) void init_phi_kernel(FineArgs A) {
  const long total = 8L * A.L.ngrid;
  for (long c = (long)blockIdx.x * blockDim.x + threadIdx.x; c < total; c += (long)gridDim.x * blockDim.x) {
    const int i = (int)(c % A.L.ngrid), ind = (int)(c / A.L.ngrid);
    double v = 0.0;
    if (A.interp) v = interpol_phi_child(A.T.father[A.L.igrid[i] - 1], ind, A.phi, A.phi_old, A.tfrac, A.T);
    A.L.u1[c] = v;
    A.L.u4[c] = 1.0;
  }
}

// make_fine_bc_rhs: u2 = fourpi (rho - rho_tot) - sum over faces without a neighbour oct of 2/dx^2 phi_b
__global__ __launch_bounds__(256) void bc_rhs_kernel(FineArgs A) {
  const long total = 8L * A.L.ngrid;
  for (long c = (long)blockIdx.x * blockDim.x + threadIdx.x; c < total; c += (long)gridDim.x * blockDim.x) {
    const int i = (int)(c % A.L.ngrid), ind = (int)(c / A.L.ngrid);
    const int g = A.L.igrid[i];
    const long cell = A.T.ncoarse + (long)ind * A.T.ngridmax + g;       // 1-based
    double rhs = A.fourpi * (A.rho[cell - 1] - A.rho_tot);
    const double m = A.L.u4[c];
    if (m > 0.0) {
#pragma unroll 1
      for (int axis = 0; axis < 3; axis++)
#pragma unroll 1
        for (int up = 0; up < 2; up++) {
          const int bit = (ind >> axis) & 1;
          if (bit != up) continue;                                      // neighbour inside the oct: active
          const int nb = A.T.nbor[(long)(2 * axis + up) * A.T.ngridmax + g - 1];
          const int g2 = A.T.son[nb - 1];
          double nb_mask, nb_phi;
          if (g2 == 0) {
            nb_mask = -1.0;
            nb_phi = interpol_phi_child(nb, ind ^ (1 << axis), A.phi, A.phi_old, A.tfrac, A.T);
          } else {
            const int j = A.T.lookup[g2 - 1];
            if (j <= 0 || j > A.L.ngrid || A.L.igrid[j - 1] != g2) continue;   // (cannot happen: one rank, no walls)
            const long n = (long)(ind ^ (1 << axis)) * A.L.ngrid + (j - 1);
            nb_mask = A.L.u4[n];
            if (nb_mask > 0.0) continue;
            nb_phi = A.L.u1[n];
          }
          const double w = nb_mask / (nb_mask - m);
          const double phi_b = ((1.0 - w) * nb_phi + w * A.L.u1[c]);
          rhs = rhs - 2.0 * A.oneoverdx2 * phi_b;
        }
    }
    A.L.u2[c] = rhs;
  }
}

// build_parent_comms_mg, stage 1: the octs holding the 3^3 father cells around every oct of level F join the
// level below (first claim wins; list[] receives them in claim order)
__global__ __launch_bounds__(256) void hier_mark_kernel(const int *igrid, int ngrid, MgAmrTree T, int *lookup, int *list, int cap, int *count) {
  const long total = 27L * ngrid;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    const int i = (int)(t / 27), k = (int)(t % 27);
    const int f0 = T.father[igrid[i] - 1];
    const int fc = cell_at(f0, k % 3 - 1, (k / 3) % 3 - 1, k / 9 - 1, T);
    if (fc <= T.ncoarse) continue;                                    // (level 1 octs have no level below)
    const int pos = (int)((fc - T.ncoarse - 1) / T.ngridmax);
    const int g = (int)(fc - T.ncoarse - (long)pos * T.ngridmax);
    if (atomicCAS(&lookup[g - 1], 0, -1) == 0) {
      const int idx = atomicAdd(count, 1);
      if (idx < cap) list[idx] = g;
    }
  }
}
__global__ void set_lookup_kernel(const int *igrid, int ngrid, int *lookup, int value_or_index) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < ngrid) lookup[igrid[i] - 1] = value_or_index < 0 ? i + 1 : value_or_index;
}

// restrict_mask_*_reverse: volume fraction (1+mask)/2 of the 8 children, added in octant order
__global__ __launch_bounds__(256) void mask_restrict_kernel(MgAmrLevel F, MgAmrLevel C, MgAmrTree T) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < F.ngrid; i += gridDim.x * blockDim.x) {
    const int g = F.igrid[i];
    const int fc = T.father[g - 1];
    const int ind_c = (int)((fc - T.ncoarse - 1) / T.ngridmax);
    const int g_c = (int)(fc - T.ncoarse - (long)ind_c * T.ngridmax);
    const int j = T.lookup[g_c - 1];
    if (j <= 0 || j > C.ngrid || C.igrid[j - 1] != g_c) continue;
    double acc = 0.0;
#pragma unroll
    for (int ind = 0; ind < 8; ind++) acc = acc + (1.0 + F.u4[(long)ind * F.ngrid + i]) / 2 / 8.0;
    C.u4[(long)ind_c * C.ngrid + (j - 1)] = acc;
  }
}
// volume fraction -> mask value; any[0] = 1 if some cell of the level stays unmasked
__global__ __launch_bounds__(256) void mask_convert_kernel(MgAmrLevel C, int *any) {
  const long total = 8L * C.ngrid;
  int mine = 0;
  for (long c = (long)blockIdx.x * blockDim.x + threadIdx.x; c < total; c += (long)gridDim.x * blockDim.x) {
    const double m = 2 * C.u4[c] - 1.0;
    C.u4[c] = m;
    if (m > 0.0) mine = 1;
  }
  if (mine) atomicOr(any, 1);
}

// neighbour of cell (ind, oct position i): >= 0 cell of the level's layout, -1 no such oct in the level
__device__ __forceinline__ long lvl_nbr(const MgAmrLevel &L, const MgAmrTree &T, int ind, int i, int axis, int up) {
  const int bit = (ind >> axis) & 1;
  const int jnd = ind ^ (1 << axis);
  if (bit != up) return (long)jnd * L.ngrid + i;
  const int g = L.igrid[i];
  const int nb = T.nbor[(long)(2 * axis + up) * T.ngridmax + g - 1];
  const int g2 = T.son[nb - 1];
  if (g2 == 0) return -1;
  const int j = T.lookup[g2 - 1];
  if (j <= 0 || j > L.ngrid || L.igrid[j - 1] != g2) return -1;
  return (long)jnd * L.ngrid + (j - 1);
}
// set_scan_flag_*: 0 for a cell with mask 1 whose six neighbours exist and are unmasked, else 1
__global__ __launch_bounds__(256) void scan_flag_kernel(MgAmrLevel L, MgAmrTree T, int *scan) {
  const long total = 8L * L.ngrid;
  for (long c = (long)blockIdx.x * blockDim.x + threadIdx.x; c < total; c += (long)gridDim.x * blockDim.x) {
    const int i = (int)(c % L.ngrid), ind = (int)(c / L.ngrid);
    int flag = 1;
    if (L.u4[c] == 1.0) {
      flag = 0;
      for (int axis = 0; axis < 3 && !flag; axis++)
        for (int up = 0; up < 2 && !flag; up++) {
          const long n = lvl_nbr(L, T, ind, i, axis, up);
          if (n < 0) flag = 1;
          else if (L.u4[n] <= 0.0) flag = 1;
        }
    }
    scan[c] = flag;
  }
}

// gradient_phi (poisson/force_fine.f90:199-324): f_d = a (phi(-1) - phi(+1)) - b (phi(-2) - phi(+2)); a value in a
// neighbouring oct that does not exist is interpolated from the level above (interpol_phi of the neighbouring
// father cell).  out = packed [3][8*ngrid]; leaf[c] = 1 where the cell is not refined (for the diagnostics)
struct ForceArgs {
  MgAmrLevel L;
  MgAmrTree T;
  const double *phi, *phi_old;     // AMR cell vectors: phi of the level (and of the level above, with phi_old, if interp)
  double tfrac, a, b;
  int interp;
  double *out;
  int *leaf;
};
__device__ __forceinline__ double force_phi_at(const ForceArgs &A, int g_nb, int nb_father, int octant) {
  if (g_nb > 0) return A.phi[A.T.ncoarse + (long)octant * A.T.ngridmax + g_nb - 1];
  return interpol_phi_child(nb_father, octant, A.phi, A.phi_old, A.tfrac, A.T);
}
__global__ __launch_bounds__(256) void force_kernel(ForceArgs A) {
  const long total = 8L * A.L.ngrid;
  for (long c = (long)blockIdx.x * blockDim.x + threadIdx.x; c < total; c += (long)gridDim.x * blockDim.x) {
    const int i = (int)(c % A.L.ngrid), ind = (int)(c / A.L.ngrid);
    const int g = A.L.igrid[i];
    A.leaf[c] = A.T.son[A.T.ncoarse + (long)ind * A.T.ngridmax + g - 1] == 0;
#pragma unroll 1
    for (int d = 0; d < 3; d++) {
      const int nbl = A.T.nbor[(long)(2 * d) * A.T.ngridmax + g - 1], nbr = A.T.nbor[(long)(2 * d + 1) * A.T.ngridmax + g - 1];
      const int gl = A.T.son[nbl - 1], gr = A.T.son[nbr - 1];
      const int bit = (ind >> d) & 1, jnd = ind ^ (1 << d);
      const double phi1 = bit ? A.phi[A.T.ncoarse + (long)jnd * A.T.ngridmax + g - 1] : force_phi_at(A, gl, nbl, jnd);
      const double phi2 = bit ? force_phi_at(A, gr, nbr, jnd) : A.phi[A.T.ncoarse + (long)jnd * A.T.ngridmax + g - 1];
      const double phi3 = force_phi_at(A, gl, nbl, ind);
      const double phi4 = force_phi_at(A, gr, nbr, ind);
      A.out[(long)d * total + c] = A.a * (phi1 - phi2) - A.b * (phi3 - phi4);
    }
  }
}
// packed level values of 3 components -> AMR cell vectors f(1:ncell,1:3) on the device
__global__ void vec3_scatter_kernel(double *vec, const double *in, const int *igrid, int ngrid, long ncoarse, long ngridmax, long ncell) {
  const long total = 8L * ngrid;
  for (long c = (long)blockIdx.x * blockDim.x + threadIdx.x; c < total; c += (long)gridDim.x * blockDim.x) {
    const long cell = ncoarse + (long)(c / ngrid) * ngridmax + igrid[c % ngrid] - 1;
    for (int d = 0; d < 3; d++) vec[(long)d * ncell + cell] = in[(long)d * total + c];
  }
}
__global__ void vec_gather_kernel(const double *vec, double *out, const int *igrid, int ngrid, long ncoarse, long ngridmax) {
  const long total = 8L * ngrid;
  for (long c = (long)blockIdx.x * blockDim.x + threadIdx.x; c < total; c += (long)gridDim.x * blockDim.x)
    out[c] = vec[ncoarse + (long)(c / ngrid) * ngridmax + igrid[c % ngrid] - 1];
}

// cell vector (AMR layout, device) <- packed level values (ind*ngrid + i) and back
__global__ void vec_scatter_kernel(double *vec, const double *in, const int *igrid, int ngrid, long ncoarse, long ngridmax) {
  const long total = 8L * ngrid;
  for (long c = (long)blockIdx.x * blockDim.x + threadIdx.x; c < total; c += (long)gridDim.x * blockDim.x)
    vec[ncoarse + (long)(c / ngrid) * ngridmax + igrid[c % ngrid] - 1] = in[c];
}

inline int grid_for(long work, int cap = 4096) {
  long g = (work + 255) / 256;
  if (g < 1) g = 1;
  if (g > cap) g = cap;
  return (int)g;
}

struct Level {
  int ngrid = 0;
  Buf igrid, u1, u2, u3, u4, scan;
  MgAmrLevel view() {
    MgAmrLevel L;
    L.ngrid = ngrid; L.nact = ngrid; L.igrid = igrid.as<int>();
    L.u1 = u1.as<double>(); L.u2 = u2.as<double>(); L.u3 = u3.as<double>(); L.u4 = u4.as<double>();
    L.scan = scan.as<int>();
    return L;
  }
  hipError_t ensure(int n) {
    const size_t nn = 8 * (size_t)(n > 0 ? n : 1);
    hipError_t e;
    if ((e = u1.ensure(sizeof(double) * nn)) != hipSuccess) return e;
    if ((e = u2.ensure(sizeof(double) * nn)) != hipSuccess) return e;
    if ((e = u3.ensure(sizeof(double) * nn)) != hipSuccess) return e;
    if ((e = u4.ensure(sizeof(double) * nn)) != hipSuccess) return e;
    return scan.ensure(sizeof(int) * nn);
  }
};

struct PoisAmr {
  // the tree, as of `epoch`
  bool tree_valid = false;
  int epoch = -1;
  long ncoarse = 0, ngridmax = 0, ncell = 0;
  Buf son, nbor, father, lookup;
  // the reference's cell vectors (AMR layout)
  Buf phi, phi_old, rho, f;
  // what the device copies hold: phi and rho of level `have_level` (ngrid `have_ngrid`), and, if have_above, phi and
  // phi_old of the level above it -- left there by the last multigrid solve for force_fine
  int have_level = 0, have_ngrid = 0, have_epoch = -1;
  bool have_above = false;
  Buf fpack, leaf, diag;
  Buf igrid_c;                      // octs of the level above (whose phi feeds the interpolation)
  Level lev[32];
  Buf count, any, partial, norm, pack;
  PinBuf stage;
  int levelmin_mg = 1;
  MgAmrTree tree() {
    MgAmrTree T;
    T.son = son.as<int>(); T.nbor = nbor.as<int>(); T.father = father.as<int>(); T.lookup = lookup.as<int>();
    T.ncoarse = ncoarse; T.ngridmax = ngridmax;
    return T;
  }
};
PoisAmr g_pa;

// recursive_multigrid_coarse (multigrid_fine_commons.f90:307-390)
int coarse_cycle(PoisAmr &P, int level, int safe, int ngs_coarse, int ncycles_safe) {
  hipStream_t s = nullptr;
  MgAmrLevel L = P.lev[level].view();
  const MgAmrTree T = P.tree();
  const double dx = std::ldexp(1.0, -level);
  if (level <= P.levelmin_mg) {
    for (int i = 0; i < 2 * ngs_coarse; i++) {
      HCHK(mgamr_launch_gs(L, T, 0, safe, dx * dx, s), "gs");
      HCHK(mgamr_launch_gs(L, T, 1, safe, dx * dx, s), "gs");
    }
    return 0;
  }
  const int ncycle = safe ? ncycles_safe : 1;
  for (int icycle = 0; icycle < ncycle; icycle++) {
    for (int i = 0; i < ngs_coarse; i++) {
      HCHK(mgamr_launch_gs(L, T, 0, safe, dx * dx, s), "gs");
      HCHK(mgamr_launch_gs(L, T, 1, safe, dx * dx, s), "gs");
    }
    HCHK(mgamr_launch_residual(L, T, 1.0 / (dx * dx), s), "residual");
    HCHK(mgamr_launch_restrict(L, P.lev[level - 1].view(), T, s), "restrict");
    if (int rc = coarse_cycle(P, level - 1, safe, ngs_coarse, ncycles_safe)) return rc;
    HCHK(mgamr_launch_interp(L, P.lev[level - 1].view(), T, s), "interp");
    for (int i = 0; i < ngs_coarse; i++) {
      HCHK(mgamr_launch_gs(L, T, 0, safe, dx * dx, s), "gs");
      HCHK(mgamr_launch_gs(L, T, 1, safe, dx * dx, s), "gs");
    }
  }
  return 0;
}

}  // namespace

extern "C" {

// the tree arrays son(1:ncell), nbor(1:ngridmax,1:6), father(1:ngridmax); sent only when `epoch` (a counter the
// caller advances whenever refine_fine / load_balance may have changed the tree) differs from the cached one
int ramses_amd_poisamr_tree(int epoch, int64_t ngridmax, int64_t ncoarse, const int *son, const int *nbor, const int *father) {
  if (!son || !nbor || !father) return failf(RAMSES_AMD_EINVAL, "NULL argument");
  if (ngridmax < 1 || ncoarse != 1) return failf(RAMSES_AMD_EUNSUPPORTED, "the device AMR multigrid driver covers a box of one coarse cell (nx=ny=nz=1)");
  PoisAmr &P = g_pa;
  if (P.tree_valid && P.epoch == epoch && P.ngridmax == ngridmax && P.ncoarse == ncoarse) return 0;
  P.tree_valid = false;
  P.ncoarse = ncoarse; P.ngridmax = ngridmax; P.ncell = ncoarse + 8 * ngridmax;
  const bool fresh = P.lookup.cap < sizeof(int) * (size_t)ngridmax;
  HCHK(P.son.ensure(sizeof(int) * (size_t)P.ncell), "hipMalloc son");
  HCHK(P.nbor.ensure(sizeof(int) * 6 * (size_t)ngridmax), "hipMalloc nbor");
  HCHK(P.father.ensure(sizeof(int) * (size_t)ngridmax), "hipMalloc father");
  HCHK(P.lookup.ensure(sizeof(int) * (size_t)ngridmax), "hipMalloc lookup");
  if (fresh) HCHK(hipMemsetAsync(P.lookup.p, 0, sizeof(int) * (size_t)ngridmax, nullptr), "memset lookup");
  HCHK(hipMemcpyAsync(P.son.p, son, sizeof(int) * (size_t)P.ncell, hipMemcpyHostToDevice, nullptr), "H2D son");
  HCHK(hipMemcpyAsync(P.nbor.p, nbor, sizeof(int) * 6 * (size_t)ngridmax, hipMemcpyHostToDevice, nullptr), "H2D nbor");
  HCHK(hipMemcpyAsync(P.father.p, father, sizeof(int) * (size_t)ngridmax, hipMemcpyHostToDevice, nullptr), "H2D father");
  HCHK(hipStreamSynchronize(nullptr), "sync");
  P.epoch = epoch;
  P.tree_valid = true;
  return 0;
}

// multigrid_fine(ilevel) on an AMR level of a periodic single-rank run.
//   igrid[ngrid]       active(ilevel)%igrid;  igrid_c[ngrid_c]  active(ilevel-1)%igrid (ignored unless interp)
//   phi, phi_old, rho  the reference's cell vectors (host).  Read: rho on the level; phi, phi_old on the level above
//                      (interp = 1: ilevel > levelmin).  Written: phi on the level.
//   safe_mode          in/out, the level's safe_mode flag;  iters, err: what the reference prints
int ramses_amd_poisamr_multigrid(int ilevel, int ngrid, const int *igrid, int ngrid_c, const int *igrid_c, double *phi,
                                 const double *phi_old, const double *rho, int *flag2, double rho_tot, double fourpi, double tfrac,
                                 int interp, double epsilon, int ngs_fine, int ngs_coarse, int ncycles_coarse_safe, int *safe_mode,
                                 int *iters, double *err_out) {
  PoisAmr &P = g_pa;
  if (!P.tree_valid) return failf(RAMSES_AMD_EINVAL, "poisamr_multigrid: no tree (ramses_amd_poisamr_tree)");
  if (!igrid || !phi || !phi_old || !rho || !flag2 || !safe_mode || !iters || !err_out) return failf(RAMSES_AMD_EINVAL, "NULL argument");
  if (ilevel < 2 || ilevel > 30 || ngrid < 1 || ngrid > P.ngridmax) return failf(RAMSES_AMD_EINVAL, "bad level %d / ngrid %d", ilevel, ngrid);
  if (interp && (!igrid_c || ngrid_c < 1)) return failf(RAMSES_AMD_EINVAL, "the level above is empty");
  const int MAXITER = 10;
  const double SAFE_FACTOR = 0.5;
  hipStream_t s = nullptr;
  const long ncoarse = P.ncoarse, ngridmax = P.ngridmax;
  const size_t vb = sizeof(double) * (size_t)P.ncell;
  HCHK(P.phi.ensure(vb), "hipMalloc phi"); HCHK(P.phi_old.ensure(vb), "hipMalloc phi_old"); HCHK(P.rho.ensure(vb), "hipMalloc rho");
  HCHK(P.count.ensure(sizeof(int) * 64), "hipMalloc"); HCHK(P.any.ensure(sizeof(int) * 64), "hipMalloc");
  HCHK(P.partial.ensure(sizeof(double) * 1024), "hipMalloc"); HCHK(P.norm.ensure(sizeof(double)), "hipMalloc");

  // ---- the cells the solve reads: host gather into the pinned area, one copy, device scatter
  const long nf = 8L * ngrid, nc = interp ? 8L * ngrid_c : 0;
  HCHK(P.stage.ensure(sizeof(double) * (size_t)(nf + 2 * nc) + sizeof(int) * (size_t)(ngrid + (interp ? ngrid_c : 0))), "hipHostMalloc");
  HCHK(P.pack.ensure(sizeof(double) * (size_t)(nf + 2 * nc)), "hipMalloc pack");
  double *hs = P.stage.as<double>();
  for (int ind = 0; ind < 8; ind++) {
    const double *src = rho + ncoarse + (size_t)ind * ngridmax - 1;
    double *dst = hs + (size_t)ind * ngrid;
    for (int i = 0; i < ngrid; i++) dst[i] = src[igrid[i]];
  }
  if (interp)
    for (int ind = 0; ind < 8; ind++) {
      const double *s1 = phi + ncoarse + (size_t)ind * ngridmax - 1, *s2 = phi_old + ncoarse + (size_t)ind * ngridmax - 1;
      double *d1 = hs + nf + (size_t)ind * ngrid_c, *d2 = hs + nf + nc + (size_t)ind * ngrid_c;
      for (int i = 0; i < ngrid_c; i++) { d1[i] = s1[igrid_c[i]]; d2[i] = s2[igrid_c[i]]; }
    }
  Level &F = P.lev[ilevel];
  F.ngrid = ngrid;
  HCHK(F.igrid.ensure(sizeof(int) * (size_t)ngrid), "hipMalloc igrid"); HCHK(F.ensure(ngrid), "hipMalloc level");
  HCHK(hipMemcpyAsync(F.igrid.p, igrid, sizeof(int) * (size_t)ngrid, hipMemcpyHostToDevice, s), "H2D igrid");
  HCHK(hipMemcpyAsync(P.pack.p, hs, sizeof(double) * (size_t)(nf + 2 * nc), hipMemcpyHostToDevice, s), "H2D level data");
  hipLaunchKernelGGL(vec_scatter_kernel, dim3(grid_for(nf)), dim3(256), 0, s, P.rho.as<double>(), P.pack.as<double>(), F.igrid.as<int>(), ngrid, ncoarse, ngridmax);
  if (interp) {
    HCHK(P.igrid_c.ensure(sizeof(int) * (size_t)ngrid_c), "hipMalloc igrid_c");
    HCHK(hipMemcpyAsync(P.igrid_c.p, igrid_c, sizeof(int) * (size_t)ngrid_c, hipMemcpyHostToDevice, s), "H2D igrid_c");
    hipLaunchKernelGGL(vec_scatter_kernel, dim3(grid_for(nc)), dim3(256), 0, s, P.phi.as<double>(), P.pack.as<double>() + nf, P.igrid_c.as<int>(), ngrid_c, ncoarse, ngridmax);
    hipLaunchKernelGGL(vec_scatter_kernel, dim3(grid_for(nc)), dim3(256), 0, s, P.phi_old.as<double>(), P.pack.as<double>() + nf + nc, P.igrid_c.as<int>(), ngrid_c, ncoarse, ngridmax);
  }
  HCHK(hipGetLastError(), "scatter launch");

  // ---- the levels of the solve: the AMR level, then the octs around its fathers, and so on up to level 1
  const MgAmrTree T = P.tree();
  hipLaunchKernelGGL(set_lookup_kernel, dim3((ngrid + 255) / 256), dim3(256), 0, s, F.igrid.as<int>(), ngrid, P.lookup.as<int>(), -1);
  HCHK(hipMemsetAsync(P.count.p, 0, sizeof(int) * 64, s), "memset"); HCHK(hipMemsetAsync(P.any.p, 0, sizeof(int) * 64, s), "memset");
  for (int l = ilevel - 1; l >= 1; l--) {
    Level &Fl = P.lev[l + 1], &C = P.lev[l];
    long cap = 8L * Fl.ngrid;
    const long full = 1L << (3 * (l - 1) > 40 ? 40 : 3 * (l - 1));
    if (cap > full) cap = full;
    if (cap > ngridmax) cap = ngridmax;
    HCHK(C.igrid.ensure(sizeof(int) * (size_t)cap), "hipMalloc igrid");
    hipLaunchKernelGGL(hier_mark_kernel, dim3(grid_for(27L * Fl.ngrid)), dim3(256), 0, s, Fl.igrid.as<int>(), Fl.ngrid, T, P.lookup.as<int>(),
                       C.igrid.as<int>(), (int)cap, P.count.as<int>() + l);
    HCHK(hipGetLastError(), "hier_mark launch");
    int n = 0;
    HCHK(hipMemcpyAsync(&n, P.count.as<int>() + l, sizeof(int), hipMemcpyDeviceToHost, s), "D2H count");
    HCHK(hipStreamSynchronize(s), "sync");
    if (n < 1 || n > cap) return failf(RAMSES_AMD_EINVAL, "multigrid level %d: %d octs found (capacity %ld): tree inconsistent", l, n, cap);
    C.ngrid = n;
    HCHK(C.ensure(n), "hipMalloc level");
    hipLaunchKernelGGL(set_lookup_kernel, dim3((n + 255) / 256), dim3(256), 0, s, C.igrid.as<int>(), n, P.lookup.as<int>(), -1);
    HCHK(hipMemsetAsync(C.u4.p, 0, sizeof(double) * 8 * (size_t)n, s), "memset mask");
  }

  // ---- first guess, mask, boundary-modified right-hand side on the AMR level
  FineArgs A;
  A.L = F.view(); A.T = T;
  A.phi = P.phi.as<double>(); A.phi_old = P.phi_old.as<double>(); A.rho = P.rho.as<double>();
  A.tfrac = tfrac; A.fourpi = fourpi; A.rho_tot = rho_tot;
  const double dxf = std::ldexp(1.0, -ilevel);
  A.oneoverdx2 = 1.0 / (dxf * dxf);
  A.interp = interp ? 1 : 0;
  hipLaunchKernelGGL(init_phi_kernel, dim3(grid_for(nf)), dim3(256), 0, s, A);
  hipLaunchKernelGGL(bc_rhs_kernel, dim3(grid_for(nf)), dim3(256), 0, s, A);
  HCHK(hipGetLastError(), "setup launch");

  // ---- masks of the coarse levels, levelmin_mg, scan flags
  for (int l = ilevel - 1; l >= 1; l--) {
    hipLaunchKernelGGL(mask_restrict_kernel, dim3(grid_for(P.lev[l + 1].ngrid)), dim3(256), 0, s, P.lev[l + 1].view(), P.lev[l].view(), T);
    hipLaunchKernelGGL(mask_convert_kernel, dim3(grid_for(8L * P.lev[l].ngrid, 1024)), dim3(256), 0, s, P.lev[l].view(), P.any.as<int>() + l);
  }
  HCHK(hipGetLastError(), "mask launch");
  int any[32] = {0};
  HCHK(hipMemcpyAsync(any, P.any.p, sizeof(int) * 32, hipMemcpyDeviceToHost, s), "D2H mask state");
  HCHK(hipStreamSynchronize(s), "sync");
  // (multigrid_fine_commons.f90:96-170: the first fully masked level, counted from the top, ends the stack)
  if (!any[ilevel - 1]) {
    P.levelmin_mg = ilevel;
  } else {
    P.levelmin_mg = 1;
    for (int ifine = ilevel - 1; ifine >= 2; ifine--)
      if (!any[ifine - 1]) { P.levelmin_mg = ifine; break; }
  }
  for (int l = ilevel; l >= P.levelmin_mg && l >= 1; l--) {
    Level &Lv = P.lev[l];
    hipLaunchKernelGGL(scan_flag_kernel, dim3(grid_for(8L * Lv.ngrid)), dim3(256), 0, s, Lv.view(), T, Lv.scan.as<int>());
  }
  HCHK(hipGetLastError(), "scan flag launch");
  // The reference keeps the fine level's scan flag in its work array flag2 as flag2 += ngridmax*scan and tests
  // flag2/ngridmax == 0 (multigrid_fine_fine.f90:190,388,763-768); what flag2 held before counts: it is only reset when
  // > ngridmax or < 0, so a cell flagged in the level's previous solve (flag2 == ngridmax exactly) keeps taking the
  // neighbour-scanning branch -- same value, other summation order -- until some routine resets flag2 (flag_fine does,
  // below nlevelmax).  build_parent_comms_mg also parks each coarse level's oct list in flag2(1:n) (:476).  Both are
  // reproduced on the host array (the parked values are this driver's lists: any 0 < value < ngridmax acts alike).
  {
    std::vector<int> tmp;
    for (int l = ilevel - 1; l >= 1; l--) {
      const int n = P.lev[l].ngrid;
      tmp.resize(n);
      HCHK(hipMemcpy(tmp.data(), P.lev[l].igrid.p, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost), "D2H list");
      for (int i = 0; i < n; i++) flag2[i] = tmp[i];
    }
    tmp.resize(nf);
    HCHK(hipMemcpy(tmp.data(), F.scan.p, sizeof(int) * (size_t)nf, hipMemcpyDeviceToHost), "D2H scan");
    for (int ind = 0; ind < 8; ind++)
      for (int i = 0; i < ngrid; i++) {
        int *fl = flag2 + ncoarse + (size_t)ind * ngridmax + igrid[i] - 1;
        int v = *fl;
        if (v > ngridmax || v < 0) v = 0;
        v += (int)ngridmax * tmp[(size_t)ind * ngrid + i];
        *fl = v;
        tmp[(size_t)ind * ngrid + i] = (v / (int)ngridmax) != 0;
      }
    HCHK(hipMemcpy(F.scan.p, tmp.data(), sizeof(int) * (size_t)nf, hipMemcpyHostToDevice), "H2D scan");
  }

  // ---- the iteration (multigrid_fine_commons.f90:176-282)
  MgAmrLevel L = F.view();
  const double dx2 = dxf * dxf;
  int safe = *safe_mode ? 1 : 0;
  const bool trace = getenv("RAMSES_AMD_MG_TRACE") != nullptr;   // debugging aid: residual norms with all digits
  int iter = 0;
  double err = 1.0, last_err, i_res_norm2 = 0.0, res_norm2 = 0.0;
  for (;;) {
    iter++;
    for (int i = 0; i < ngs_fine; i++) {
      HCHK(mgamr_launch_gs(L, T, 0, safe, dx2, s), "gs");
      HCHK(mgamr_launch_gs(L, T, 1, safe, dx2, s), "gs");
    }
    HCHK(mgamr_launch_residual(L, T, A.oneoverdx2, s), "residual");
    if (iter == 1) {
      HCHK(mgamr_launch_norm(L, dx2 * dxf, P.partial.as<double>(), P.norm.as<double>(), s), "norm");
      HCHK(hipMemcpyAsync(&i_res_norm2, P.norm.p, sizeof(double), hipMemcpyDeviceToHost, s), "D2H norm");
    }
    HCHK(mgamr_launch_restrict(L, P.lev[ilevel - 1].view(), T, s), "restrict");
    if (int rc = coarse_cycle(P, ilevel - 1, safe, ngs_coarse, ncycles_coarse_safe)) return rc;
    HCHK(mgamr_launch_interp(L, P.lev[ilevel - 1].view(), T, s), "interp");
    for (int i = 0; i < ngs_fine; i++) {
      HCHK(mgamr_launch_gs(L, T, 0, safe, dx2, s), "gs");
      HCHK(mgamr_launch_gs(L, T, 1, safe, dx2, s), "gs");
    }
    HCHK(mgamr_launch_residual(L, T, A.oneoverdx2, s), "residual");
    HCHK(mgamr_launch_norm(L, dx2 * dxf, P.partial.as<double>(), P.norm.as<double>(), s), "norm");
    HCHK(hipMemcpyAsync(&res_norm2, P.norm.p, sizeof(double), hipMemcpyDeviceToHost, s), "D2H norm");
    HCHK(hipStreamSynchronize(s), "sync");
    if (trace) fprintf(stderr, "ramses_amd: AMR multigrid level %d iteration %d: |r|^2 = %.17e (first %.17e), safe = %d\n", ilevel, iter, res_norm2, i_res_norm2, safe);
    last_err = err;
    err = std::sqrt(res_norm2 / (i_res_norm2 + 1e-20 * (rho_tot * rho_tot)));
    if (err < epsilon || iter >= MAXITER) break;
    if (err > last_err * SAFE_FACTOR && !safe) safe = 1;
  }
  *safe_mode = safe;
  *iters = iter;
  *err_out = err;

  // ---- phi of the level back; the levels leave the lookup table (cleanup_mg_level)
  HCHK(hipMemcpyAsync(hs, F.u1.p, sizeof(double) * (size_t)nf, hipMemcpyDeviceToHost, s), "D2H phi");
  hipLaunchKernelGGL(vec_scatter_kernel, dim3(grid_for(nf)), dim3(256), 0, s, P.phi.as<double>(), F.u1.as<double>(), F.igrid.as<int>(), ngrid, ncoarse, ngridmax);
  for (int l = ilevel; l >= 1; l--)
    hipLaunchKernelGGL(set_lookup_kernel, dim3((P.lev[l].ngrid + 255) / 256), dim3(256), 0, s, P.lev[l].igrid.as<int>(), P.lev[l].ngrid, P.lookup.as<int>(), 0);
  HCHK(hipGetLastError(), "cleanup launch");
  HCHK(hipStreamSynchronize(s), "sync");
  for (int ind = 0; ind < 8; ind++) {
    double *dst = phi + ncoarse + (size_t)ind * ngridmax - 1;
    const double *src = hs + (size_t)ind * ngrid;
    for (int i = 0; i < ngrid; i++) dst[igrid[i]] = src[i];
  }
  P.have_level = ilevel; P.have_ngrid = ngrid; P.have_epoch = P.epoch; P.have_above = interp != 0;
  return 0;
}

// force_fine(ilevel) on an AMR level of a periodic single-rank run (poisson/force_fine.f90:5-194 with gravity_type = 0):
// f(:,1:3) of the level's cells from phi (gradient_phi, values beyond the level's edge interpolated from the level above),
// diag[0] = the level's term of epot_tot (fact * sum f^2 over leaf cells), diag[1] = rho_max(ilevel).
//   fresh = 1: ramses_amd_poisamr_multigrid has just solved this level -- phi, rho of the level and phi, phi_old of the
//   level above are still on the device; otherwise they are read from the host vectors.  f = f(1:ncell,1:3) (host), written
//   on the level's cells.
// several ranks: igrid = the rank's ngrid_own own octs of the level followed by its reception octs (ngrid in all), igrid_c likewise
// for the level above; phi / rho of all of them are read, f and the two diagnostics (the rank's own share, before the caller's
// MPI_ALLREDUCEs, poisson/force_fine.f90:181-186) are computed on the own octs.
static int poisamr_force_impl(int ilevel, int ngrid_own, int ngrid, const int *igrid, int ngrid_c, const int *igrid_c, const double *phi,
                              const double *phi_old, const double *rho, double *f, double tfrac, int interp, int fresh, double fact,
                              double *diag);
int ramses_amd_poisamr_force(int ilevel, int ngrid, const int *igrid, int ngrid_c, const int *igrid_c, const double *phi,
                             const double *phi_old, const double *rho, double *f, double tfrac, int interp, int fresh, double fact,
                             double *diag) {
  return poisamr_force_impl(ilevel, ngrid, ngrid, igrid, ngrid_c, igrid_c, phi, phi_old, rho, f, tfrac, interp, fresh, fact, diag);
}
int ramses_amd_poisamr_force_mpi(int ilevel, int ngrid_own, int ngrid_all, const int *igrid_all, int ngrid_c_all, const int *igrid_c_all,
                                 const double *phi, const double *phi_old, const double *rho, double *f, double tfrac, int interp,
                                 double fact, double *diag) {
  if (ngrid_own < 0 || ngrid_own > ngrid_all) return failf(RAMSES_AMD_EINVAL, "poisamr_force_mpi: bad own / total oct counts");
  if (ngrid_own == 0) { if (diag) { diag[0] = 0.0; diag[1] = 0.0; } return 0; }
  return poisamr_force_impl(ilevel, ngrid_own, ngrid_all, igrid_all, ngrid_c_all, igrid_c_all, phi, phi_old, rho, f, tfrac, interp, 0, fact, diag);
}
// The same for a run whose cell vectors are resident (ramses_amd_amrres_*): f of the rank's own cells goes from the kernel's
// buffer into the resident acceleration on the device (ramses_amd_amrres_take_f_device) and nowhere else; the caller exchanges
// the virtual octs there (ramses_amd_amrres_halo_*, direction 7) instead of the reference's three host exchanges
// (poisson/force_fine.f90:137-139).
extern "C" int ramses_amd_amrres_take_f_device(int ngrid, const int *igrid, const double *d_fpack);
static bool g_force_to_resident = false;
int ramses_amd_poisamr_force_mpi_resident(int ilevel, int ngrid_own, int ngrid_all, const int *igrid_all, int ngrid_c_all, const int *igrid_c_all,
                                          const double *phi, const double *phi_old, const double *rho, double tfrac, int interp, double fact,
                                          double *diag) {
  if (ngrid_own < 0 || ngrid_own > ngrid_all) return failf(RAMSES_AMD_EINVAL, "poisamr_force_mpi: bad own / total oct counts");
  if (ngrid_own == 0) { if (diag) { diag[0] = 0.0; diag[1] = 0.0; } return 0; }
  g_force_to_resident = true;
  double dummy = 0.0;
  const int rc = poisamr_force_impl(ilevel, ngrid_own, ngrid_all, igrid_all, ngrid_c_all, igrid_c_all, phi, phi_old, rho, &dummy, tfrac, interp, 0, fact, diag);
  g_force_to_resident = false;
  return rc;
}
static int poisamr_force_impl(int ilevel, int ngrid_own, int ngrid, const int *igrid, int ngrid_c, const int *igrid_c, const double *phi,
                              const double *phi_old, const double *rho, double *f, double tfrac, int interp, int fresh, double fact,
                              double *diag) {
  PoisAmr &P = g_pa;
  if (!P.tree_valid) return failf(RAMSES_AMD_EINVAL, "poisamr_force: no tree (ramses_amd_poisamr_tree)");
  if (!igrid || !phi || !phi_old || !rho || !f || !diag) return failf(RAMSES_AMD_EINVAL, "NULL argument");
  if (ilevel < 2 || ilevel > 30 || ngrid < 1 || ngrid > P.ngridmax) return failf(RAMSES_AMD_EINVAL, "bad level %d / ngrid %d", ilevel, ngrid);
  if (interp && (!igrid_c || ngrid_c < 1)) return failf(RAMSES_AMD_EINVAL, "the level above is empty");
  hipStream_t s = nullptr;
  const long ncoarse = P.ncoarse, ngridmax = P.ngridmax;
  const long nf = 8L * ngrid, nc = interp ? 8L * ngrid_c : 0;
  const long nfo = 8L * ngrid_own;     // the cells f is computed on
  const size_t vb = sizeof(double) * (size_t)P.ncell;
  HCHK(P.phi.ensure(vb), "hipMalloc phi"); HCHK(P.phi_old.ensure(vb), "hipMalloc phi_old"); HCHK(P.rho.ensure(vb), "hipMalloc rho");
  HCHK(P.f.ensure(3 * vb), "hipMalloc f");
  HCHK(P.fpack.ensure(sizeof(double) * 3 * (size_t)nf), "hipMalloc"); HCHK(P.leaf.ensure(sizeof(int) * (size_t)nf), "hipMalloc");
  HCHK(P.diag.ensure(sizeof(double) * (2 * 512 + 2)), "hipMalloc"); HCHK(P.pack.ensure(sizeof(double) * (size_t)(2 * nf + 2 * nc)), "hipMalloc pack");
  HCHK(P.stage.ensure(sizeof(double) * (size_t)(3 * nf + 2 * nc)), "hipHostMalloc");
  Level &F = P.lev[ilevel];
  const bool have = fresh && ngrid_own == ngrid && P.have_level == ilevel && P.have_ngrid == ngrid && P.have_epoch == P.epoch && (!interp || P.have_above);
  double *hs = P.stage.as<double>();
  if (!have) {
    F.ngrid = ngrid;
    HCHK(F.igrid.ensure(sizeof(int) * (size_t)ngrid), "hipMalloc igrid");
    HCHK(hipMemcpyAsync(F.igrid.p, igrid, sizeof(int) * (size_t)ngrid, hipMemcpyHostToDevice, s), "H2D igrid");
    for (int ind = 0; ind < 8; ind++) {
      const double *s1 = phi + ncoarse + (size_t)ind * ngridmax - 1, *s2 = rho + ncoarse + (size_t)ind * ngridmax - 1;
      double *d1 = hs + (size_t)ind * ngrid, *d2 = hs + nf + (size_t)ind * ngrid;
      for (int i = 0; i < ngrid; i++) { d1[i] = s1[igrid[i]]; d2[i] = s2[igrid[i]]; }
    }
    if (interp)
      for (int ind = 0; ind < 8; ind++) {
        const double *s1 = phi + ncoarse + (size_t)ind * ngridmax - 1, *s2 = phi_old + ncoarse + (size_t)ind * ngridmax - 1;
        double *d1 = hs + 2 * nf + (size_t)ind * ngrid_c, *d2 = hs + 2 * nf + nc + (size_t)ind * ngrid_c;
        for (int i = 0; i < ngrid_c; i++) { d1[i] = s1[igrid_c[i]]; d2[i] = s2[igrid_c[i]]; }
      }
    HCHK(hipMemcpyAsync(P.pack.p, hs, sizeof(double) * (size_t)(2 * nf + 2 * nc), hipMemcpyHostToDevice, s), "H2D level data");
    hipLaunchKernelGGL(vec_scatter_kernel, dim3(grid_for(nf)), dim3(256), 0, s, P.phi.as<double>(), P.pack.as<double>(), F.igrid.as<int>(), ngrid, ncoarse, ngridmax);
    hipLaunchKernelGGL(vec_scatter_kernel, dim3(grid_for(nf)), dim3(256), 0, s, P.rho.as<double>(), P.pack.as<double>() + nf, F.igrid.as<int>(), ngrid, ncoarse, ngridmax);
    if (interp) {
      HCHK(P.igrid_c.ensure(sizeof(int) * (size_t)ngrid_c), "hipMalloc igrid_c");
      HCHK(hipMemcpyAsync(P.igrid_c.p, igrid_c, sizeof(int) * (size_t)ngrid_c, hipMemcpyHostToDevice, s), "H2D igrid_c");
      hipLaunchKernelGGL(vec_scatter_kernel, dim3(grid_for(nc)), dim3(256), 0, s, P.phi.as<double>(), P.pack.as<double>() + 2 * nf, P.igrid_c.as<int>(), ngrid_c, ncoarse, ngridmax);
      hipLaunchKernelGGL(vec_scatter_kernel, dim3(grid_for(nc)), dim3(256), 0, s, P.phi_old.as<double>(), P.pack.as<double>() + 2 * nf + nc, P.igrid_c.as<int>(), ngrid_c, ncoarse, ngridmax);
    }
    HCHK(hipGetLastError(), "scatter launch");
    HCHK(hipStreamSynchronize(s), "sync");     // the staging area is reused below
  }
  P.have_level = 0;
  ForceArgs A;
  A.L = F.view(); A.L.ngrid = ngrid_own; A.T = P.tree();
  A.phi = P.phi.as<double>(); A.phi_old = P.phi_old.as<double>();
  A.tfrac = tfrac; A.interp = interp ? 1 : 0;
  const double dx = std::ldexp(1.0, -ilevel);
  A.a = 0.50 * 4.0 / 3.0 / dx;
  A.b = 0.25 * 1.0 / 3.0 / dx;
  A.out = P.fpack.as<double>(); A.leaf = P.leaf.as<int>();
  hipLaunchKernelGGL(force_kernel, dim3(grid_for(nfo)), dim3(256), 0, s, A);
  HCHK(hipGetLastError(), "force launch");
  // diagnostics over the level's cells (rho packed from the device vector)
  hipLaunchKernelGGL(vec_gather_kernel, dim3(grid_for(nfo)), dim3(256), 0, s, P.rho.as<double>(), P.pack.as<double>(), F.igrid.as<int>(), ngrid_own, ncoarse, ngridmax);
  HCHK(launch_force_diag(P.fpack.as<double>(), P.pack.as<double>(), P.leaf.as<int>(), nfo, fact, P.diag.as<double>() + 2, P.diag.as<double>(), s), "diag launch");
  if (g_force_to_resident) {
    HCHK(hipMemcpyAsync(diag, P.diag.p, sizeof(double) * 2, hipMemcpyDeviceToHost, s), "D2H diag");
    if (int rc = ramses_amd_amrres_take_f_device(ngrid_own, igrid, P.fpack.as<double>())) return rc;    // (same stream: ordered behind the force kernel)
    HCHK(hipStreamSynchronize(s), "sync");
    return 0;
  }
  hipLaunchKernelGGL(vec3_scatter_kernel, dim3(grid_for(nfo)), dim3(256), 0, s, P.f.as<double>(), P.fpack.as<double>(), F.igrid.as<int>(), ngrid_own, ncoarse, ngridmax, P.ncell);
  HCHK(hipMemcpyAsync(hs, P.fpack.p, sizeof(double) * 3 * (size_t)nfo, hipMemcpyDeviceToHost, s), "D2H f");
  HCHK(hipMemcpyAsync(diag, P.diag.p, sizeof(double) * 2, hipMemcpyDeviceToHost, s), "D2H diag");
  HCHK(hipStreamSynchronize(s), "sync");
  for (int d = 0; d < 3; d++)
    for (int ind = 0; ind < 8; ind++) {
      double *dst = f + (size_t)d * P.ncell + ncoarse + (size_t)ind * ngridmax - 1;
      const double *src = hs + (size_t)d * nfo + (size_t)ind * ngrid_own;
      for (int i = 0; i < ngrid_own; i++) dst[igrid[i]] = src[i];
    }
  return 0;
}

int ramses_amd_poisamr_levelmin_mg(void) { return g_pa.levelmin_mg; }

// RAMSES_AMD_PROFILE=1: wall time per shadowed routine and level, accumulated by the Fortran shims
// (ramses_amd_iface: ramses_amd_tic / ramses_amd_toc) and printed when the program ends
int ramses_amd_prof_add(const char *name, int level, double seconds) {
  struct Row { char name[48]; int level; double t, tmax; long n; };
  static Row rows[256];
  static int nrows = 0, state = -1;
  if (state < 0) {
    state = getenv("RAMSES_AMD_PROFILE") != nullptr;
    if (state) atexit([] {
      const char *e = getenv("RAMSES_AMD_PROFILE");          // "1": stderr, anything else: a file to append to
      FILE *fo = (e && e[0] && strcmp(e, "1") != 0) ? fopen(e, "a") : stderr;
      if (!fo) fo = stderr;
      fprintf(fo, "ramses_amd profile (wall seconds inside the shadowed routines)\n");
      for (int i = 0; i < nrows; i++) fprintf(fo, "  %-32s level %2d  calls %6ld  %10.4f s  (longest call %.4f s)\n", rows[i].name, rows[i].level, rows[i].n, rows[i].t, rows[i].tmax);
      if (fo != stderr) fclose(fo);
    });
  }
  if (!state || !name) return 0;
  for (int i = 0; i < nrows; i++)
    if (rows[i].level == level && strncmp(rows[i].name, name, 47) == 0) { rows[i].t += seconds; rows[i].n++; if (seconds > rows[i].tmax) rows[i].tmax = seconds; return 0; }
  if (nrows < 256) { strncpy(rows[nrows].name, name, 47); rows[nrows].name[47] = 0; rows[nrows].level = level; rows[nrows].t = rows[nrows].tmax = seconds; rows[nrows].n = 1; nrows++; }
  return 0;
}

}  // extern "C"

#include "warm.hpp"
RAMSES_AMD_TU_WARM(pois_amr)
