// mhd_amr.hip -- godfine1 of a SOLVER=mhd run on a level of an AMR tree, on MI355X (gfx950): the half of
// mhd/godunov_fine.f90:538-1459 that the brick sweep (csrc/mhd_sweep.hip) leaves out -- levels that are only partly
// refined, with finer levels inside and coarser levels around.
//
//   gather       the 6^3 stencil of every oct of the call's list through the tree (get3cubefather, amr/nbors_utils.f90:5-194);
//                a neighbour oct that does not exist is interpolated from its father cell and that cell's six neighbours
//                (getnborfather :404-525) by mhd/interpol_hydro.f90's interpol_hydro :612-793 -- the five Euler variables with
//                the chosen limiter, the face fields by the divergence-free scheme of interpol_mag :990-1047 (interpol_faces
//                :1052-1241, copy_from_refined_faces :1246-1349, cmp_central_faces :1354-1473, compute_2d_tvd :1478-1527);
//   mag_unsplit  mhd/umuscl.f90:31-238 on the stencil: the functions of mhd_core.hpp / mhd_assemble.hpp, which
//                tests/test_mhd_core_host.py holds bit-exact against the compiled reference;
//   resets       fluxes through faces and EMFs on edges that touch a refined cell (:760-903);
//   update       unew of the oct's eight cells (:909-1022), Euler system then constrained transport;
//   coarse level the fluxes through oct faces and the EMFs on oct edges that border leaf cells of level ilevel-1
//                (:1024-1457).  Floating-point addition is not associative and several octs add to the same coarse cell, so
//                every contribution is emitted with the key of its place in the reference's loops -- (batch of nvector octs,
//                Euler | induction, direction and side | edge, fine face, oct inside the batch) -- sorted (two stable radix
//                sorts: by that key, then by target) and added sequentially per (cell, variable).
//
// One wavefront per oct, the stencil and every intermediate of mag_unsplit in LDS (71 KB: two octs in flight per CU).  This is
// the reference's own decomposition -- every oct recomputes its 6^3 neighbourhood -- and costs accordingly (8 x the traces, 2 x
// the EMFs of the brick sweep); it is the FIRST CORRECT path of AMR levels under SOLVER=mhd, staged through the host's arrays
// (ramses_amd_mhd_godunov_fine_amr_f90), strict arithmetic (-ffp-contract=off, the reference's operation order).
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../include/ramses_amd.h"
#include "mhd_assemble.hpp"

using namespace ramses_amd;
using namespace ramses_amd::mhd;

extern "C" int ramses_amd_set_error(int code, const char *msg);   // capi.hip
static int failf(int code, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  return ramses_amd_set_error(code, buf);
}
// RAMSES_AMD_DEBUG_SYNC=1: synchronise and name the stage after every launch (a fault then points at its kernel)
static int dbg_stage(const char *what) {
  static int on = -1;
  if (on < 0) { const char *e = getenv("RAMSES_AMD_DEBUG_SYNC"); on = (e && e[0] == '1') ? 1 : 0; }
  if (!on) return 0;
  hipError_t e = hipDeviceSynchronize();
  fprintf(stderr, "ramses_amd[mhd_amr]: %s -> %s\n", what, hipGetErrorString(e));
  fflush(stderr);
  return e == hipSuccess ? 0 : 1;
}
#define HCHK(call, what) do { hipError_t e_ = (call); if (e_ != hipSuccess) return failf(RAMSES_AMD_EHIP, "%s: %s", what, hipGetErrorString(e_)); } while (0)

namespace {

constexpr int NF = 11;          // uold(:,1:nvar+3): rho, rho u, rho v, rho w, E, B left faces (6:8), B right faces (nvar+1:nvar+3)

struct MhdAmrArgs {
  const double *uold;           // [11][ncell]
  double *unew;                 // [11][ncell]
  const double *grav;           // f(1:ncell,1:3) or null
  const int *son, *nbor, *father, *igrid;
  int ngrid;
  long ncell, ncoarse, ngridmax;
  double dt, dx;
  int interpol_var, interpol_type, interpol_mag_type;
  int coarse;                   // ilevel > levelmin: the coarser level takes its corrections
  int *nfc;                     // [ngrid][27] the 3^3 father cells of every oct (for the edge corrections)
  double *rec_flux;             // [ngrid][6][4][5] Euler fluxes through the oct's faces, after the resets
  double *rec_emf;              // [ngrid][12][2] EMFs on the oct's twelve edges (two fine edges each), after the resets
  int *err;
  MhdConst P;
};

// ---- the tree (amr/nbors_utils.f90) -----------------------------------------------------------------------------------
// same-level neighbour of cell c (level >= 2) in direction dir (0:-x 1:+x 2:-y 3:+y 4:-z 5:+z); if its oct does not exist:
// -(the cell of the coarser level there), getnborfather's fallback
__device__ __forceinline__ int tree_nbor_cell(const MhdAmrArgs &A, int c, int dir) {
  const int pos = (int)((c - A.ncoarse - 1) / A.ngridmax);
  const int g = (int)(c - A.ncoarse - (long)pos * A.ngridmax);
  const int axis = dir >> 1, up = dir & 1;
  const int bit = (pos >> axis) & 1;
  if (bit != up) return c + (up ? 1 : -1) * (int)((1 << axis) * A.ngridmax);
  const int nb = A.nbor[(long)dir * A.ngridmax + g - 1];
  if (nb <= 0) return -c;               // beyond the outermost boundary layer (the reference reads uold(0,:) there: never used)
  const int g2 = A.son[nb - 1];
  if (g2 == 0) return -nb;
  return (int)(A.ncoarse + (long)(pos ^ (1 << axis)) * A.ngridmax + g2);
}

// ---- mhd/interpol_hydro.f90 -------------------------------------------------------------------------------------------
__device__ __forceinline__ void il_minmod(const double (&a)[7], double (&w)[3]) {           // compute_limiter_minmod :798-825
#pragma unroll
  for (int d = 0; d < 3; d++) {
    const double dl = 0.5 * (a[2 * d + 2] - a[0]);
    const double dr = 0.5 * (a[0] - a[2 * d + 1]);
    double mm = 0.0;
    if (!(dl * dr <= 0.0)) mm = fmin2(__builtin_fabs(dl), __builtin_fabs(dr)) * dl / __builtin_fabs(dl);
    w[d] = mm;
  }
}
__device__ __forceinline__ void il_central(const double (&a)[7], double (&w)[3]) {          // compute_limiter_central :853-985
#pragma unroll
  for (int d = 0; d < 3; d++) w[d] = 0.25 * (a[2 * d + 2] - a[2 * d + 1]);
  double ac[8];
#pragma unroll
  for (int ind = 0; ind < 8; ind++) ac[ind] = a[0];
#pragma unroll
  for (int d = 0; d < 3; d++)
#pragma unroll
    for (int ind = 0; ind < 8; ind++) {
      const double xc = (double)((ind >> d) & 1) - 0.5;
      ac[ind] = ac[ind] + 2.0 * w[d] * xc;
    }
  double corner = ac[0], kernel = a[1];
#pragma unroll
  for (int j = 1; j < 8; j++) corner = fmax2(corner, ac[j]);
#pragma unroll
  for (int j = 2; j <= 6; j++) kernel = fmax2(kernel, a[j]);
  double dk = a[0] - kernel, dc = a[0] - corner;
  double max_lim = 0.0;
  if (dk * dc > 0.0) max_lim = fmin2(1.0, dk / dc);
  corner = ac[0]; kernel = a[1];
#pragma unroll
  for (int j = 1; j < 8; j++) corner = fmin2(corner, ac[j]);
#pragma unroll
  for (int j = 2; j <= 6; j++) kernel = fmin2(kernel, a[j]);
  dk = a[0] - kernel; dc = a[0] - corner;
  double min_lim = 0.0;
  if (dk * dc > 0.0) min_lim = fmin2(1.0, dk / dc);
  const double lim = fmin2(min_lim, max_lim);
#pragma unroll
  for (int d = 0; d < 3; d++) w[d] = w[d] * lim;
}
// compute_2d_tvd :1478-1527: the two transverse slopes of a face field from the face's own value b[0] and its four transverse
// neighbours b[1..4]
__device__ __forceinline__ void tvd2d(const double (&b)[5], int imt, double (&s)[2]) {
  s[0] = 0.0; s[1] = 0.0;
  if (imt <= 0) return;
#pragma unroll
  for (int t = 0; t < 2; t++) {
    const double bl = b[1 + 2 * t], br = b[2 + 2 * t];
    if (imt == 3) {
      const double dlft = 0.5 * (b[0] - bl), drgt = 0.5 * (br - b[0]);
      s[t] = dlft + drgt;
    } else {
      const double fi = (double)imt;
      const double dlft = fi * (b[0] - bl), drgt = fi * (br - b[0]);
      const double dcen = 0.5 * (dlft + drgt) / fi;
      const double dsgn = __builtin_copysign(1.0, dcen);
      double dlim = fmin2(__builtin_fabs(dlft), __builtin_fabs(drgt));
      if ((dlft * drgt) <= 0.0) dlim = 0.0;
      s[t] = dsgn * fmin2(dlim, __builtin_fabs(dcen));
    }
  }
}
// interpol_hydro :612-793 of one father cell: u1[0] the cell, u1[1..6] its -x,+x,-y,+y,-z,+z neighbours, ind1[j] = son of
// those cells; u2[ind] = the eight children (ind = ix + 2 iy + 4 iz)
__device__ __noinline__ void mhd_interpol_oct(const MhdAmrArgs &A, double (&u1)[7][NF], const int (&ind1)[7], double (&u2)[8][NF]) {
  const double smallr = A.P.smallr;
  if (A.interpol_var == 1) {
    for (int j = 0; j < 7; j++) {
      double ekin = 0.0, emag = 0.0;
      for (int d = 0; d < 3; d++) ekin = ekin + 0.5 * (u1[j][d + 1] * u1[j][d + 1]) / fmax2(u1[j][0], smallr);
      for (int d = 0; d < 3; d++) { const double bs = u1[j][d + 5] + u1[j][d + 8]; emag = emag + 0.125 * (bs * bs); }
      const double erad = 0.0;
      u1[j][4] = u1[j][4] - ekin - emag - erad;
    }
  }
  // the cell-centred variables: ivar <= neul (nvar = 8, ndim = 3: nothing beyond the three face fields)
  for (int v = 0; v < 5; v++) {
    double a[7], w[3] = {0.0, 0.0, 0.0};
    for (int j = 0; j < 7; j++) a[j] = u1[j][v];
    if (A.interpol_type == 1) il_minmod(a, w);
    else if (A.interpol_type == 2) il_central(a, w);
    else if (A.interpol_type == 3) { for (int d = 0; d < 3; d++) w[d] = 0.5 * (a[2 * d + 2] - a[2 * d + 1]); }      // compute_central :830-848
    for (int ind = 0; ind < 8; ind++) {
      double val = a[0];
      for (int d = 0; d < 3; d++) val = val + w[d] * ((double)((ind >> d) & 1) - 0.5);
      u2[ind][v] = val;
    }
  }
  // interpol_mag: fine fields on the coarse cell's six faces (u, v, w at index -1 / +1 of the normal direction) ...
  double U[3][2][2], V[2][3][2], W[2][2][3];       // u(-1:1,0:1,0:1), v(0:1,-1:1,0:1), w(0:1,0:1,-1:1): the normal index + 1
  const int imt = A.interpol_mag_type;
  for (int side = 0; side < 2; side++) {
    const int off = side ? 8 : 5;                  // right faces: uold(:,nvar+1:nvar+3), left: uold(:,6:8)
    double b[5], s[2];
    // Bx on an x face: transverse neighbours along y (u1[3], u1[4]) and z (u1[5], u1[6])
    b[0] = u1[0][off]; b[1] = u1[3][off]; b[2] = u1[4][off]; b[3] = u1[5][off]; b[4] = u1[6][off];
    tvd2d(b, imt, s);
    for (int j = 0; j < 2; j++)
      for (int k = 0; k < 2; k++) U[2 * side][j][k] = b[0] + 0.5 * s[0] * ((double)j - 0.5) + 0.5 * s[1] * ((double)k - 0.5);
    // By on a y face: along x (u1[1], u1[2]) and z
    b[0] = u1[0][off + 1]; b[1] = u1[1][off + 1]; b[2] = u1[2][off + 1]; b[3] = u1[5][off + 1]; b[4] = u1[6][off + 1];
    tvd2d(b, imt, s);
    for (int i = 0; i < 2; i++)
      for (int k = 0; k < 2; k++) V[i][2 * side][k] = b[0] + 0.5 * s[0] * ((double)i - 0.5) + 0.5 * s[1] * ((double)k - 0.5);
    // Bz on a z face: along x and y
    b[0] = u1[0][off + 2]; b[1] = u1[1][off + 2]; b[2] = u1[2][off + 2]; b[3] = u1[3][off + 2]; b[4] = u1[4][off + 2];
    tvd2d(b, imt, s);
    for (int i = 0; i < 2; i++)
      for (int j = 0; j < 2; j++) W[i][j][2 * side] = b[0] + 0.5 * s[0] * ((double)i - 0.5) + 0.5 * s[1] * ((double)j - 0.5);
  }
  // ... taken from the finer level where the neighbour across the face is refined (copy_from_refined_faces) ...
  const long N = A.ncell;
  auto fine = [&](int oct, int ind, int var) -> double { return A.uold[(long)var * N + A.ncoarse + (long)ind * A.ngridmax + oct - 1]; };
  for (int a = 0; a < 2; a++)
    for (int b = 0; b < 2; b++) {
      if (ind1[1] > 0) U[0][a][b] = fine(ind1[1], 1 + 2 * a + 4 * b, 8);       // -x neighbour's +x cells, their right Bx
      if (ind1[2] > 0) U[2][a][b] = fine(ind1[2], 0 + 2 * a + 4 * b, 5);
      if (ind1[3] > 0) V[a][0][b] = fine(ind1[3], a + 2 + 4 * b, 9);
      if (ind1[4] > 0) V[a][2][b] = fine(ind1[4], a + 0 + 4 * b, 6);
      if (ind1[5] > 0) W[a][b][0] = fine(ind1[5], a + 2 * b + 4, 10);
      if (ind1[6] > 0) W[a][b][2] = fine(ind1[6], a + 2 * b + 0, 7);
    }
  // ... and the fields on the faces inside the coarse cell, divergence free (cmp_central_faces, Toth & Balsara)
  double UXX = 0.0, VYY = 0.0, WZZ = 0.0, UXYZ = 0.0, VXYZ = 0.0, WXYZ = 0.0;
  for (int i = 0; i < 2; i++)
    for (int j = 0; j < 2; j++)
      for (int k = 0; k < 2; k++) {
        const int ii = 2 * i - 1, jj = 2 * j - 1, kk = 2 * k - 1;
        const double uu = U[ii + 1][j][k], vv = V[i][jj + 1][k], ww = W[i][j][kk + 1];
        UXX = UXX + ((double)(ii * jj) * vv + (double)(ii * kk) * ww) * 0.125;
        VYY = VYY + ((double)(jj * kk) * ww + (double)(ii * jj) * uu) * 0.125;
        WZZ = WZZ + ((double)(ii * kk) * uu + (double)(jj * kk) * vv) * 0.125;
        UXYZ = UXYZ + ((double)(ii * jj * kk) * uu) * 0.125;
        VXYZ = VXYZ + ((double)(ii * jj * kk) * vv) * 0.125;
        WXYZ = WXYZ + ((double)(ii * jj * kk) * ww) * 0.125;
      }
  for (int a = 0; a < 2; a++)
    for (int b = 0; b < 2; b++) {
      U[1][a][b] = 0.5 * (U[0][a][b] + U[2][a][b]) + UXX + ((double)b - 0.5) * VXYZ + ((double)a - 0.5) * WXYZ;     // u(0,j=a,k=b)
      V[a][1][b] = 0.5 * (V[a][0][b] + V[a][2][b]) + VYY + ((double)a - 0.5) * WXYZ + ((double)b - 0.5) * UXYZ;     // v(i=a,0,k=b)
      W[a][b][1] = 0.5 * (W[a][b][0] + W[a][b][2]) + WZZ + ((double)b - 0.5) * UXYZ + ((double)a - 0.5) * VXYZ;     // w(i=a,j=b,0)
    }
  for (int i = 0; i < 2; i++)
    for (int j = 0; j < 2; j++)
      for (int k = 0; k < 2; k++) {
        const int ind = i + 2 * j + 4 * k;
        u2[ind][5] = U[i][j][k]; u2[ind][6] = V[i][j][k]; u2[ind][7] = W[i][j][k];
        u2[ind][8] = U[i + 1][j][k]; u2[ind][9] = V[i][j + 1][k]; u2[ind][10] = W[i][j][k + 1];
      }
  if (A.interpol_var == 1) {
    for (int ind = 0; ind < 8; ind++) {
      double ekin = 0.0, emag = 0.0;
      for (int d = 0; d < 3; d++) ekin = ekin + 0.5 * (u2[ind][d + 1] * u2[ind][d + 1]) / fmax2(u2[ind][0], smallr);
      for (int d = 0; d < 3; d++) { const double bs = u2[ind][d + 5] + u2[ind][d + 8]; emag = emag + 0.125 * (bs * bs); }
      const double erad = 0.0;
      u2[ind][4] = u2[ind][4] + ekin + emag + erad;
    }
  }
}

// ---- one oct per wavefront --------------------------------------------------------------------------------------------
constexpr int SC = 216;                                  // cells of the 6^3 stencil, index (i+1) + 6 (j+1) + 36 (k+1), i,j,k = -1..4
__device__ __forceinline__ int sidx(int i, int j, int k) { return (i + 1) + 6 * ((j + 1) + 6 * (k + 1)); }
struct OctLds {
  double U[NF][SC];
  double Q[8][SC];
  double G[3][SC];
  double Ef[3][SC];
  double TR[NPRED][64];                                  // trace_predict of the cells 0..3 in each direction
  double FX[3][5][27];                                   // Euler fluxes through the low faces of the cells 1..3, scaled by dt/dx
  double EM[3][27];                                      // EMFs on the low edges of the cells 1..3, scaled by dt/dx
  int fc[27], fs[27];                                    // the 27 father cells and their sons
  unsigned char ok[SC];                                  // refined cells
};
struct OctAcc {
  const OctLds *L;
  __device__ __forceinline__ double q(int n, int i, int j, int k) const { return L->Q[n][sidx(i, j, k)]; }
  // ctoprim's bf :2062-2100: the left-face field of the cell, the right-face field of the last cell beyond the stencil
  __device__ __forceinline__ double bf(int c, int i, int j, int k) const {
    const int t[3] = {i, j, k};
    if (t[c] <= 4) return L->U[5 + c][sidx(i, j, k)];
    return L->U[8 + c][sidx(i - (c == 0), j - (c == 1), k - (c == 2))];
  }
  __device__ __forceinline__ double E(int c, int i, int j, int k) const { return L->Ef[c][sidx(i, j, k)]; }
};
struct LdsPred {
  const OctLds *L;
  int cell;                                              // i + 4 j + 16 k, i,j,k = 0..3
  __device__ __forceinline__ double c(int n) const { return L->TR[n][cell]; }
  __device__ __forceinline__ double f(int n) const { return L->TR[8 + n][cell]; }
  __device__ __forceinline__ double h(int n) const { return L->TR[14 + n][cell]; }
};
__device__ __forceinline__ int tcell(int i, int j, int k) { return i + 4 * (j + 4 * k); }

template <int D>
__device__ __forceinline__ void oct_face_flux(OctLds &L, const MhdAmrArgs &A, int i, int j, int k) {
  const LdsPred lo{&L, tcell(i - (D == 0), j - (D == 1), k - (D == 2))}, me{&L, tcell(i, j, k)};
  double qm_[8], qp_[8], f[8];
  trace_state<T_QM, D>(lo, A.P, qm_);
  trace_state<T_QP, D>(me, A.P, qp_);
  cmpflxm_face<-1>(qm_, qp_, D, A.P, f);
  const int o = (i - 1) + 3 * ((j - 1) + 3 * (k - 1));
  for (int n = 0; n < 5; n++) L.FX[D][n][o] = f[n] * A.dt / A.dx;
}
template <int E>
__device__ __forceinline__ void oct_edge_emf(OctLds &L, const MhdAmrArgs &A, int i, int j, int k) {
  double rt[8], rb[8], lt[8], lb[8];
  const LdsPred me{&L, tcell(i, j, k)};
  if constexpr (E == 2) {
    const LdsPred a{&L, tcell(i - 1, j - 1, k)}, b{&L, tcell(i - 1, j, k)}, c{&L, tcell(i, j - 1, k)};
    trace_state<T_QRT, 2>(a, A.P, rt); trace_state<T_QRB, 2>(b, A.P, rb); trace_state<T_QLT, 2>(c, A.P, lt);
  } else if constexpr (E == 1) {
    const LdsPred a{&L, tcell(i - 1, j, k - 1)}, b{&L, tcell(i, j, k - 1)}, c{&L, tcell(i - 1, j, k)};
    trace_state<T_QRT, 1>(a, A.P, rt); trace_state<T_QLT, 1>(b, A.P, rb); trace_state<T_QRB, 1>(c, A.P, lt);
  } else {
    const LdsPred a{&L, tcell(i, j - 1, k - 1)}, b{&L, tcell(i, j - 1, k)}, c{&L, tcell(i, j, k - 1)};
    trace_state<T_QRT, 0>(a, A.P, rt); trace_state<T_QRB, 0>(b, A.P, rb); trace_state<T_QLT, 0>(c, A.P, lt);
  }
  trace_state<T_QLB, E>(me, A.P, lb);
  L.EM[E][(i - 1) + 3 * ((j - 1) + 3 * (k - 1))] = cmp_mag_flx_edge<-1>(rt, rb, lt, lb, E, A.P) * A.dt / A.dx;
}

template <bool S3, bool GRAV>
__global__ __launch_bounds__(64) void mhd_amr_oct_kernel(MhdAmrArgs A) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  OctLds &L = *reinterpret_cast<OctLds *>(smem_raw);
  const int lane = threadIdx.x;
  const int io = blockIdx.x;
  const int g = A.igrid[io];
  const long N = A.ncell;
  // ---- the 27 father cells: from the oct's own father cell, x, then y, then z steps (get3cubefather)
  if (lane < 27) {
    const int st[3] = {lane % 3 - 1, (lane / 3) % 3 - 1, lane / 9 - 1};
    int c = A.father[g - 1];
    for (int axis = 0; axis < 3; axis++)
      if (st[axis] != 0 && c > 0) {
        if (c <= A.ncoarse) { c = 0; break; }            // (a level-1 father cell: the launcher refuses ilevel < 3)
        c = tree_nbor_cell(A, c, 2 * axis + (st[axis] > 0 ? 1 : 0));
        if (c < 0) c = 0;
      }
    if (c <= 0) { atomicAdd(A.err, 1); c = A.father[g - 1]; }
    L.fc[lane] = c;
    L.fs[lane] = A.son[c - 1];
    A.nfc[(long)io * 27 + lane] = c;
  }
  __syncthreads();
  // ---- the stencil: octs that exist are copied ...
  for (int t = lane; t < 27 * 8; t += 64) {
    const int nb = t >> 3, ind = t & 7;
    const int s = L.fs[nb];
    if (s <= 0) continue;
    const int i = 1 + 2 * (nb % 3 - 1) + (ind & 1), j = 1 + 2 * ((nb / 3) % 3 - 1) + ((ind >> 1) & 1), k = 1 + 2 * (nb / 9 - 1) + (ind >> 2);   // i3 = 1 + 2 (i1 - 1) + i2
    const long c = A.ncoarse + (long)ind * A.ngridmax + s - 1;
    const int o = sidx(i, j, k);
    for (int v = 0; v < NF; v++) L.U[v][o] = A.uold[(long)v * N + c];
    if (GRAV) for (int d = 0; d < 3; d++) L.G[d][o] = A.grav[(long)d * N + c];
    L.ok[o] = A.son[c] > 0 ? 1 : 0;
  }
  // ... the others are interpolated from the coarser level, one lane each
  if (lane < 27 && L.fs[lane] <= 0) {
    const int nb = lane;
    const int c0 = L.fc[nb];
    int cells[7];
    cells[0] = c0;
    for (int d = 0; d < 6; d++) {
      int c = c0 > A.ncoarse ? tree_nbor_cell(A, c0, d) : c0;
      if (c < 0) c = -c;
      cells[d + 1] = c;
    }
    double u1[7][NF], u2[8][NF];
    int ind1[7];
    for (int jn = 0; jn < 7; jn++) {
      for (int v = 0; v < NF; v++) u1[jn][v] = A.uold[(long)v * N + cells[jn] - 1];
      ind1[jn] = A.son[cells[jn] - 1];
    }
    mhd_interpol_oct(A, u1, ind1, u2);
    for (int ind = 0; ind < 8; ind++) {
      const int i = 1 + 2 * (nb % 3 - 1) + (ind & 1), j = 1 + 2 * ((nb / 3) % 3 - 1) + ((ind >> 1) & 1), k = 1 + 2 * (nb / 9 - 1) + (ind >> 2);   // i3 = 1 + 2 (i1 - 1) + i2
      const int o = sidx(i, j, k);
      for (int v = 0; v < NF; v++) L.U[v][o] = u2[ind][v];
      if (GRAV) for (int d = 0; d < 3; d++) L.G[d][o] = A.grav[(long)d * N + c0 - 1];
      L.ok[o] = 0;
    }
  }
  __syncthreads();
  // ---- ctoprim
  for (int o = lane; o < SC; o += 64) {
    const double u[5] = {L.U[0][o], L.U[1][o], L.U[2][o], L.U[3][o], L.U[4][o]};
    const double bl[3] = {L.U[5][o], L.U[6][o], L.U[7][o]};
    const double br[3] = {L.U[8][o], L.U[9][o], L.U[10][o]};
    double gv[3] = {0.0, 0.0, 0.0}, q[8];
    if (GRAV) { gv[0] = L.G[0][o]; gv[1] = L.G[1][o]; gv[2] = L.G[2][o]; }
    ctoprim_cell(u, bl, br, GRAV ? gv : nullptr, A.dt, A.P, q);
    for (int n = 0; n < 8; n++) L.Q[n][o] = q[n];
  }
  __syncthreads();
  const OctAcc acc{&L};
  // ---- edge-centred electric fields of the cells 0..4
  for (int t = lane; t < 125 * 3; t += 64) {
    const int c = t / 125, r = t % 125;
    const int i = r % 5, j = (r / 5) % 5, k = r / 25;
    L.Ef[c][sidx(i, j, k)] = efield(acc, c, i, j, k);
  }
  __syncthreads();
  // ---- slopes + predictor of the cells 0..3
  {
    const int i = lane & 3, j = (lane >> 2) & 3, k = lane >> 4;
    TraceIn I;
    trace_inputs<S3>(acc, i, j, k, A.P, I);
    TracePred T;
    const double dtdx = A.dt / A.dx;
    trace_predict(I, dtdx, dtdx, dtdx, A.P, T);
    for (int n = 0; n < NPRED; n++) L.TR[n][lane] = T.v[n];
  }
  __syncthreads();
  // ---- fluxes through the faces if1..if2 (12 per direction) and EMFs on the edges (18 per direction)
  if (lane < 36) {
    const int d = lane / 12, r = lane % 12;
    // the direction of the flux runs 1..3, the two others 1..2
    int p[3];
    if (d == 0) { p[0] = 1 + r % 3; p[1] = 1 + (r / 3) % 2; p[2] = 1 + r / 6; }
    else if (d == 1) { p[1] = 1 + r % 3; p[0] = 1 + (r / 3) % 2; p[2] = 1 + r / 6; }
    else { p[2] = 1 + r % 3; p[0] = 1 + (r / 3) % 2; p[1] = 1 + r / 6; }
    if (d == 0) oct_face_flux<0>(L, A, p[0], p[1], p[2]);
    else if (d == 1) oct_face_flux<1>(L, A, p[0], p[1], p[2]);
    else oct_face_flux<2>(L, A, p[0], p[1], p[2]);
  }
  if (lane < 54) {
    const int e = lane / 18, r = lane % 18;
    // the direction of the edge runs 1..2, the two others 1..3
    int p[3];
    if (e == 0) { p[0] = 1 + r % 2; p[1] = 1 + (r / 2) % 3; p[2] = 1 + r / 6; }
    else if (e == 1) { p[1] = 1 + r % 2; p[0] = 1 + (r / 2) % 3; p[2] = 1 + r / 6; }
    else { p[2] = 1 + r % 2; p[0] = 1 + (r / 2) % 3; p[1] = 1 + r / 6; }
    if (e == 0) oct_edge_emf<0>(L, A, p[0], p[1], p[2]);
    else if (e == 1) oct_edge_emf<1>(L, A, p[0], p[1], p[2]);
    else oct_edge_emf<2>(L, A, p[0], p[1], p[2]);
  }
  __syncthreads();
  // ---- the resets (:760-903): a flux through a face with a refined cell on either side, an EMF on an edge with a refined
  // cell among the four around it
  auto okc = [&](int i, int j, int k) -> bool { return L.ok[sidx(i, j, k)] != 0; };
  auto FXv = [&](int d, int n, int i, int j, int k) -> double {
    if (okc(i - (d == 0), j - (d == 1), k - (d == 2)) || okc(i, j, k)) return 0.0;
    return L.FX[d][n][(i - 1) + 3 * ((j - 1) + 3 * (k - 1))];
  };
  auto EMv = [&](int e, int i, int j, int k) -> double {
    bool z;
    if (e == 2) z = okc(i, j, k) || okc(i, j - 1, k) || okc(i - 1, j, k) || okc(i - 1, j - 1, k);
    else if (e == 1) z = okc(i, j, k) || okc(i, j, k - 1) || okc(i - 1, j, k) || okc(i - 1, j, k - 1);
    else z = okc(i, j, k) || okc(i, j, k - 1) || okc(i, j - 1, k) || okc(i, j - 1, k - 1);
    if (z) return 0.0;
    return L.EM[e][(i - 1) + 3 * ((j - 1) + 3 * (k - 1))];
  };
  // ---- conservative update of the oct's eight cells (:909-1022)
  if (lane < 8) {
    const int i2 = lane & 1, j2 = (lane >> 1) & 1, k2 = lane >> 2;
    const int i3 = 1 + i2, j3 = 1 + j2, k3 = 1 + k2;
    const long c = A.ncoarse + (long)lane * A.ngridmax + g - 1;
    double un[NF];
    for (int v = 0; v < NF; v++) un[v] = A.unew[(long)v * N + c];
    for (int d = 0; d < 3; d++) {
      const int i0 = d == 0, j0 = d == 1, k0 = d == 2;
      for (int n = 0; n < 5; n++) un[n] = un[n] + (FXv(d, n, i3, j3, k3) - FXv(d, n, i3 + i0, j3 + j0, k3 + k0));
      // the face fields take part with their Euler fluxes reset to zero
      const double z = 0.0;
      for (int n = 5; n < NF; n++) un[n] = un[n] + (z - z);
    }
    double df;
    df = (EMv(1, i3, j3, k3) - EMv(1, i3, j3, k3 + 1)) - (EMv(2, i3, j3, k3) - EMv(2, i3, j3 + 1, k3));                 un[5] = un[5] + df;
    df = (EMv(1, i3 + 1, j3, k3) - EMv(1, i3 + 1, j3, k3 + 1)) - (EMv(2, i3 + 1, j3, k3) - EMv(2, i3 + 1, j3 + 1, k3)); un[8] = un[8] + df;
    df = (EMv(2, i3, j3, k3) - EMv(2, i3 + 1, j3, k3)) - (EMv(0, i3, j3, k3) - EMv(0, i3, j3, k3 + 1));                 un[6] = un[6] + df;
    df = (EMv(2, i3, j3 + 1, k3) - EMv(2, i3 + 1, j3 + 1, k3)) - (EMv(0, i3, j3 + 1, k3) - EMv(0, i3, j3 + 1, k3 + 1)); un[9] = un[9] + df;
    df = (EMv(0, i3, j3, k3) - EMv(0, i3, j3 + 1, k3)) - (EMv(1, i3, j3, k3) - EMv(1, i3 + 1, j3, k3));                 un[7] = un[7] + df;
    df = (EMv(0, i3, j3, k3 + 1) - EMv(0, i3, j3 + 1, k3 + 1)) - (EMv(1, i3, j3, k3 + 1) - EMv(1, i3 + 1, j3, k3 + 1)); un[10] = un[10] + df;
    for (int v = 0; v < NF; v++) A.unew[(long)v * N + c] = un[v];
  }
  // ---- what the coarser level may be owed: the fluxes through the oct's six faces (four fine faces each: the transverse
  // coordinates in ascending axis order, as the reference loops k3, j3, i3 with i3 fastest) and the EMFs on its twelve edges
  if (A.coarse) {
    if (lane < 24) {
      const int f = lane >> 2, q = lane & 3;
      const int d = f >> 1, side = f & 1;
      int p[3];
      const int t0 = d == 0 ? 1 : 0, t1 = d == 2 ? 1 : 2;
      p[d] = side ? 3 : 1; p[t0] = 1 + (q & 1); p[t1] = 1 + (q >> 1);
      for (int n = 0; n < 5; n++) A.rec_flux[(((long)io * 6 + f) * 4 + q) * 5 + n] = FXv(d, n, p[0], p[1], p[2]);
    }
    if (lane < 24) {
      // edge ed = 0..11 in the reference's order (EMFz: X0Y0 X0Y1 X1Y1 X1Y0; EMFx: Y0Z0 Y0Z1 Y1Z1 Y1Z0; EMFy: X0Z0 X0Z1 X1Z1 X1Z0),
      // its two fine edges h = 0, 1 along the edge
      const int ed = lane >> 1, h = lane & 1;
      const int e = ed < 4 ? 2 : (ed < 8 ? 0 : 1);
      const int m = ed & 3;
      const int a = (m == 0 || m == 1) ? 1 : 3, b = (m == 0 || m == 3) ? 1 : 3;     // first / second transverse coordinate
      int p[3];
      if (e == 2) { p[0] = a; p[1] = b; p[2] = 1 + h; }
      else if (e == 0) { p[1] = a; p[2] = b; p[0] = 1 + h; }
      else { p[0] = a; p[2] = b; p[1] = 1 + h; }
      A.rec_emf[((long)io * 12 + ed) * 2 + h] = EMv(e, p[0], p[1], p[2]);
    }
  }
}

// ---- the coarser level (:1024-1457) -----------------------------------------------------------------------------------
// the twelve edges: offsets of the three father cells (buffer 1, 2, 3) in the 3^3 cube and the six updates
// (buffer, variable 5..10, sign, halved and only where the three cells are leaves)
struct EdgeUpd { signed char buf, var, sgn, half; };
struct EdgeRule { signed char f[3][3]; EdgeUpd u[6]; };
__constant__ EdgeRule EDGE_RULES[12] = {
    // EMFz: Bx (5 left, 8 right), By (6, 9)
    {{{0, -1, 0}, {-1, -1, 0}, {-1, 0, 0}}, {{1, 5, +1, 0}, {2, 8, +1, 0}, {2, 9, -1, 0}, {3, 6, -1, 0}, {3, 8, -1, 1}, {1, 9, +1, 1}}},
    {{{-1, 0, 0}, {-1, +1, 0}, {0, +1, 0}}, {{1, 9, -1, 0}, {2, 6, -1, 0}, {2, 8, -1, 0}, {3, 5, -1, 0}, {3, 6, +1, 1}, {1, 8, +1, 1}}},
    {{{0, +1, 0}, {+1, +1, 0}, {+1, 0, 0}}, {{1, 8, -1, 0}, {2, 5, -1, 0}, {2, 6, +1, 0}, {3, 9, +1, 0}, {3, 5, +1, 1}, {1, 6, -1, 1}}},
    {{{+1, 0, 0}, {+1, -1, 0}, {0, -1, 0}}, {{1, 6, +1, 0}, {2, 9, +1, 0}, {2, 5, +1, 0}, {3, 8, +1, 0}, {3, 9, -1, 1}, {1, 5, -1, 1}}},
    // EMFx: By (6, 9), Bz (7, 10)
    {{{0, 0, -1}, {0, -1, -1}, {0, -1, 0}}, {{1, 6, +1, 0}, {2, 9, +1, 0}, {2, 10, -1, 0}, {3, 7, -1, 0}, {1, 10, +1, 1}, {3, 9, -1, 1}}},
    {{{0, -1, 0}, {0, -1, +1}, {0, 0, +1}}, {{1, 10, -1, 0}, {2, 7, -1, 0}, {2, 9, -1, 0}, {3, 6, -1, 0}, {1, 9, +1, 1}, {3, 7, +1, 1}}},
    {{{0, 0, +1}, {0, +1, +1}, {0, +1, 0}}, {{1, 9, -1, 0}, {2, 6, -1, 0}, {2, 7, +1, 0}, {3, 10, +1, 0}, {3, 6, +1, 1}, {1, 7, -1, 1}}},
    {{{0, +1, 0}, {0, +1, -1}, {0, 0, -1}}, {{1, 7, +1, 0}, {2, 10, +1, 0}, {2, 6, +1, 0}, {3, 9, +1, 0}, {3, 10, -1, 1}, {1, 6, -1, 1}}},
    // EMFy: Bx (5, 8), Bz (7, 10)
    {{{0, 0, -1}, {-1, 0, -1}, {-1, 0, 0}}, {{1, 5, -1, 0}, {2, 8, -1, 0}, {2, 10, +1, 0}, {3, 7, +1, 0}, {3, 8, +1, 1}, {1, 10, -1, 1}}},
    {{{-1, 0, 0}, {-1, 0, +1}, {0, 0, +1}}, {{1, 10, +1, 0}, {2, 7, +1, 0}, {2, 8, +1, 0}, {3, 5, +1, 0}, {3, 7, -1, 1}, {1, 8, -1, 1}}},
    {{{0, 0, +1}, {+1, 0, +1}, {+1, 0, 0}}, {{1, 8, +1, 0}, {2, 5, +1, 0}, {2, 7, -1, 0}, {3, 10, -1, 0}, {3, 5, -1, 1}, {1, 7, +1, 1}}},
    {{{+1, 0, 0}, {+1, 0, -1}, {0, 0, -1}}, {{1, 7, -1, 0}, {2, 10, -1, 0}, {2, 5, -1, 0}, {3, 8, -1, 0}, {3, 10, +1, 1}, {1, 5, +1, 1}}},
};

// contributions of oct io: COUNT = true counts them, else writes (order key, target, value) from offset[io] on
// order key = ((((batch * 2 + phase) * 12 + sub) * 4 + q) * nvector + i)
template <bool COUNT>
__global__ __launch_bounds__(256) void mhd_amr_emit_kernel(MhdAmrArgs A, int nvector, const unsigned *__restrict__ offset, unsigned *__restrict__ count,
                                                           unsigned long long *__restrict__ okey, unsigned long long *__restrict__ tkey,
                                                           double *__restrict__ val) {
  const int io = blockIdx.x * blockDim.x + threadIdx.x;
  if (io >= A.ngrid) return;
  const int g = A.igrid[io];
  const long N = A.ncell;
  unsigned n = 0;
  const unsigned base = COUNT ? 0u : offset[io];
  const unsigned long long batch = (unsigned long long)(io / nvector), iin = (unsigned long long)(io % nvector);
  auto put = [&](int phase, int sub, int q, long cell1, int var, double v) {
    if (!COUNT) {
      okey[base + n] = (((batch * 2ull + (unsigned)phase) * 12ull + (unsigned)sub) * 4ull + (unsigned)q) * (unsigned long long)nvector + iin;
      tkey[base + n] = (unsigned long long)((long)var * N + cell1 - 1);
      val[base + n] = v;
    }
    n++;
  };
  // Euler system (:1030-1170): per direction, left then right; the face fields ride along with zero fluxes -- x - 0 is x,
  // x + 0 turns a negative zero into a positive one: one such addition per right face and face field
  for (int f = 0; f < 6; f++) {
    const int nb = A.nbor[(long)f * A.ngridmax + g - 1];
    if (A.son[nb - 1] != 0) continue;
    const bool left = (f & 1) == 0;
    for (int v = 0; v < 5; v++)
      for (int q = 0; q < 4; q++) {
        const double t = (COUNT ? 0.0 : A.rec_flux[(((long)io * 6 + f) * 4 + q) * 5 + v]) * 0.125;
        put(0, f, q, nb, v, left ? -t : t);
      }
    if (!left)
      for (int v = 5; v < NF; v++) put(0, f, 0, nb, v, 0.0);
  }
  // induction system (:1172-1457)
  for (int ed = 0; ed < 12; ed++) {
    const EdgeRule &R = EDGE_RULES[ed];
    int b[3], s[3];
    for (int m = 0; m < 3; m++) {
      b[m] = A.nfc[(long)io * 27 + (1 + R.f[m][0]) + 3 * (1 + R.f[m][1]) + 9 * (1 + R.f[m][2])];
      s[m] = A.son[b[m] - 1];
    }
    if (s[0] > 0 && s[2] > 0) continue;
    double weight = 1.0;
    if (s[0] > 0 || s[1] > 0 || s[2] > 0) weight = 0.5;
    const bool leaves = s[0] == 0 && s[1] == 0 && s[2] == 0;
    double dflux = 0.0;
    if (!COUNT) dflux = (A.rec_emf[((long)io * 12 + ed) * 2] + A.rec_emf[((long)io * 12 + ed) * 2 + 1]) * 0.25 * weight;
    for (int k = 0; k < 6; k++) {
      const EdgeUpd &u = R.u[k];
      if (u.half && !leaves) continue;
      const double d = u.half ? dflux * 0.5 : dflux;
      put(1, ed, 0, b[u.buf - 1], u.var, u.sgn > 0 ? d : -d);
    }
  }
  if (COUNT) count[io] = n;
}
__global__ __launch_bounds__(256) void mhd_amr_gather_kernel(const unsigned *__restrict__ perm, const unsigned long long *__restrict__ kin, const double *__restrict__ vin,
                                                             unsigned long long *__restrict__ kout, double *__restrict__ vout, unsigned n) {
  const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  kout[t] = kin[perm[t]];
  vout[t] = vin[perm[t]];
}
__global__ __launch_bounds__(256) void mhd_amr_iota_kernel(unsigned *p, unsigned n) {
  const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n) p[t] = t;
}
// the sorted contributions, added one after the other per (cell, variable) by the thread of the first one
__global__ __launch_bounds__(256) void mhd_amr_apply_kernel(double *__restrict__ unew, const unsigned long long *__restrict__ tkey, const double *__restrict__ val, unsigned n) {
  const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const unsigned long long k = tkey[t];
  if (t > 0 && tkey[t - 1] == k) return;
  double x = unew[k];
  for (unsigned i = t; i < n && tkey[i] == k; i++) x = x + val[i];
  unew[k] = x;
}

struct DBuf {
  void *p = nullptr;
  size_t cap = 0;
  hipError_t ensure(size_t bytes) {
    if (bytes <= cap && p) return hipSuccess;
    if (p) (void)hipFree(p);
    p = nullptr; cap = 0;
    hipError_t e = hipMalloc(&p, bytes ? bytes : 8);
    if (e == hipSuccess) cap = bytes;
    return e;
  }
  template <class T> T *as() { return reinterpret_cast<T *>(p); }
};

int make_const(const ramses_amd_mhd_params *p, MhdConst &P) {
  if (!p) return failf(RAMSES_AMD_EINVAL, "params is NULL");
  P.gamma = p->gamma; P.smallr = p->smallr; P.smallc = p->smallc; P.slope_theta = p->slope_theta;
  P.slope_type = p->slope_type;
  P.slope_mag_type = p->slope_mag_type == -1 ? p->slope_type : p->slope_mag_type;      // hydro/read_hydro_params.f90:528-530
  P.riemann = p->riemann; P.riemann2d = p->riemann2d;
  if (!(p->gamma > 1.0)) return failf(RAMSES_AMD_EINVAL, "gamma must be > 1");
  if (!slope_type_supported(P.slope_type) || !slope_mag_type_supported(P.slope_mag_type))
    return failf(RAMSES_AMD_EUNSUPPORTED, "MHD sweep: slope_type 0, 1, 2, 3, 7, 8 and slope_mag_type 0, 1, 2, 7, 8 are on the device (got %d / %d)", P.slope_type, P.slope_mag_type);
  if (!riemann_supported(P.riemann)) return failf(RAMSES_AMD_EINVAL, "MHD sweep: riemann must be 0 (llf) .. 5 (hydro) (got %d)", P.riemann);
  if (!riemann2d_supported(P.riemann2d)) return failf(RAMSES_AMD_EINVAL, "MHD sweep: riemann2d must be 0 (llf) .. 5 (hlld) (got %d)", P.riemann2d);
  return 0;
}

struct MhdAmrState {
  DBuf uold, unew, f, son, nbor, father, ig, nfc, rflux, remf, err, cnt, off, okey, tkey, val, okey2, tkey2, val2, perm, perm2, tmp;
  int64_t sweeps = 0, octs = 0, ref_sweeps = 0;
  int ref_levels[64] = {0};
};
MhdAmrState g_ma;

// one line at exit: how godunov_fine of the levels of a SOLVER=mhd AMR run was done (always printed when anything was counted:
// a level that went to the reference's host routine must not look like a device run)
void mhd_amr_report(void) {
  if (g_ma.sweeps + g_ma.ref_sweeps == 0) return;
  fprintf(stdout, " ramses_amd: MHD godunov_fine of AMR levels: %ld sweeps on the device (%ld octs), %ld through the reference's host routine",
          (long)g_ma.sweeps, (long)g_ma.octs, (long)g_ma.ref_sweeps);
  if (g_ma.ref_sweeps) {
    fprintf(stdout, " (per level:");
    for (int l = 0; l < 64; l++) if (g_ma.ref_levels[l]) fprintf(stdout, " %d:%d", l, g_ma.ref_levels[l]);
    fprintf(stdout, ")");
  }
  fprintf(stdout, "\n");
  fflush(stdout);
}
void mhd_amr_register(void) {
  static bool registered = false;
  if (!registered) { registered = true; atexit(mhd_amr_report); }
}

}  // namespace

extern "C" {

int64_t ramses_amd_mhd_amr_sweeps(void) { return g_ma.sweeps; }
int64_t ramses_amd_mhd_amr_octs(void) { return g_ma.octs; }
// the drop-in tells when a level takes the reference's godunov_fine instead (what the device path does not cover): counted
// per level and printed in the exit line
int ramses_amd_mhd_note_reference_sweep(int ilevel) {
  mhd_amr_register();
  g_ma.ref_sweeps++;
  if (ilevel >= 0 && ilevel < 64) g_ma.ref_levels[ilevel]++;
  return 0;
}

// godfine1 over the octs d_igrid[0..ngrid) of level ilevel on DEVICE arrays: d_uold / d_unew [11][ncell] (uold(1:ncell,1:nvar+3),
// column major), d_f [3][ncell] or null, the tree arrays son [ncell], nbor [6][ngridmax], father [ngridmax].  unew of the
// listed octs' cells and -- coarse != 0 (ilevel > levelmin) -- of the leaf cells of level ilevel-1 around them is updated in
// the reference's order (nvector: the batch length of godunov_fine's loop, mhd/godunov_fine.f90:23-29).
int ramses_amd_mhd_godfine_amr_device(const ramses_amd_mhd_params *p, int ilevel, int ngrid, const int *d_igrid, const int *d_son, const int *d_nbor,
                                      const int *d_father, int64_t ngridmax, int64_t ncoarse, const double *d_uold, double *d_unew, const double *d_f,
                                      double dx, double dt, int nvector, int interpol_var, int interpol_type, int interpol_mag_type, int coarse,
                                      void *stream) {
  MhdAmrArgs A;
  if (int rc = make_const(p, A.P)) return rc;
  if (ngrid <= 0) return 0;
  if (!d_igrid || !d_son || !d_nbor || !d_father || !d_uold || !d_unew) return failf(RAMSES_AMD_EINVAL, "NULL device pointer");
  if (ilevel < 3) return failf(RAMSES_AMD_EUNSUPPORTED, "MHD godfine1 on the device: levels >= 3 (got %d)", ilevel);
  if (nvector < 1 || nvector > 65536) return failf(RAMSES_AMD_EINVAL, "nvector out of range");
  if (interpol_var < 0 || interpol_var > 1 || interpol_type < 0 || interpol_type > 3 || interpol_mag_type < 0 || interpol_mag_type > 3)
    return failf(RAMSES_AMD_EUNSUPPORTED, "MHD godfine1: interpol_var 0..1, interpol_type 0..3, interpol_mag_type 0..3 (got %d %d %d)", interpol_var, interpol_type, interpol_mag_type);
  if (!(dx > 0.0) || !(dt >= 0.0)) return failf(RAMSES_AMD_EINVAL, "dx must be > 0 and dt >= 0");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  MhdAmrState &S = g_ma;
  A.uold = d_uold; A.unew = d_unew; A.grav = d_f;
  A.son = d_son; A.nbor = d_nbor; A.father = d_father; A.igrid = d_igrid; A.ngrid = ngrid;
  A.ncoarse = ncoarse; A.ngridmax = ngridmax; A.ncell = ncoarse + 8 * ngridmax;
  if ((unsigned long)NF * (unsigned long)A.ncell >= (1ul << 40)) return failf(RAMSES_AMD_EUNSUPPORTED, "cell vectors too long");
  A.dt = dt; A.dx = dx; A.interpol_var = interpol_var; A.interpol_type = interpol_type; A.interpol_mag_type = interpol_mag_type;
  A.coarse = coarse ? 1 : 0;
  HCHK(S.nfc.ensure(sizeof(int) * 27 * (size_t)ngrid), "hipMalloc");
  HCHK(S.rflux.ensure(sizeof(double) * 120 * (size_t)ngrid), "hipMalloc");
  HCHK(S.remf.ensure(sizeof(double) * 24 * (size_t)ngrid), "hipMalloc");
  HCHK(S.err.ensure(sizeof(int)), "hipMalloc");
  A.nfc = S.nfc.as<int>(); A.rec_flux = S.rflux.as<double>(); A.rec_emf = S.remf.as<double>(); A.err = S.err.as<int>();
  HCHK(hipMemsetAsync(A.err, 0, sizeof(int), s), "memset");
  const size_t lds = sizeof(OctLds);
  const bool s3 = A.P.slope_type == 3, gr = d_f != nullptr;
#define MHD_AMR_LAUNCH(S3_, GR_)                                                                                              \
  do {                                                                                                                        \
    auto k = mhd_amr_oct_kernel<S3_, GR_>;                                                                                    \
    HCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), "LDS");  \
    hipLaunchKernelGGL(k, dim3(ngrid), dim3(64), lds, s, A);                                                                  \
  } while (0)
  if (s3) { if (gr) MHD_AMR_LAUNCH(true, true); else MHD_AMR_LAUNCH(true, false); }
  else { if (gr) MHD_AMR_LAUNCH(false, true); else MHD_AMR_LAUNCH(false, false); }
#undef MHD_AMR_LAUNCH
  HCHK(hipGetLastError(), "MHD godfine1 launch");
  dbg_stage("oct kernel");
  int bad = 0;
  HCHK(hipMemcpyAsync(&bad, A.err, sizeof(int), hipMemcpyDeviceToHost, s), "D2H");
  HCHK(hipStreamSynchronize(s), "sync");
  if (bad) return failf(RAMSES_AMD_EINVAL, "level %d: %d father cells needed by an oct do not exist (tree inconsistent)", ilevel, bad);
  mhd_amr_register();
  S.sweeps++; S.octs += ngrid;
  if (!A.coarse) return 0;
  // ---- the coarser level: count, scan, emit, sort by the reference's order, then (stable) by target, add
  HCHK(S.cnt.ensure(sizeof(unsigned) * (size_t)(ngrid + 1)), "hipMalloc");
  HCHK(S.off.ensure(sizeof(unsigned) * (size_t)(ngrid + 1)), "hipMalloc");
  const dim3 go((ngrid + 255) / 256), bo(256);
  HCHK(hipMemsetAsync(S.cnt.p, 0, sizeof(unsigned) * (size_t)(ngrid + 1), s), "memset");
  hipLaunchKernelGGL(mhd_amr_emit_kernel<true>, go, bo, 0, s, A, nvector, (const unsigned *)nullptr, S.cnt.as<unsigned>(), (unsigned long long *)nullptr,
                     (unsigned long long *)nullptr, (double *)nullptr);
  dbg_stage("count");
  size_t tb = 0;
  HCHK(hipcub::DeviceScan::ExclusiveSum(nullptr, tb, S.cnt.as<unsigned>(), S.off.as<unsigned>(), ngrid + 1, s), "scan");
  HCHK(S.tmp.ensure(tb), "hipMalloc");
  HCHK(hipcub::DeviceScan::ExclusiveSum(S.tmp.p, tb, S.cnt.as<unsigned>(), S.off.as<unsigned>(), ngrid + 1, s), "scan");
  unsigned total = 0;
  HCHK(hipMemcpyAsync(&total, S.off.as<unsigned>() + ngrid, sizeof(unsigned), hipMemcpyDeviceToHost, s), "D2H");
  HCHK(hipStreamSynchronize(s), "sync");
  if (total == 0) return 0;
  if ((unsigned long)ngrid * 96ul >= (1ul << 63) / (unsigned long)nvector) return failf(RAMSES_AMD_EUNSUPPORTED, "too many octs for the order key");
  for (DBuf *b : {&S.okey, &S.tkey, &S.okey2, &S.tkey2}) HCHK(b->ensure(sizeof(unsigned long long) * (size_t)total), "hipMalloc");
  for (DBuf *b : {&S.val, &S.val2}) HCHK(b->ensure(sizeof(double) * (size_t)total), "hipMalloc");
  for (DBuf *b : {&S.perm, &S.perm2}) HCHK(b->ensure(sizeof(unsigned) * (size_t)total), "hipMalloc");
  hipLaunchKernelGGL(mhd_amr_emit_kernel<false>, go, bo, 0, s, A, nvector, S.off.as<unsigned>(), (unsigned *)nullptr, S.okey.as<unsigned long long>(),
                     S.tkey.as<unsigned long long>(), S.val.as<double>());
  dbg_stage("emit");
  const dim3 gt((total + 255) / 256);
  hipLaunchKernelGGL(mhd_amr_iota_kernel, gt, bo, 0, s, S.perm.as<unsigned>(), total);
  // by the place in the reference's loops ...
  HCHK(hipcub::DeviceRadixSort::SortPairs(nullptr, tb, S.okey.as<unsigned long long>(), S.okey2.as<unsigned long long>(), S.perm.as<unsigned>(), S.perm2.as<unsigned>(), (int)total, 0, 64, s), "sort");
  HCHK(S.tmp.ensure(tb), "hipMalloc");
  HCHK(hipcub::DeviceRadixSort::SortPairs(S.tmp.p, tb, S.okey.as<unsigned long long>(), S.okey2.as<unsigned long long>(), S.perm.as<unsigned>(), S.perm2.as<unsigned>(), (int)total, 0, 64, s), "sort");
  dbg_stage("sort 1");
  hipLaunchKernelGGL(mhd_amr_gather_kernel, gt, bo, 0, s, S.perm2.as<unsigned>(), S.tkey.as<unsigned long long>(), S.val.as<double>(), S.tkey2.as<unsigned long long>(), S.val2.as<double>(), total);
  // ... then by target; the radix sort is stable, so the first order survives inside every target
  HCHK(hipcub::DeviceRadixSort::SortPairs(nullptr, tb, S.tkey2.as<unsigned long long>(), S.tkey.as<unsigned long long>(), S.val2.as<double>(), S.val.as<double>(), (int)total, 0, 40, s), "sort");
  HCHK(S.tmp.ensure(tb), "hipMalloc");
  HCHK(hipcub::DeviceRadixSort::SortPairs(S.tmp.p, tb, S.tkey2.as<unsigned long long>(), S.tkey.as<unsigned long long>(), S.val2.as<double>(), S.val.as<double>(), (int)total, 0, 40, s), "sort");
  dbg_stage("sort 2");
  hipLaunchKernelGGL(mhd_amr_apply_kernel, gt, bo, 0, s, d_unew, S.tkey.as<unsigned long long>(), S.val.as<double>(), total);
  HCHK(hipGetLastError(), "coarse corrections");
  HCHK(hipStreamSynchronize(s), "sync");
  return 0;
}

// godunov_fine(ilevel) of a SOLVER=mhd run on the reference's own arrays, any level of an AMR tree (staged: the tree and uold
// go up, unew goes up and comes back).  f: the acceleration f(1:ncell,1:3), read when use_f != 0 (poisson).  nvector, interpol_*:
// the reference's parameters of the same names; levelmin: the coarser level takes corrections when ilevel > levelmin.
int ramses_amd_mhd_godunov_fine_amr_f90(const ramses_amd_mhd_params *p, int ilevel, int levelmin, int ngrid, const int *igrid, const int *son,
                                        const int *nbor, const int *father, int64_t ngridmax, int64_t ncoarse, const double *uold, double *unew,
                                        const double *f, int use_f, double dx, double dt, int nvector, int interpol_var, int interpol_type,
                                        int interpol_mag_type) {
  if (ngrid <= 0) return 0;
  if (!use_f) f = nullptr;
  else if (!f) return failf(RAMSES_AMD_EINVAL, "use_f without f");
  if (!igrid || !son || !nbor || !father || !uold || !unew) return failf(RAMSES_AMD_EINVAL, "NULL argument");
  MhdAmrState &S = g_ma;
  const size_t ncell = (size_t)(ncoarse + 8 * ngridmax);
  hipStream_t s = nullptr;
  HCHK(S.uold.ensure(sizeof(double) * NF * ncell), "hipMalloc uold");
  HCHK(S.unew.ensure(sizeof(double) * NF * ncell), "hipMalloc unew");
  HCHK(S.son.ensure(sizeof(int) * ncell), "hipMalloc son");
  HCHK(S.nbor.ensure(sizeof(int) * 6 * (size_t)ngridmax), "hipMalloc nbor");
  HCHK(S.father.ensure(sizeof(int) * (size_t)ngridmax), "hipMalloc father");
  HCHK(S.ig.ensure(sizeof(int) * (size_t)ngrid), "hipMalloc igrid");
  if (f) HCHK(S.f.ensure(sizeof(double) * 3 * ncell), "hipMalloc f");
  HCHK(hipMemcpyAsync(S.uold.p, uold, sizeof(double) * NF * ncell, hipMemcpyHostToDevice, s), "H2D uold");
  HCHK(hipMemcpyAsync(S.unew.p, unew, sizeof(double) * NF * ncell, hipMemcpyHostToDevice, s), "H2D unew");
  HCHK(hipMemcpyAsync(S.son.p, son, sizeof(int) * ncell, hipMemcpyHostToDevice, s), "H2D son");
  HCHK(hipMemcpyAsync(S.nbor.p, nbor, sizeof(int) * 6 * (size_t)ngridmax, hipMemcpyHostToDevice, s), "H2D nbor");
  HCHK(hipMemcpyAsync(S.father.p, father, sizeof(int) * (size_t)ngridmax, hipMemcpyHostToDevice, s), "H2D father");
  HCHK(hipMemcpyAsync(S.ig.p, igrid, sizeof(int) * (size_t)ngrid, hipMemcpyHostToDevice, s), "H2D igrid");
  if (f) HCHK(hipMemcpyAsync(S.f.p, f, sizeof(double) * 3 * ncell, hipMemcpyHostToDevice, s), "H2D f");
  dbg_stage("uploads");
  if (int rc = ramses_amd_mhd_godfine_amr_device(p, ilevel, ngrid, S.ig.as<int>(), S.son.as<int>(), S.nbor.as<int>(), S.father.as<int>(), ngridmax, ncoarse,
                                                 S.uold.as<double>(), S.unew.as<double>(), f ? S.f.as<double>() : nullptr, dx, dt, nvector, interpol_var,
                                                 interpol_type, interpol_mag_type, ilevel > levelmin ? 1 : 0, s)) return rc;
  HCHK(hipMemcpyAsync(unew, S.unew.p, sizeof(double) * NF * ncell, hipMemcpyDeviceToHost, s), "D2H unew");
  HCHK(hipStreamSynchronize(s), "sync");
  return 0;
}

}  // extern "C"

#include "warm.hpp"
RAMSES_AMD_TU_WARM(mhd_amr)
