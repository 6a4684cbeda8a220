// mhd_sweep.hip -- the constrained-transport MHD Godunov sweep of one fully refined periodic level on MI355X (gfx950)
// (SOLVER=mhd of the reference: mhd/godunov_fine.f90 godfine1 :538-1459 on a level without coarse-fine boundaries,
// mhd/umuscl.f90 mag_unsplit :31-238).
//
// Round 4: the FIRST correct path -- one kernel per stage of mag_unsplit over the dense brick, intermediates in HBM:
//   prim     ctoprim (:2029-2186)                                   q[8] per cell
//   efield   trace3d's edge-centred v x B (:811-838)                E[3] per cell (its low x-, y-, z-edge)
//   trace    uslope + trace3d's predictor (:2187-2844, :841-983)    47 numbers per cell: predicted state, face fields, half slopes
//   flux     the traced face states rebuilt (:985-1052) + cmpflxm x 3 (:95-157, :1308-1448): the five Euler fluxes through the three low faces
//   emf      the traced corner states rebuilt (:1054-1276) + cmp_mag_flx x 3 (:160-236, :1453-2028): the EMF on the three low edges
//   update   godfine1's conservative update + constrained transport (mhd/godunov_fine.f90:909-1022), fused with
//            set_unew (unew = uold + ...)
// Every stage calls the functions of mhd_core.hpp / mhd_assemble.hpp, which tests/test_mhd_core_host.py holds bit-exact
// against the compiled reference on the CPU; each cell, face and edge is computed ONCE (the reference recomputes a 6^3
// stencil per oct).  HBM traffic of this version: ~1.6 kB per cell update against 176 B algorithmic (11 fields read and
// written) -- the z-marching LDS pipeline of the hydro sweep (hydro_sweep.hip) is the model for the next step.
//
// Layout: uold / unew = [11][nz][ny][nx] doubles: rho, rho u, rho v, rho w, E, the three LEFT-face fields (uold(:,6:8)),
// the three RIGHT-face fields (uold(:,nvar+1:nvar+3)); periodic in the three directions.  The sweep reads the field of a
// face from the LEFT-face array of the cell above it, which is what ctoprim's bf holds everywhere except on the last
// face of a 6^3 stencil (:2062-2100); ramses_amd_mhd_godunov_brick therefore insists that right(i) == left(i+1) bit for
// bit on entry -- true on any level the scheme itself has advanced.
// Compiled with -ffp-contract=off: IEEE operations in the reference's order.
//
// Compiled a second time (round 6, -DRAMSES_AMD_MHD_FAST_TU -fapprox-func -ffp-contract=fast: mhd_sweep_fast.o) into
// ramses_amd_mhd_godunov_brick_fast: the same kernels with the f64 divisions as v_rcp_f64 + Newton steps and multiply-adds
// contracted -- the IEEE division expansions are a fifth of the flux / EMF kernels' instructions -- held to <= 1e-12 relative
// L-infinity of the reference program (tests/test_mhd_fast_certificate_gpu.py).  Every face flux and every edge EMF is still
// computed ONCE and used by both cells / all four faces around it, so the constrained-transport update keeps div B at rounding
// and the right-face field of a cell stays the left-face field of its neighbour bit for bit.  RAMSES_AMD_MHD_FAST=1 routes
// ramses_amd_mhd_godunov_brick (and with it the drop-in's staged and resident sweeps) there; the default is the strict build.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <utility>
#include <cstdlib>
#include <cstring>

#include "../../include/ramses_amd.h"
#include "mhd_assemble.hpp"
#include "pack_args.hpp"

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
#define HCHK(call, what) do { hipError_t e_ = (call); if (e_ != hipSuccess) return failf(RAMSES_AMD_EHIP, "%s: %s", what, hipGetErrorString(e_)); } while (0)

namespace {

constexpr int NF = 11;          // fields of uold / unew
constexpr int NTR = NPRED;     // numbers the trace leaves per cell (mhd_core.hpp: trace_predict)

struct MhdArgs {
  const double *uold;
  double *unew;
  double *q, *E, *tr, *flux, *emf;
  int *bad;
  int nx, ny, nz;
  long ncell;
  double dt, dx;
  MhdConst P;
};

__device__ __forceinline__ int wrap(int i, int n) { return i < 0 ? i + n : (i >= n ? i - n : i); }

struct Grid {
  int nx, ny, nz;
  __device__ __forceinline__ long at(int i, int j, int k) const {
    return wrap(i, nx) + (long)nx * (wrap(j, ny) + (long)ny * wrap(k, nz));
  }
};
// accessor of mhd_assemble.hpp over the brick
struct DevAcc {
  Grid g;
  const double *qv, *bl, *Ev;
  long ncell;
  __device__ __forceinline__ double q(int n, int i, int j, int k) const { return qv[(long)n * ncell + g.at(i, j, k)]; }
  __device__ __forceinline__ double bf(int c, int i, int j, int k) const { return bl[(long)c * ncell + g.at(i, j, k)]; }
  __device__ __forceinline__ double E(int c, int i, int j, int k) const { return Ev[(long)c * ncell + g.at(i, j, k)]; }
};

#define MHD_CELL_LOOP(A)                                                                                             \
  for (long c_ = (long)blockIdx.x * blockDim.x + threadIdx.x; c_ < (A).ncell; c_ += (long)gridDim.x * blockDim.x)
#define MHD_IJK(A)                                                                \
  const int i = (int)(c_ % (A).nx), j = (int)((c_ / (A).nx) % (A).ny), k = (int)(c_ / ((long)(A).nx * (A).ny))

// The stencil kernels (trace, flux, emf, update) read their neighbours in y and z from rows and planes that other workgroups
// touch: what decides their HBM traffic is whether those lines are still in the L2 of the XCD that asks again.  Workgroup b runs
// on XCD b mod 8, so XCD x is given ONE slab of planes (nz / 8 of them), and inside its slab the cells are visited x first, then
// MHD_YS rows of y, then z, then the next strip of rows: the plane below a cell was visited nx * MHD_YS cells ago (1.5 MB of
// predicted states at 256^3) instead of a whole plane ago (25 MB against 4 MB of L2).  One cell per thread; the launch covers
// 8 x (workgroups of the largest slab).
constexpr int MHD_YS = 16;
struct CellMap {
  int nx, ny, nz, zs;            // zs: planes per XCD slab
  __host__ __device__ long slab_cells(int x) const { const int z0 = x * zs, z1 = z0 + zs < nz ? z0 + zs : nz; return z1 > z0 ? (long)nx * ny * (z1 - z0) : 0; }
};
__device__ __forceinline__ bool mhd_cell_of(const CellMap &M, int tpb, int &i, int &j, int &k, long &c) {
  const int xcd = blockIdx.x & 7;
  const long p = (long)(blockIdx.x >> 3) * tpb + threadIdx.x;
  const int z0 = xcd * M.zs, nzs = (z0 + M.zs < M.nz ? z0 + M.zs : M.nz) - z0;
  if (nzs <= 0 || p >= (long)M.nx * M.ny * nzs) return false;
  const long strip = (long)M.nx * MHD_YS * nzs;
  const int ys = (int)(p / strip);
  const long q = p - (long)ys * strip;
  const int sy = M.ny - ys * MHD_YS < MHD_YS ? M.ny - ys * MHD_YS : MHD_YS;
  i = (int)(q % M.nx);
  j = ys * MHD_YS + (int)((q / M.nx) % sy);
  k = z0 + (int)(q / ((long)M.nx * sy));
  c = i + (long)M.nx * (j + (long)M.ny * k);
  return true;
}
#define MHD_CELL_REMAP(A, TPB)                                           \
  const CellMap cm_{(A).nx, (A).ny, (A).nz, ((A).nz + 7) / 8};           \
  int i, j, k;                                                           \
  long c_;                                                               \
  if (mhd_cell_of(cm_, TPB, i, j, k, c_))

__global__ __launch_bounds__(256) void mhd_prim_kernel(MhdArgs A) {
  const Grid g{A.nx, A.ny, A.nz};
  int bad = 0;
  MHD_CELL_LOOP(A) {
    MHD_IJK(A);
    const long N = A.ncell;
    const double u[5] = {A.uold[c_], A.uold[N + c_], A.uold[2 * N + c_], A.uold[3 * N + c_], A.uold[4 * N + c_]};
    const double bl[3] = {A.uold[5 * N + c_], A.uold[6 * N + c_], A.uold[7 * N + c_]};
    const double br[3] = {A.uold[8 * N + c_], A.uold[9 * N + c_], A.uold[10 * N + c_]};
    double q[8];
    ctoprim_cell(u, bl, br, nullptr, A.dt, A.P, q);
#pragma unroll
    for (int n = 0; n < 8; n++) A.q[(long)n * N + c_] = q[n];
    // the right faces must be the neighbours' left faces (see the header)
    // (by VALUE: +0.0 and -0.0 are the same field)
    if (!(br[0] == A.uold[5 * N + g.at(i + 1, j, k)])) bad++;
    if (!(br[1] == A.uold[6 * N + g.at(i, j + 1, k)])) bad++;
    if (!(br[2] == A.uold[7 * N + g.at(i, j, k + 1)])) bad++;
  }
  if (bad) atomicAdd(A.bad, bad);
}

__global__ __launch_bounds__(256) void mhd_efield_kernel(MhdArgs A) {
  DevAcc a{{A.nx, A.ny, A.nz}, A.q, A.uold + 5 * A.ncell, nullptr, A.ncell};
  MHD_CELL_LOOP(A) {
    MHD_IJK(A);
#pragma unroll
    for (int c = 0; c < 3; c++) A.E[(long)c * A.ncell + c_] = efield(a, c, i, j, k);
  }
}

// what the trace leaves in HBM: the 47 numbers of trace_predict per cell (mhd_core.hpp), the cells in groups of 64 (one
// wavefront of the trace kernel): number n of cell c at tr[(c / 64) * 47 * 64 + n * 64 + c % 64].  A wavefront writes ONE
// contiguous 24 KB piece, the 47 addresses of a cell differ by constants, and what the flux and EMF kernels read of a cell
// and of its x neighbour sits in the same piece (47 separate planes, the first layout: 12.2 instead of 10.0 ms per sweep at
// 256^3, profiles/r04_mhd_prof.txt)
__device__ __forceinline__ long pred_at(int n, long cell) { return (cell >> 6) * (long)(NPRED * 64) + (long)n * 64 + (cell & 63); }
struct PredSrc {
  const double *pr;
  long cell;
  __device__ __forceinline__ double c(int n) const { return pr[pred_at(n, cell)]; }
  __device__ __forceinline__ double f(int n) const { return pr[pred_at(8 + n, cell)]; }
  __device__ __forceinline__ double h(int n) const { return pr[pred_at(14 + n, cell)]; }
};
template <bool S3>
__global__ __launch_bounds__(128) void mhd_trace_kernel(MhdArgs A) {
  DevAcc a{{A.nx, A.ny, A.nz}, A.q, A.uold + 5 * A.ncell, A.E, A.ncell};
  const double dtdx = A.dt / A.dx;
  MHD_CELL_REMAP(A, 128) {
    TraceIn I;
    trace_inputs<S3>(a, i, j, k, A.P, I);
    TracePred T;
    trace_predict(I, dtdx, dtdx, dtdx, A.P, T);
#pragma unroll
    for (int n = 0; n < NPRED; n++) A.tr[pred_at(n, c_)] = T.v[n];
  }
}

// ---- ctoprim + the edge fields + uslope / trace3d's predictor in ONE launch (round 6) ---------------------------------------
// The three kernels above hand q (8 numbers per cell) and E (3) through HBM and the trace then gathers them back: seven (or 27)
// cells x 8 + 30 face fields + 12 edge fields per cell through the L1 -- 4.1 of the sweep's 10.2 ms at 256^3 for 0.9 ms of
// arithmetic.  Here a workgroup owns a tile of 32 x 4 x 4 cells: it converts the 34 x 6 x 6 cells around it once into LDS (q and
// the left-face fields: 108 KB), forms the edge fields of the 33 x 5 x 5 edges the tile's traces read (20 KB), and every thread
// traces its cell from LDS.  What reaches HBM is what the flux and EMF kernels need: the 47 predicted numbers per cell.  The
// same functions on the same values as the three kernels (which stay for the A/B: RAMSES_AMD_MHD_FUSED=0).
// (tile shapes measured at 256^3, round 6, ms per sweep: 32x4x4 8.89, 16x4x4 9.13, 16x8x4 9.12, 16x4x8 9.23, 32x4x2 10.43;
//  profiles/r06_mhd_pmc.txt)
#ifndef MHD_FT
#define MHD_FT 32, 4, 4
#endif
constexpr int FT_DIMS[3] = {MHD_FT};
constexpr int FT_X = FT_DIMS[0], FT_Y = FT_DIMS[1], FT_Z = FT_DIMS[2];
#ifndef MHD_BATCH_LOADS
#define MHD_BATCH_LOADS 1
#endif
struct FusedLds {
  double Q[8][FT_Z + 2][FT_Y + 2][FT_X + 2];
  double BF[3][FT_Z + 2][FT_Y + 2][FT_X + 2];
  double Ef[3][FT_Z + 1][FT_Y + 1][FT_X + 1];
};
struct TileAcc {
  const FusedLds *L;
  int ox, oy, oz;             // global coordinates of the tile's first interior cell
  __device__ __forceinline__ double q(int n, int i, int j, int k) const { return L->Q[n][k - oz + 1][j - oy + 1][i - ox + 1]; }
  __device__ __forceinline__ double bf(int c, int i, int j, int k) const { return L->BF[c][k - oz + 1][j - oy + 1][i - ox + 1]; }
  __device__ __forceinline__ double E(int c, int i, int j, int k) const { return L->Ef[c][k - oz][j - oy][i - ox]; }
};
__device__ __forceinline__ int wrapn(int i, int n) { i %= n; return i < 0 ? i + n : i; }
template <bool S3>
__global__ __launch_bounds__(FT_X * FT_Y * FT_Z) void mhd_prim_trace_kernel(MhdArgs A, int ntx, int nty) {
  extern __shared__ __attribute__((aligned(16))) unsigned char fused_raw[];
  FusedLds &L = *reinterpret_cast<FusedLds *>(fused_raw);
  const int tid = threadIdx.x;
  const int bx = blockIdx.x % ntx, by = (blockIdx.x / ntx) % nty, bz = blockIdx.x / (ntx * nty);
  const int ox = bx * FT_X, oy = by * FT_Y, oz = bz * FT_Z;
  const long N = A.ncell;
  // ---- 1. ctoprim of the tile and one cell around it
  constexpr int HX = FT_X + 2, HY = FT_Y + 2, HZ = FT_Z + 2;
  // (all the loads of the thread's two or three cells are issued before the first conversion: one block of eight wavefronts
  //  owns the CU, nothing else hides the latency of this phase)
  constexpr int NT = FT_X * FT_Y * FT_Z, NIT = (HX * HY * HZ + NT - 1) / NT;
#if MHD_BATCH_LOADS
  {
    double w[NIT][NF];
#pragma unroll
    for (int it = 0; it < NIT; it++) {
      const int t = min(tid + it * NT, HX * HY * HZ - 1);
      const int li = t % HX, lj = (t / HX) % HY, lk = t / (HX * HY);
      const long c = wrapn(ox + li - 1, A.nx) + (long)A.nx * (wrapn(oy + lj - 1, A.ny) + (long)A.ny * wrapn(oz + lk - 1, A.nz));
#pragma unroll
      for (int n = 0; n < NF; n++) w[it][n] = A.uold[(long)n * N + c];
    }
#pragma unroll
    for (int it = 0; it < NIT; it++) {
      const int t = tid + it * NT;
      if (t < HX * HY * HZ) {
        const int li = t % HX, lj = (t / HX) % HY, lk = t / (HX * HY);
        const double u[5] = {w[it][0], w[it][1], w[it][2], w[it][3], w[it][4]};
        const double bl[3] = {w[it][5], w[it][6], w[it][7]};
        const double br[3] = {w[it][8], w[it][9], w[it][10]};
        double q[8];
        ctoprim_cell(u, bl, br, nullptr, A.dt, A.P, q);
#pragma unroll
        for (int n = 0; n < 8; n++) L.Q[n][lk][lj][li] = q[n];
#pragma unroll
        for (int n = 0; n < 3; n++) L.BF[n][lk][lj][li] = bl[n];
      }
    }
  }
#else
  for (int t = tid; t < HX * HY * HZ; t += NT) {
    const int li = t % HX, lj = (t / HX) % HY, lk = t / (HX * HY);
    const long c = wrapn(ox + li - 1, A.nx) + (long)A.nx * (wrapn(oy + lj - 1, A.ny) + (long)A.ny * wrapn(oz + lk - 1, A.nz));
    const double u[5] = {A.uold[c], A.uold[N + c], A.uold[2 * N + c], A.uold[3 * N + c], A.uold[4 * N + c]};
    const double bl[3] = {A.uold[5 * N + c], A.uold[6 * N + c], A.uold[7 * N + c]};
    const double br[3] = {A.uold[8 * N + c], A.uold[9 * N + c], A.uold[10 * N + c]};
    double q[8];
    ctoprim_cell(u, bl, br, nullptr, A.dt, A.P, q);
#pragma unroll
    for (int n = 0; n < 8; n++) L.Q[n][lk][lj][li] = q[n];
#pragma unroll
    for (int n = 0; n < 3; n++) L.BF[n][lk][lj][li] = bl[n];
  }
#endif
  __syncthreads();
  const TileAcc a{&L, ox, oy, oz};
  // ---- 2. the edge fields on the low edges of the cells (ox .. ox+FT_X) x (oy .. oy+FT_Y) x (oz .. oz+FT_Z)
  constexpr int EX = FT_X + 1, EY = FT_Y + 1, EZ = FT_Z + 1;
  for (int t = tid; t < 3 * EX * EY * EZ; t += FT_X * FT_Y * FT_Z) {
    const int c = t / (EX * EY * EZ), r = t % (EX * EY * EZ);
    const int li = r % EX, lj = (r / EX) % EY, lk = r / (EX * EY);
    L.Ef[c][lk][lj][li] = efield(a, c, ox + li, oy + lj, oz + lk);
  }
  __syncthreads();
  // ---- 3. the trace of the thread's cell; the face-consistency check of the prim kernel
  const int i = ox + tid % FT_X, j = oy + (tid / FT_X) % FT_Y, k = oz + tid / (FT_X * FT_Y);
  if (i >= A.nx || j >= A.ny || k >= A.nz) return;
  const long c_ = i + (long)A.nx * (j + (long)A.ny * k);
  {
    int bad = 0;
    if (!(A.uold[8 * N + c_] == a.bf(0, i + 1, j, k))) bad++;
    if (!(A.uold[9 * N + c_] == a.bf(1, i, j + 1, k))) bad++;
    if (!(A.uold[10 * N + c_] == a.bf(2, i, j, k + 1))) bad++;
    if (bad) atomicAdd(A.bad, bad);
  }
  TraceIn I;
  trace_inputs<S3>(a, i, j, k, A.P, I);
  TracePred T;
  const double dtdx = A.dt / A.dx;
  trace_predict(I, dtdx, dtdx, dtdx, A.P, T);
#pragma unroll
  for (int n = 0; n < NPRED; n++) A.tr[pred_at(n, c_)] = T.v[n];
}

// A/B knob (round 6, measured and OFF): a traced state of the cell to the LEFT (i-1) of a cell whose own numbers the thread reads
// anyway can come from the lane of that cell by a wavefront shift instead of from the left cell's predicted numbers -- three of
// the seven cells around an edge triple and one of the four around a face triple are then never loaded (EMF 228 -> 160 wave-wide
// loads per cell, flux 90 -> 70); the first lane of a wave and the first cell of a row rebuild the state from memory as before.
// Same function on the same numbers: same bits (the parity tests pass with it).  At 256^3 it is SLOWER: EMF 3.63 -> 3.80 ms,
// flux 1.72 -> 1.86, the sweep 8.78 -> 9.27 ms strict, 8.09 -> 8.40 fast (profiles/r06_mhd_xshare.txt) -- the one-lane pass of
// every wave and the shifts in front of the solver cost more than the loads they replace; the kernels are not bound by the
// number of loads.
#ifndef MHD_XSHARE
#define MHD_XSHARE 0
#endif
__device__ __forceinline__ double mhd_wave_shr1(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(lo, lo, 0x138, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(hi, hi, 0x138, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
template <int KIND, int D>
__device__ __forceinline__ void state_of_left(const PredSrc &own, const double *tr, long left_cell, bool left_in_wave, const MhdConst &P,
                                              double (&s)[8]) {
#if MHD_XSHARE
  double t[8];
  trace_state<KIND, D>(own, P, t);
#pragma unroll
  for (int n = 0; n < 8; n++) s[n] = mhd_wave_shr1(t[n]);
  if (!left_in_wave) {
    const PredSrc l{tr, left_cell};
    trace_state<KIND, D>(l, P, s);
  }
#else
  const PredSrc l{tr, left_cell};
  trace_state<KIND, D>(l, P, s);
#endif
}

// flux[d][0..4] through the LOW face of direction d of the cell, scaled as mag_unsplit does (fx*dt/dx, :105-111): the +d
// state of the cell below and the -d state of the cell, each rebuilt from its predicted state and half slopes
template <int D, int RS>
__device__ __forceinline__ void mhd_face_flux(const MhdArgs &A, const Grid &g, int i, int j, int k, long c_, bool left_in_wave) {
  const PredSrc me{A.tr, c_};
  double qm_[8], qp_[8], f[8];
  if constexpr (D == 0) state_of_left<T_QM, 0>(me, A.tr, g.at(i - 1, j, k), left_in_wave, A.P, qm_);
  else {
    const PredSrc lo{A.tr, g.at(i, j - (D == 1), k - (D == 2))};
    trace_state<T_QM, D>(lo, A.P, qm_);
  }
  trace_state<T_QP, D>(me, A.P, qp_);
  cmpflxm_face<RS>(qm_, qp_, D, A.P, f);
#pragma unroll
  for (int n = 0; n < 5; n++) A.flux[(long)(D * 5 + n) * A.ncell + c_] = f[n] * A.dt / A.dx;
}
template <int RS>
__global__ __launch_bounds__(128) void mhd_flux_kernel(MhdArgs A) {
  const Grid g{A.nx, A.ny, A.nz};
  MHD_CELL_REMAP(A, 128) {
    const bool left_in_wave = (threadIdx.x & 63) > 0 && i > 0;      // (consecutive threads of a block: consecutive cells of a row)
    mhd_face_flux<0, RS>(A, g, i, j, k, c_, left_in_wave);
    mhd_face_flux<1, RS>(A, g, i, j, k, c_, left_in_wave);
    mhd_face_flux<2, RS>(A, g, i, j, k, c_, left_in_wave);
  }
}

// emf[e] on the LOW edge of direction e of the cell, scaled as mag_unsplit does (emf*dt/dx, :171-177); which cell's corner
// state plays which part: mhd_assemble.hpp edge_sources
template <int E, int R2>
__device__ __forceinline__ void mhd_edge_emf(const MhdArgs &A, const Grid &g, int i, int j, int k, long c_, bool left_in_wave) {
  double rt[8], rb[8], lt[8], lb[8];
  using Src = PredSrc;
  const Src me{A.tr, c_};
  if constexpr (E == 2) {
    // a = (i-1, j-1, k): the left neighbour of c; b = (i-1, j, k): the left neighbour of the cell itself
    const Src c{A.tr, g.at(i, j - 1, k)};
    state_of_left<T_QRT, 2>(c, A.tr, g.at(i - 1, j - 1, k), left_in_wave, A.P, rt);
    state_of_left<T_QRB, 2>(me, A.tr, g.at(i - 1, j, k), left_in_wave, A.P, rb);
    trace_state<T_QLT, 2>(c, A.P, lt);
  } else if constexpr (E == 1) {
    // a = (i-1, j, k-1): the left neighbour of b; c = (i-1, j, k): the left neighbour of the cell itself
    const Src b{A.tr, g.at(i, j, k - 1)};
    state_of_left<T_QRT, 1>(b, A.tr, g.at(i - 1, j, k - 1), left_in_wave, A.P, rt);
    trace_state<T_QLT, 1>(b, A.P, rb);
    state_of_left<T_QRB, 1>(me, A.tr, g.at(i - 1, j, k), left_in_wave, A.P, lt);
  } else {
    const Src a{A.tr, g.at(i, j - 1, k - 1)}, b{A.tr, g.at(i, j - 1, k)}, c{A.tr, g.at(i, j, k - 1)};
    trace_state<T_QRT, 0>(a, A.P, rt); trace_state<T_QRB, 0>(b, A.P, rb); trace_state<T_QLT, 0>(c, A.P, lt);
  }
  trace_state<T_QLB, E>(me, A.P, lb);
  A.emf[(long)E * A.ncell + c_] = cmp_mag_flx_edge<R2>(rt, rb, lt, lb, E, A.P) * A.dt / A.dx;
}
template <int R2>
__global__ __launch_bounds__(128) void mhd_emf_kernel(MhdArgs A) {
  const Grid g{A.nx, A.ny, A.nz};
  MHD_CELL_REMAP(A, 128) {
    const bool left_in_wave = (threadIdx.x & 63) > 0 && i > 0;
    mhd_edge_emf<0, R2>(A, g, i, j, k, c_, left_in_wave);
    mhd_edge_emf<1, R2>(A, g, i, j, k, c_, left_in_wave);
    mhd_edge_emf<2, R2>(A, g, i, j, k, c_, left_in_wave);
  }
}

// set_unew + godfine1's update of a level without coarse-fine boundaries (mhd/godunov_fine.f90:909-1022)
__global__ __launch_bounds__(256) void mhd_update_kernel(MhdArgs A) {
  const Grid g{A.nx, A.ny, A.nz};
  const long N = A.ncell;
  MHD_CELL_REMAP(A, 256) {
    const long cx = g.at(i + 1, j, k), cy = g.at(i, j + 1, k), cz = g.at(i, j, k + 1);
    // Euler system: ((u + (Fx- - Fx+)) + (Fy- - Fy+)) + (Fz- - Fz+)
#pragma unroll
    for (int n = 0; n < 5; n++) {
      double un = A.uold[(long)n * N + c_];
      un = un + (A.flux[(long)(0 * 5 + n) * N + c_] - A.flux[(long)(0 * 5 + n) * N + cx]);
      un = un + (A.flux[(long)(1 * 5 + n) * N + c_] - A.flux[(long)(1 * 5 + n) * N + cy]);
      un = un + (A.flux[(long)(2 * 5 + n) * N + c_] - A.flux[(long)(2 * 5 + n) * N + cz]);
      A.unew[(long)n * N + c_] = un;
    }
    // the six face fields take part in that loop with their Euler fluxes reset to zero (:801-903): b + (0 - 0), three times
    double b[6];
#pragma unroll
    for (int n = 0; n < 6; n++) {
      double v = A.uold[(long)(5 + n) * N + c_];
      const double z = 0.0;
      v = v + (z - z); v = v + (z - z); v = v + (z - z);
      b[n] = v;
    }
    // induction system, constrained transport (:966-1022); emf*(i3,j3,k3) = the EMF on the low edge of that cell
    const double *ex = A.emf, *ey = A.emf + N, *ez = A.emf + 2 * N;
    const long cxy = g.at(i + 1, j + 1, k), cxz = g.at(i + 1, j, k + 1), cyz = g.at(i, j + 1, k + 1);
    double df;
    df = (ey[c_] - ey[cz]) - (ez[c_] - ez[cy]);          b[0] = b[0] + df;     // Bx, left face
    df = (ey[cx] - ey[cxz]) - (ez[cx] - ez[cxy]);        b[3] = b[3] + df;     // Bx, right face
    df = (ez[c_] - ez[cx]) - (ex[c_] - ex[cz]);          b[1] = b[1] + df;     // By, left
    df = (ez[cy] - ez[cxy]) - (ex[cy] - ex[cyz]);        b[4] = b[4] + df;     // By, right
    df = (ex[c_] - ex[cy]) - (ey[c_] - ey[cx]);          b[2] = b[2] + df;     // Bz, left
    df = (ex[cz] - ex[cyz]) - (ey[cz] - ey[cxz]);        b[5] = b[5] + df;     // Bz, right
#pragma unroll
    for (int n = 0; n < 6; n++) A.unew[(long)(5 + n) * N + c_] = b[n];
  }
}

#ifndef RAMSES_AMD_MHD_FAST_TU
// courant_fine of the resident level (mhd/courant_fine.f90:56-146): the minimum of cmpdt's cell time steps (exact) and the
// four sums of the conservation diagnostics -- mass, total energy, internal energy, magnetic energy -- as a fixed two-stage
// tree (they feed the printed mass / energy balance only: amr/update_time.f90; deterministic, equal to the reference's serial
// sums up to rounding).  partial[block][5] -> courant_final_kernel.
constexpr int COUR_BLOCKS = 1024;
__global__ __launch_bounds__(256) void mhd_courant_kernel(MhdArgs A, double vol, double courant_factor, double *__restrict__ partial) {
  const long N = A.ncell;
  double dt = 1.0e300, mass = 0.0, etot = 0.0, eint = 0.0, emag = 0.0;
  MHD_CELL_LOOP(A) {
    double u[11];
#pragma unroll
    for (int n = 0; n < 11; n++) u[n] = A.uold[(long)n * N + c_];
    dt = fmin2(dt, cmpdt_cell(u, A.dx, courant_factor, A.P));
    mass = mass + u[0] * vol;
    etot = etot + u[4] * vol;
    double ei = u[4] * vol, em = 0.0;
#pragma unroll
    for (int d = 0; d < 3; d++) {
      const double b = u[5 + d] + u[8 + d];
      em = em + 0.125 * (b * b) * vol;
      ei = ei - 0.5 * (u[1 + d] * u[1 + d]) / u[0] * vol - 0.125 * (b * b) * vol;
    }
    eint = eint + ei;
    emag = emag + em;
  }
  __shared__ double red[256][5];
  red[threadIdx.x][0] = dt; red[threadIdx.x][1] = mass; red[threadIdx.x][2] = etot; red[threadIdx.x][3] = eint; red[threadIdx.x][4] = emag;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) {
      red[threadIdx.x][0] = fmin2(red[threadIdx.x][0], red[threadIdx.x + s][0]);
#pragma unroll
      for (int q = 1; q < 5; q++) red[threadIdx.x][q] = red[threadIdx.x][q] + red[threadIdx.x + s][q];
    }
    __syncthreads();
  }
  if (threadIdx.x < 5) partial[(long)blockIdx.x * 5 + threadIdx.x] = red[0][threadIdx.x];
}
__global__ __launch_bounds__(256) void mhd_courant_final_kernel(const double *__restrict__ partial, int nblocks, double *__restrict__ out5) {
  __shared__ double red[256][5];
  double v[5] = {1.0e300, 0.0, 0.0, 0.0, 0.0};
  for (int b = threadIdx.x; b < nblocks; b += 256) {
    v[0] = fmin2(v[0], partial[(long)b * 5]);
#pragma unroll
    for (int q = 1; q < 5; q++) v[q] = v[q] + partial[(long)b * 5 + q];
  }
#pragma unroll
  for (int q = 0; q < 5; q++) red[threadIdx.x][q] = v[q];
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) {
      red[threadIdx.x][0] = fmin2(red[threadIdx.x][0], red[threadIdx.x + s][0]);
#pragma unroll
      for (int q = 1; q < 5; q++) red[threadIdx.x][q] = red[threadIdx.x][q] + red[threadIdx.x + s][q];
    }
    __syncthreads();
  }
  if (threadIdx.x < 5) out5[threadIdx.x] = red[0][threadIdx.x];
}

#endif  // !RAMSES_AMD_MHD_FAST_TU

inline int grid_for(long n, int block) {
  long g = (n + block - 1) / block;
  if (g < 1) g = 1;
  if (g > 65536) g = 65536;
  return (int)g;
}

int make_const(const ramses_amd_mhd_params *p, MhdConst &P) {
  if (!p) return failf(RAMSES_AMD_EINVAL, "params is NULL");
  P.gamma = p->gamma; P.smallr = p->smallr; P.smallc = p->smallc; P.slope_theta = p->slope_theta;
  P.slope_type = p->slope_type;
  P.slope_mag_type = p->slope_mag_type == -1 ? p->slope_type : p->slope_mag_type;      // hydro/read_hydro_params.f90:528-530
  P.riemann = p->riemann; P.riemann2d = p->riemann2d;
  if (!(p->gamma > 1.0)) return failf(RAMSES_AMD_EINVAL, "gamma must be > 1");
  if (!slope_type_supported(P.slope_type) || !slope_mag_type_supported(P.slope_mag_type))
    return failf(RAMSES_AMD_EUNSUPPORTED, "MHD sweep: slope_type 0, 1, 2, 3, 7, 8 and slope_mag_type 0, 1, 2, 7, 8 are on the device (got %d / %d)", P.slope_type, P.slope_mag_type);
  if (!riemann_supported(P.riemann))
    return failf(RAMSES_AMD_EINVAL, "MHD sweep: riemann must be 0 (llf) .. 5 (hydro) (got %d)", P.riemann);
  if (!riemann2d_supported(P.riemann2d))
    return failf(RAMSES_AMD_EINVAL, "MHD sweep: riemann2d must be 0 (llf) .. 5 (hlld) (got %d)", P.riemann2d);
  return 0;
}

constexpr long WORK_DOUBLES_PER_CELL = 8 + 3 + NTR + 15 + 3;

}  // namespace

extern "C" {

#ifndef RAMSES_AMD_MHD_FAST_TU
int64_t ramses_amd_mhd_workspace_bytes(int nx, int ny, int nz) {
  if (nx < 1 || ny < 1 || nz < 1) return failf(RAMSES_AMD_EINVAL, "bad brick extents");
  return (int64_t)sizeof(double) * (WORK_DOUBLES_PER_CELL * nx * ny * nz + NTR * 64) + 256;   // (the trace's last block of 64 cells)
}
#define MHD_BRICK_FN ramses_amd_mhd_godunov_brick
#else
#define MHD_BRICK_FN ramses_amd_mhd_godunov_brick_fast
#endif

// One MHD sweep of a periodic nx x ny x nz level: d_unew = d_uold advanced by dt (set_unew + godunov_fine of SOLVER=mhd).
// d_uold / d_unew: [11][nz][ny][nx] device doubles (see the header of this file), distinct buffers.
int MHD_BRICK_FN(const ramses_amd_mhd_params *p, int nx, int ny, int nz, const double *d_uold, double *d_unew,
                 double dx, double dt, void *d_work, int64_t work_bytes, void *stream) {
#ifndef RAMSES_AMD_MHD_FAST_TU
  {      // (read on every call: the certificate test and the bench flip it inside one process)
    const char *e = getenv("RAMSES_AMD_MHD_FAST");
    if (e && e[0] == '1') return ramses_amd_mhd_godunov_brick_fast(p, nx, ny, nz, d_uold, d_unew, dx, dt, d_work, work_bytes, stream);
  }
#endif
  MhdArgs A;
  if (int rc = make_const(p, A.P)) return rc;
  if (nx < 4 || ny < 4 || nz < 4) return failf(RAMSES_AMD_EINVAL, "MHD sweep: the periodic brick needs at least 4 cells per direction (got %d %d %d)", nx, ny, nz);
  if (!d_uold || !d_unew || !d_work) return failf(RAMSES_AMD_EINVAL, "NULL device pointer");
  if (d_uold == d_unew) return failf(RAMSES_AMD_EINVAL, "uold and unew must be distinct buffers");
  if (!(dx > 0.0) || !(dt >= 0.0)) return failf(RAMSES_AMD_EINVAL, "dx must be > 0 and dt >= 0");
  if (work_bytes < ramses_amd_mhd_workspace_bytes(nx, ny, nz)) return failf(RAMSES_AMD_EINVAL, "MHD sweep: workspace too small (ramses_amd_mhd_workspace_bytes)");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const long N = (long)nx * ny * nz;
  A.uold = d_uold; A.unew = d_unew;
  double *w = reinterpret_cast<double *>(d_work);
  A.q = w; w += 8 * N;
  A.E = w; w += 3 * N;
  A.tr = w; w += (long)NTR * ((N + 63) / 64 * 64);
  A.flux = w; w += 15 * N;
  A.emf = w; w += 3 * N;
  A.bad = reinterpret_cast<int *>(w);
  A.nx = nx; A.ny = ny; A.nz = nz; A.ncell = N; A.dt = dt; A.dx = dx;
  HCHK(hipMemsetAsync(A.bad, 0, sizeof(int), s), "memset");
  // ctoprim, the edge fields and the trace: one launch over tiles of 32 x 4 x 4 cells (RAMSES_AMD_MHD_FUSED=0: the three kernels)
  static int fused = -1;
  if (fused < 0) { const char *e = getenv("RAMSES_AMD_MHD_FUSED"); fused = (e && e[0] == '0') ? 0 : 1; }
  if (!fused) {
    hipLaunchKernelGGL(mhd_prim_kernel, dim3(grid_for(N, 256)), dim3(256), 0, s, A);
    hipLaunchKernelGGL(mhd_efield_kernel, dim3(grid_for(N, 256)), dim3(256), 0, s, A);
  }
  // (the stencil kernels: one cell per thread in the XCD-slab order of mhd_cell_of)
  const CellMap cm{nx, ny, nz, (nz + 7) / 8};
  auto remap_grid = [&](int tpb) { return dim3((unsigned)(8 * ((cm.slab_cells(0) + tpb - 1) / tpb))); };
  if (remap_grid(128).x > 0x7fffffffu / 2) return failf(RAMSES_AMD_EUNSUPPORTED, "MHD sweep: level too large for one launch");
  if (fused) {
    const int ntx = (nx + FT_X - 1) / FT_X, nty = (ny + FT_Y - 1) / FT_Y, ntz = (nz + FT_Z - 1) / FT_Z;
    const dim3 fg((unsigned)((long)ntx * nty * ntz)), fb(FT_X * FT_Y * FT_Z);
    const size_t lds = sizeof(FusedLds);
    if (A.P.slope_type == 3) {
      auto kf = mhd_prim_trace_kernel<true>;
      HCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(kf), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), "LDS");
      hipLaunchKernelGGL(kf, fg, fb, lds, s, A, ntx, nty);
    } else {
      auto kf = mhd_prim_trace_kernel<false>;
      HCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(kf), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), "LDS");
      hipLaunchKernelGGL(kf, fg, fb, lds, s, A, ntx, nty);
    }
  } else if (A.P.slope_type == 3) hipLaunchKernelGGL(mhd_trace_kernel<true>, remap_grid(128), dim3(128), 0, s, A);
  else hipLaunchKernelGGL(mhd_trace_kernel<false>, remap_grid(128), dim3(128), 0, s, A);
  // the flux / EMF kernels: one general instance (the solver is a run-time switch) and one for the Roe solver, whose
  // eigenmatrices would otherwise cost every solver its registers.  (One instance per solver was measured too: fewer
  // registers -- hlld: flux 136 instead of 156, EMF 188 instead of 214 -- but 10.95 instead of 10.08 ms per sweep at 256^3.)
  const dim3 g128(remap_grid(128)), b128(128);
  // (fluxes and EMFs in ONE launch -- the predicted numbers read once -- was measured in round 6: 256 VGPRs + 44 B of scratch at two
  //  waves per SIMD, 9.33 against 8.89 ms; the EMF kernel compiled for three waves per SIMD: 124 B of scratch, 9.83 ms)
  if (A.P.riemann == RIEMANN_ROE) hipLaunchKernelGGL(mhd_flux_kernel<RIEMANN_ROE>, g128, b128, 0, s, A);
  else hipLaunchKernelGGL(mhd_flux_kernel<-2>, g128, b128, 0, s, A);
  // (one launch per edge direction -- 164 VGPRs and three waves per SIMD instead of 212 and two -- was measured: 9.00 against 8.87 ms)
  if (A.P.riemann2d == RIEMANN2D_ROE) hipLaunchKernelGGL(mhd_emf_kernel<RIEMANN2D_ROE>, g128, b128, 0, s, A);
  else hipLaunchKernelGGL(mhd_emf_kernel<-2>, g128, b128, 0, s, A);
  hipLaunchKernelGGL(mhd_update_kernel, remap_grid(256), dim3(256), 0, s, A);
  HCHK(hipGetLastError(), "MHD sweep launch");
  int bad = 0;
  HCHK(hipMemcpyAsync(&bad, A.bad, sizeof(int), hipMemcpyDeviceToHost, s), "D2H");
  HCHK(hipStreamSynchronize(s), "sync");
  if (bad) return failf(RAMSES_AMD_EINVAL, "MHD sweep: %d right-face fields differ from the neighbour's left-face field (uold(:,nvar+1:nvar+3) vs uold(:,6:8))", bad);
  return 0;
}

#ifndef RAMSES_AMD_MHD_FAST_TU
// godunov_fine(ilevel) of a SOLVER=mhd run on the reference's own arrays (staged: uold(1:ncell,1:nvar+3) of the level's
// cells goes up, unew comes back): fully refined periodic level of a single-rank run, nx = ny = nz = 1, NVAR = 8.
// The caller (ramses_amd/patch_mhd/godunov_fine.f90) keeps set_unew / set_uold and everything else of the reference.
int ramses_amd_mhd_godunov_fine_f90(const ramses_amd_mhd_params *p, int ilevel, int ngrid, const int *igrid, const double *xg,
                                    int64_t ngridmax, int64_t ncoarse, int nx_loc, const double *uold, double *unew, double dx,
                                    double dt) {
  if (!igrid || !xg || !uold || !unew) return failf(RAMSES_AMD_EINVAL, "NULL argument");
  if (nx_loc != 1) return failf(RAMSES_AMD_EUNSUPPORTED, "MHD sweep on the device needs a periodic box with nx=ny=nz=1 (got nx_loc=%d)", nx_loc);
  if (ilevel < 2 || ilevel > 10) return failf(RAMSES_AMD_EINVAL, "level out of range");
  const int n = 1 << ilevel;
  const long N = (long)n * n * n;
  if ((long)ngrid * 8 != N) return failf(RAMSES_AMD_EUNSUPPORTED, "level %d is not fully refined on this rank (ngrid=%d)", ilevel, ngrid);
  const long ncell = ncoarse + 8 * ngridmax;
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
  };
  static DBuf b_vec, b_old, b_new, b_work, b_ig, b_xg, b_org, b_flag;
  hipStream_t s = nullptr;
  const size_t wb = (size_t)ramses_amd_mhd_workspace_bytes(n, n, n);
  HCHK(b_vec.ensure(sizeof(double) * NF * ncell), "hipMalloc cell vectors");
  HCHK(b_old.ensure(sizeof(double) * NF * N), "hipMalloc brick");
  HCHK(b_new.ensure(sizeof(double) * NF * N), "hipMalloc brick");
  HCHK(b_work.ensure(wb), "hipMalloc workspace");
  HCHK(b_org.ensure(sizeof(long) * ngrid), "hipMalloc octorg");
  HCHK(b_ig.ensure(sizeof(int) * ngrid), "hipMalloc igrid");
  HCHK(b_xg.ensure(sizeof(double) * 3 * ngridmax), "hipMalloc xg");
  HCHK(b_flag.ensure(sizeof(int)), "hipMalloc flag");
  void *d_vec = b_vec.p, *d_old = b_old.p, *d_new = b_new.p, *d_work = b_work.p, *d_ig = b_ig.p, *d_xg = b_xg.p, *d_org = b_org.p, *d_flag = b_flag.p;
  HCHK(hipMemcpyAsync(d_vec, uold, sizeof(double) * NF * ncell, hipMemcpyHostToDevice, s), "H2D uold");
  HCHK(hipMemcpyAsync(d_ig, igrid, sizeof(int) * ngrid, hipMemcpyHostToDevice, s), "H2D igrid");
  HCHK(hipMemcpyAsync(d_xg, xg, sizeof(double) * 3 * ngridmax, hipMemcpyHostToDevice, s), "H2D xg");
  HCHK(hipMemsetAsync(d_flag, 0, sizeof(int), s), "memset");
  const double skip[3] = {0.0, 0.0, 0.0};
  HCHK(launch_oct_origin((const int *)d_ig, (const double *)d_xg, ngridmax, ngrid, n, skip, (long *)d_org, (int *)d_flag, s), "oct origin launch");
  int bad = 0;
  HCHK(hipMemcpyAsync(&bad, d_flag, sizeof(int), hipMemcpyDeviceToHost, s), "D2H flag");
  HCHK(hipStreamSynchronize(s), "sync");
  if (bad) return failf(RAMSES_AMD_EINVAL, "%d octs of level %d do not sit on the level lattice", bad, ilevel);
  PackArgs PA;
  PA.igrid = (const int *)d_ig; PA.octorg = (const long *)d_org;
  PA.ngrid = ngrid; PA.n = n; PA.nvar = NF;
  PA.ncoarse = ncoarse; PA.ngridmax = ngridmax; PA.ncell = ncell; PA.pitch_var = N;
  PA.brick = (double *)d_old; PA.cellvec = (double *)d_vec;
  HCHK(launch_oct_copy(PA, true, s), "gather launch");
  if (int rc = ramses_amd_mhd_godunov_brick(p, n, n, n, (const double *)d_old, (double *)d_new, dx, dt, d_work, (int64_t)wb, s)) return rc;
  // unew of the level's cells: the other cells of the host array keep their values (H2D of unew first)
  HCHK(hipMemcpyAsync(d_vec, unew, sizeof(double) * NF * ncell, hipMemcpyHostToDevice, s), "H2D unew");
  PA.brick = (double *)d_new;
  HCHK(launch_oct_copy(PA, false, s), "scatter launch");
  HCHK(hipMemcpyAsync(unew, d_vec, sizeof(double) * NF * ncell, hipMemcpyDeviceToHost, s), "D2H unew");
  HCHK(hipStreamSynchronize(s), "sync");
  return 0;
}


// ---- the level resident on the device between the routines of amr_step (single rank, one fully refined periodic level,
// levelmin = nlevelmax): courant_fine, godunov_fine (set_unew is implied), set_uold run on two bricks that swap roles; the
// host array uold is stale from the first set_uold until ramses_amd_mhd_resident_sync_host_f90 (backup_hydro).  Mirrors
// ramses_amd_resident_* of the hydro solver (capi_host.hip). ---------------------------------------------------------
namespace {
struct MhdBuf {
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
};
struct MhdResident {
  bool valid = false, host_stale = false, new_ready = false;
  int level = 0, ngrid = 0, n = 0;
  long ncell = 0, ncoarse = 0, ngridmax = 0;
  const double *h_uold = nullptr;
  MhdBuf vec, cur, nxt, work, ig, xg, org, flag, red;
};
MhdResident g_mres;

int mres_ensure(int ilevel, int ngrid, const int *igrid, const double *xg, int64_t ngridmax, int64_t ncoarse, int nx_loc,
                const double *uold) {
  MhdResident &R = g_mres;
  if (!igrid || !xg || !uold) return failf(RAMSES_AMD_EINVAL, "NULL argument");
  if (nx_loc != 1) return failf(RAMSES_AMD_EUNSUPPORTED, "MHD residency needs a periodic box with nx=ny=nz=1 (got nx_loc=%d)", nx_loc);
  if (ilevel < 2 || ilevel > 10) return failf(RAMSES_AMD_EINVAL, "level out of range");
  const int n = 1 << ilevel;
  const long N = (long)n * n * n;
  if ((long)ngrid * 8 != N) return failf(RAMSES_AMD_EUNSUPPORTED, "level %d is not fully refined on this rank (ngrid=%d)", ilevel, ngrid);
  const long ncell = ncoarse + 8 * ngridmax;
  if (R.valid && R.level == ilevel && R.ngrid == ngrid && R.h_uold == uold && R.ncell == ncell) return 0;
  if (R.valid && R.host_stale)
    return failf(RAMSES_AMD_EINVAL, "MHD residency: level %d is resident and the host array is stale; ramses_amd_mhd_resident_sync_host_f90 first", R.level);
  R.valid = false;
  hipStream_t s = nullptr;
  HCHK(R.vec.ensure(sizeof(double) * NF * ncell), "hipMalloc cell vectors");
  HCHK(R.cur.ensure(sizeof(double) * NF * N), "hipMalloc brick");
  HCHK(R.nxt.ensure(sizeof(double) * NF * N), "hipMalloc brick");
  HCHK(R.work.ensure((size_t)ramses_amd_mhd_workspace_bytes(n, n, n)), "hipMalloc workspace");
  HCHK(R.org.ensure(sizeof(long) * ngrid), "hipMalloc octorg");
  HCHK(R.ig.ensure(sizeof(int) * ngrid), "hipMalloc igrid");
  HCHK(R.xg.ensure(sizeof(double) * 3 * ngridmax), "hipMalloc xg");
  HCHK(R.flag.ensure(sizeof(int)), "hipMalloc flag");
  HCHK(R.red.ensure(sizeof(double) * (COUR_BLOCKS * 5 + 8)), "hipMalloc reduction");
  HCHK(hipMemcpyAsync(R.vec.p, uold, sizeof(double) * NF * ncell, hipMemcpyHostToDevice, s), "H2D uold");
  HCHK(hipMemcpyAsync(R.ig.p, igrid, sizeof(int) * ngrid, hipMemcpyHostToDevice, s), "H2D igrid");
  HCHK(hipMemcpyAsync(R.xg.p, xg, sizeof(double) * 3 * ngridmax, hipMemcpyHostToDevice, s), "H2D xg");
  HCHK(hipMemsetAsync(R.flag.p, 0, sizeof(int), s), "memset");
  const double skip[3] = {0.0, 0.0, 0.0};
  HCHK(launch_oct_origin((const int *)R.ig.p, (const double *)R.xg.p, ngridmax, ngrid, n, skip, (long *)R.org.p, (int *)R.flag.p, s), "oct origin launch");
  int bad = 0;
  HCHK(hipMemcpyAsync(&bad, R.flag.p, sizeof(int), hipMemcpyDeviceToHost, s), "D2H flag");
  HCHK(hipStreamSynchronize(s), "sync");
  if (bad) return failf(RAMSES_AMD_EINVAL, "%d octs of level %d do not sit on the level lattice", bad, ilevel);
  PackArgs PA;
  PA.igrid = (const int *)R.ig.p; PA.octorg = (const long *)R.org.p;
  PA.ngrid = ngrid; PA.n = n; PA.nvar = NF;
  PA.ncoarse = ncoarse; PA.ngridmax = ngridmax; PA.ncell = ncell; PA.pitch_var = N;
  PA.brick = (double *)R.cur.p; PA.cellvec = (double *)R.vec.p;
  HCHK(launch_oct_copy(PA, true, s), "gather launch");
  R.valid = true; R.host_stale = false; R.new_ready = false;
  R.level = ilevel; R.ngrid = ngrid; R.n = n; R.ncell = ncell; R.ncoarse = ncoarse; R.ngridmax = ngridmax; R.h_uold = uold;
  return 0;
}
}  // namespace

int ramses_amd_mhd_resident_active(void) { return g_mres.valid ? 1 : 0; }

// courant_fine(ilevel) (mhd/courant_fine.f90:1-160): out5 = {min(dt_in, the level's CFL step), mass, total energy, internal
// energy, magnetic energy} of the level (the four sums x dx^3 as the reference accumulates them)
int ramses_amd_mhd_resident_courant_f90(const ramses_amd_mhd_params *p, int ilevel, int ngrid, const int *igrid, const double *xg,
                                        int64_t ngridmax, int64_t ncoarse, int nx_loc, const double *uold, double dx, double dt_in,
                                        double courant_factor, double *out5) {
  if (!out5) return failf(RAMSES_AMD_EINVAL, "NULL argument");
  MhdArgs A;
  if (int rc = make_const(p, A.P)) return rc;
  if (int rc = mres_ensure(ilevel, ngrid, igrid, xg, ngridmax, ncoarse, nx_loc, uold)) return rc;
  MhdResident &R = g_mres;
  if (R.new_ready) return failf(RAMSES_AMD_EINVAL, "courant_fine between godunov_fine and set_uold");
  if (!(dx > 0.0) || !(courant_factor > 0.0)) return failf(RAMSES_AMD_EINVAL, "dx and courant_factor must be > 0");
  const long N = (long)R.n * R.n * R.n;
  A.uold = (const double *)R.cur.p; A.unew = nullptr;
  A.nx = A.ny = A.nz = R.n; A.ncell = N; A.dx = dx; A.dt = 0.0;
  double *partial = (double *)R.red.p, *d_out = partial + COUR_BLOCKS * 5;
  const int nb = (int)((N + 255) / 256 < COUR_BLOCKS ? (N + 255) / 256 : COUR_BLOCKS);
  hipLaunchKernelGGL(mhd_courant_kernel, dim3(nb), dim3(256), 0, nullptr, A, dx * dx * dx, courant_factor, partial);
  hipLaunchKernelGGL(mhd_courant_final_kernel, dim3(1), dim3(256), 0, nullptr, partial, nb, d_out);
  HCHK(hipGetLastError(), "courant launch");
  double h[5];
  HCHK(hipMemcpy(h, d_out, sizeof(h), hipMemcpyDeviceToHost), "D2H courant");
  const double dt0 = courant_factor * dx / p->smallc;            // cmpdt's starting value (:108)
  double dt = h[0] < dt0 ? h[0] : dt0;
  out5[0] = dt < dt_in ? dt : dt_in;
  for (int q = 1; q < 5; q++) out5[q] = h[q];
  return 0;
}

// set_unew + godunov_fine(ilevel) on the resident level: the other brick = uold advanced by dt
int ramses_amd_mhd_resident_godunov_f90(const ramses_amd_mhd_params *p, int ilevel, int ngrid, const int *igrid, const double *xg,
                                        int64_t ngridmax, int64_t ncoarse, int nx_loc, const double *uold, double dx, double dt) {
  if (int rc = mres_ensure(ilevel, ngrid, igrid, xg, ngridmax, ncoarse, nx_loc, uold)) return rc;
  MhdResident &R = g_mres;
  if (int rc = ramses_amd_mhd_godunov_brick(p, R.n, R.n, R.n, (const double *)R.cur.p, (double *)R.nxt.p, dx, dt, R.work.p,
                                            (int64_t)R.work.cap, nullptr)) return rc;
  R.new_ready = true;
  return 0;
}

// set_uold(ilevel) (mhd/godunov_fine.f90:185-281 without gravity / pressure_fix / passive scalars): the bricks swap roles
int ramses_amd_mhd_resident_set_uold_f90(int ilevel) {
  MhdResident &R = g_mres;
  if (!R.valid || R.level != ilevel) return failf(RAMSES_AMD_EINVAL, "set_uold: level %d is not resident", ilevel);
  if (!R.new_ready) return failf(RAMSES_AMD_EINVAL, "set_uold without a godunov_fine before it");
  std::swap(R.cur, R.nxt);
  R.new_ready = false;
  R.host_stale = true;
  return 0;
}

// backup_hydro and anything else on the host that reads uold: the level's cells come back (the other cells of the array
// keep the values they were loaded with -- nothing on the host changes them while the level is resident)
int ramses_amd_mhd_resident_sync_host_f90(double *uold) {
  MhdResident &R = g_mres;
  if (!R.valid || !R.host_stale) return 0;
  if (uold != R.h_uold) return failf(RAMSES_AMD_EINVAL, "sync: not the array the level was loaded from");
  if (R.new_ready) return failf(RAMSES_AMD_EINVAL, "sync between godunov_fine and set_uold");
  const long N = (long)R.n * R.n * R.n;
  PackArgs PA;
  PA.igrid = (const int *)R.ig.p; PA.octorg = (const long *)R.org.p;
  PA.ngrid = R.ngrid; PA.n = R.n; PA.nvar = NF;
  PA.ncoarse = R.ncoarse; PA.ngridmax = R.ngridmax; PA.ncell = R.ncell; PA.pitch_var = N;
  PA.brick = (double *)R.cur.p; PA.cellvec = (double *)R.vec.p;
  HCHK(launch_oct_copy(PA, false, nullptr), "scatter launch");
  HCHK(hipMemcpy(uold, R.vec.p, sizeof(double) * NF * R.ncell, hipMemcpyDeviceToHost), "D2H uold");
  R.host_stale = false;
  return 0;
}

int ramses_amd_mhd_resident_invalidate(void) {
  MhdResident &R = g_mres;
  if (R.valid && R.host_stale) return failf(RAMSES_AMD_EINVAL, "invalidate: the host array is stale; sync first");
  R.valid = false;
  return 0;
}

#endif  // !RAMSES_AMD_MHD_FAST_TU
}  // extern "C"

#include "warm.hpp"
#ifdef RAMSES_AMD_MHD_FAST_TU
RAMSES_AMD_TU_WARM(mhd_sweep_fast)
#else
RAMSES_AMD_TU_WARM(mhd_sweep)
#endif
