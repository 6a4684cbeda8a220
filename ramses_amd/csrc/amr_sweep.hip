// amr_sweep.hip -- godunov_fine on an AMR level (partially refined, or with
// refined cells), directly on the reference's tree arrays:
//   godfine1            hydro/godunov_fine.f90:486-911
//   get3cubefather      amr/nbors_utils.f90:5-194   (27 neighbouring father cells)
//   getnborfather       amr/nbors_utils.f90:404-525 (stencil of the interpolation)
//   interpol_hydro      hydro/interpol_hydro.f90:268-444
//   unsplit             hydro/umuscl.f90:22-171
//
// One wavefront = one oct (the reference's vector element).  The 6^3 stencil of
// the oct lives in LDS: cells of existing neighbour octs are copied from uold,
// cells of missing octs are interpolated from the father level; refined cells
// carry the `ok` flag.  The 64 lanes then ARE the 4^3 cells the reference
// computes slopes and traced states for; 36 lanes compute the 36 interface
// fluxes of the oct; 8 lanes update the oct's cells in the reference's order
// (unew += fL-fR for x, then y, then z).  Fluxes through faces whose
// neighbouring oct does not exist are kept per (oct, face) and added to the
// coarse neighbour cell by a second kernel that replays, per coarse cell, the
// reference's accumulation order: (batch of nvector octs, idim, left/right,
// the 4 fine faces) -- bit-identical for a given NVECTOR.
//
// Strict arithmetic only (-ffp-contract=off).  HBM access is gather-shaped
// (cells of an oct are ngridmax doubles apart, hydro/godunov_fine.f90:600-601);
// the dense brick sweep (hydro_sweep.hip) remains the path of fully refined,
// unrefined levels.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "amr_core.hpp"
#include "amr_sweep_args.hpp"
#include "amr_tree.hpp"
#include "hydro_core.hpp"

namespace ramses_amd {
namespace amrsweep {

constexpr int OCTS_PER_BLOCK = 4;
constexpr int AMR_SMALL_LEVEL_OCTS = 4096;     // below: the single-oct kernel straight away (launch_amr_godunov)

template <int NV>
struct OctFaces {
  double qm[3][3][2][2][NV];  // traced state on the +d face of trace cell a (a = 0..2), transverse 1..2
  double qp[3][3][2][2][NV];  // traced state on the -d face of trace cell a+1
  double fl[3][3][2][2][NV];  // flux through face a of direction d
  double tp[3][3][2][2][2];   // cmpflxm's tmp: normal velocity and internal-energy flux (pressure_fix)
};
template <int NV>
struct OctLds {
  union {
    double u[216][NV];      // primitive variables of the 6^3 stencil (until the traces are done)
    OctFaces<NV> f;             // then the face states and fluxes reuse the same memory
  };
  double uc[64][NV];        // conserved variables of the inner 4^3 cells (difmag only)
  double divc[27];          // velocity divergence at the 3^3 cell corners (difmag only)
  int fc[27];               // the 3^3 neighbouring father cells (1-based cell index)
  int ex[27];               // their son oct (0: not refined)
  unsigned char ok[216];    // cell is refined
};

__device__ __forceinline__ int sidx(int i3, int j3, int k3) { return i3 + 6 * (j3 + 6 * k3); }

// lanes of one wavefront exchange data through LDS: order the memory operations
// for the compiler (the hardware executes a wave's LDS instructions in order)
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int ST, int RS, bool GRAV, int NV, int SCHEME>
__global__ __launch_bounds__(64 * OCTS_PER_BLOCK) void amr_godunov_kernel(AmrSweepArgs A) {
  __shared__ OctLds<NV> lds[OCTS_PER_BLOCK];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int io = blockIdx.x * OCTS_PER_BLOCK + w;
  if (io >= A.ngrid) return;                   // whole wave: no block-level barrier is used
  OctLds<NV> &L = lds[w];
  const HydroConst &P = A.P;
  const int g = A.igrid[io];
  const long ncell = A.ncell;

  // ---- (A) the 3^3 neighbouring father cells -------------------------------
  if (lane < 27) {
    const int d3[3] = {lane % 3 - 1, (lane / 3) % 3 - 1, lane / 9 - 1};
    int c = A.father[g - 1];
#pragma unroll
    for (int axis = 0; axis < 3; axis++) {
      if (d3[axis] != 0 && c > 0) {
        c = nbor_cell(c, 2 * axis + (d3[axis] > 0 ? 1 : 0), A);
        if (c < 0) { atomicAdd(A.err, 1); c = 0; }   // the 3^3 father cells always exist (refinement rules)
      }
    }
    L.fc[lane] = c;
    L.ex[lane] = c > 0 ? A.son[c - 1] : 0;
  }
  wave_sync();

  // ---- (B)+(C) gather the 6^3 stencil and convert to primitive variables -----------
  // (gravity: f of the cell for existing octs, straight injection of the father
  // cell's f for interpolated cells, hydro/godunov_fine.f90:637-647)
  const double dtxhalf = A.dt * 0.5;
  const bool difmag = A.difmag > 0.0;
  // lanes run over the neighbour octs first, the octant second: cell `ind` of octs that
  // were created together (siblings, Z-order) is contiguous in uold, so neighbouring
  // lanes share cache lines instead of striding by ngridmax
  for (int e = lane; e < 216; e += 64) {
    const int ind = e / 27, t = e - 27 * ind;
    const int og = L.ex[t];
    if (og > 0) {
      const int i3 = 2 * (t % 3) + (ind & 1), j3 = 2 * ((t / 3) % 3) + ((ind >> 1) & 1), k3 = 2 * (t / 9) + (ind >> 2);
      const long cell = A.ncoarse + (long)ind * A.ngridmax + og;   // 1-based
      const int s = sidx(i3, j3, k3);
      double u[NV], q[NV], gz[3] = {0.0, 0.0, 0.0};
#pragma unroll
      for (int v = 0; v < NV; v++) u[v] = A.uold[(long)v * ncell + cell - 1];
      if (GRAV) {
#pragma unroll
        for (int d = 0; d < 3; d++) gz[d] = A.grav[(long)d * ncell + cell - 1];
      }
      ctoprim_cell<NV, GRAV>(u, gz, dtxhalf, P, q);
#pragma unroll
      for (int v = 0; v < NV; v++) L.u[s][v] = q[v];
      L.ok[s] = A.son[cell - 1] > 0;
      if (difmag && i3 >= 1 && i3 <= 4 && j3 >= 1 && j3 <= 4 && k3 >= 1 && k3 <= 4) {
        const int cc = (i3 - 1) + 4 * ((j3 - 1) + 4 * (k3 - 1));
#pragma unroll
        for (int v = 0; v < NV; v++) L.uc[cc][v] = u[v];
      }
    }
  }
  if (lane < 27 && L.ex[lane] == 0 && L.fc[lane] > 0) {
    // missing oct: interpolate the father cell with its 2*ndim neighbours
    const int c0 = L.fc[lane];
    double u1[7][NV], u2[8][NV];
#pragma unroll
    for (int j = 0; j < 7; j++) {
      int c = c0;
      if (j > 0) {
        c = nbor_cell(c0, j - 1, A);
        if (c < 0) c = -c;
      }
#pragma unroll
      for (int v = 0; v < NV; v++) u1[j][v] = A.uold[(long)v * ncell + c - 1];
    }
    interpol_hydro_cell<NV>(u1, u2, A.interpol_var, A.interpol_type, P.smallr);
    double gz[3] = {0.0, 0.0, 0.0};
    if (GRAV) {
#pragma unroll
      for (int d = 0; d < 3; d++) gz[d] = A.grav[(long)d * ncell + c0 - 1];
    }
    const int t = lane;
#pragma unroll
    for (int ind = 0; ind < 8; ind++) {
      const int i3 = 2 * (t % 3) + (ind & 1), j3 = 2 * ((t / 3) % 3) + ((ind >> 1) & 1), k3 = 2 * (t / 9) + (ind >> 2);
      const int s = sidx(i3, j3, k3);
      double q[NV];
      ctoprim_cell<NV, GRAV>(u2[ind], gz, dtxhalf, P, q);
#pragma unroll
      for (int v = 0; v < NV; v++) L.u[s][v] = q[v];
      L.ok[s] = 0;
      if (difmag && i3 >= 1 && i3 <= 4 && j3 >= 1 && j3 <= 4 && k3 >= 1 && k3 <= 4) {
        const int cc = (i3 - 1) + 4 * ((j3 - 1) + 4 * (k3 - 1));
#pragma unroll
        for (int v = 0; v < NV; v++) L.uc[cc][v] = u2[ind][v];
      }
    }
  }
  wave_sync();

  // ---- (C') cmpdivu (hydro/uplmde.f90:702-764): velocity divergence at the 3^3 corners ----
  if (difmag && lane < 27) {
    // corner (i,j,k), i,j,k = 1..3 in the reference's flux indexing = stencil cells i, i+1
    const int ci = lane % 3 + 2, cj = (lane / 3) % 3 + 2, ck = lane / 9 + 2;   // stencil coordinate of cell (i,j,k)
    const double fct = 0.25 / A.dx;
    auto Q = [&](int di, int dj, int dk, int v) { return L.u[sidx(ci + di, cj + dj, ck + dk)][v]; };
    double ux = 0.0, vy = 0.0, wz = 0.0;
    ux = ux + fct * (Q(0, 0, 0, 1) - Q(-1, 0, 0, 1));
    ux = ux + fct * (Q(0, -1, 0, 1) - Q(-1, -1, 0, 1));
    vy = vy + fct * (Q(0, 0, 0, 2) - Q(0, -1, 0, 2) + Q(-1, 0, 0, 2) - Q(-1, -1, 0, 2));
    ux = ux + fct * (Q(0, 0, -1, 1) - Q(-1, 0, -1, 1) + Q(0, -1, -1, 1) - Q(-1, -1, -1, 1));
    vy = vy + fct * (Q(0, 0, -1, 2) - Q(0, -1, -1, 2) + Q(-1, 0, -1, 2) - Q(-1, -1, -1, 2));
    wz = wz + fct * (Q(0, 0, 0, 3) - Q(0, 0, -1, 3) + Q(0, -1, 0, 3) - Q(0, -1, -1, 3) + Q(-1, 0, 0, 3) - Q(-1, 0, -1, 3) +
                     Q(-1, -1, 0, 3) - Q(-1, -1, -1, 3));
    L.divc[lane] = ux + vy + wz;
  }

  // ---- (D) slopes + trace: lane = one of the 4^3 cells ------------------------
  const double dtdx = A.dt / A.dx;
  {
    const int ti = lane & 3, tj = (lane >> 2) & 3, tk = lane >> 4;
    const int s = sidx(ti + 1, tj + 1, tk + 1);
    double qb[NV], dq[3][NV];
#pragma unroll
    for (int v = 0; v < NV; v++) {
      qb[v] = L.u[s][v];
      if constexpr (ST == 3) {
        // positivity-preserving multi-D slope: the 3^3 neighbourhood of the cell
        double nb[27], d3[3];
#pragma unroll
        for (int t = 0; t < 27; t++) nb[t] = L.u[s + (t % 3 - 1) + 6 * ((t / 3) % 3 - 1) + 36 * (t / 9 - 1)][v];
        slope3_var(nb, d3);
        dq[0][v] = d3[0]; dq[1][v] = d3[1]; dq[2][v] = d3[2];
      } else {
        dq[0][v] = slope1<ST>(L.u[s - 1][v], qb[v], L.u[s + 1][v], P);
        dq[1][v] = slope1<ST>(L.u[s - 6][v], qb[v], L.u[s + 6][v], P);
        dq[2][v] = slope1<ST>(L.u[s - 36][v], qb[v], L.u[s + 36][v], P);
      }
    }
    double qm[3][NV], qp[3][NV];
    if constexpr (SCHEME == 0) {
      trace3d_cell<NV>(qb, dq, dtdx, dtdx, dtdx, P, qm, qp);
    } else {
      const double cs = ctoprim_sound(qb[0], qb[4], P);
      tracexyz_cell<NV>(qb, dq, cs, dtdx, dtdx, dtdx, P, qm, qp);
    }
    wave_sync();            // every lane has read its stencil values: the memory is reused below
    const int tc[3] = {ti, tj, tk};
#pragma unroll
    for (int d = 0; d < 3; d++) {
      const int t0 = d == 0 ? 1 : 0, t1 = d == 2 ? 1 : 2;     // transverse axes, increasing
      const int a = tc[d], b = tc[t0] - 1, c = tc[t1] - 1;
      if (b >= 0 && b < 2 && c >= 0 && c < 2) {
        if (a <= 2) {
#pragma unroll
          for (int v = 0; v < NV; v++) L.f.qm[d][a][b][c][v] = qm[d][v];
        }
        if (a >= 1) {
#pragma unroll
          for (int v = 0; v < NV; v++) L.f.qp[d][a - 1][b][c][v] = qp[d][v];
        }
      }
    }
  }
  wave_sync();

  // ---- (E) the 36 interface fluxes, zeroed at refined interfaces --------------
  if (lane < 36) {
    const int d = lane / 12, r = lane % 12, a = r >> 2, b = r & 1, c = (r >> 1) & 1;
    double qL[NV], qR[NV], fx[NV];
#pragma unroll
    for (int v = 0; v < NV; v++) { qL[v] = L.f.qm[d][a][b][c][v]; qR[v] = L.f.qp[d][a][b][c][v]; }
    const bool pow2 = A.pow2 != 0;
    const bool pfix = A.divu != nullptr;
    double tp2[2] = {0.0, 0.0};
    if (pfix) {
      if (d == 0) scaled_interface_flux_tmp<RS, NV, 0>(qL, qR, P, A.dt, A.dx, A.rdx, pow2, fx, tp2);
      else if (d == 1) scaled_interface_flux_tmp<RS, NV, 1>(qL, qR, P, A.dt, A.dx, A.rdx, pow2, fx, tp2);
      else scaled_interface_flux_tmp<RS, NV, 2>(qL, qR, P, A.dt, A.dx, A.rdx, pow2, fx, tp2);
    } else {
      if (d == 0) scaled_interface_flux<RS, NV, 0>(qL, qR, P, A.dt, A.dx, A.rdx, dtdx, pow2, fx);
      else if (d == 1) scaled_interface_flux<RS, NV, 1>(qL, qR, P, A.dt, A.dx, A.rdx, dtdx, pow2, fx);
      else scaled_interface_flux<RS, NV, 2>(qL, qR, P, A.dt, A.dx, A.rdx, dtdx, pow2, fx);
    }
    const int t0 = d == 0 ? 1 : 0, t1 = d == 2 ? 1 : 2;
    if (difmag) {
      // consup (hydro/uplmde.f90:769-866): the face's flux index along d is a+1, the own
      // cells' transverse indices are b+1, c+1; corners (i,j,k) live at divc[(i-1)+3(j-1)+9(k-1)]
      int fi[3];
      fi[d] = a + 1; fi[t0] = b + 1; fi[t1] = c + 1;
      auto DV = [&](int i, int j, int k) { return L.divc[(i - 1) + 3 * (j - 1) + 9 * (k - 1)]; };
      const int i = fi[0], j = fi[1], k = fi[2];
      const double factor = 0.25;
      double div1;
      if (d == 0) {
        div1 = factor * DV(i, j, k);
        div1 = div1 + factor * DV(i, j + 1, k);
        div1 = div1 + factor * (DV(i, j, k + 1) + DV(i, j + 1, k + 1));
      } else if (d == 1) {
        div1 = 0.0;
        div1 = div1 + factor * (DV(i, j, k) + DV(i + 1, j, k));
        div1 = div1 + factor * (DV(i, j, k + 1) + DV(i + 1, j, k + 1));
      } else {
        div1 = factor * (DV(i, j, k) + DV(i + 1, j, k) + DV(i, j + 1, k) + DV(i + 1, j + 1, k));
      }
      div1 = A.difmag * __builtin_fmin(0.0, div1);
      // conserved variables of the two cells of the face (4^3 coordinates = the reference's cell index)
      int cr4[3] = {fi[0], fi[1], fi[2]}, cl4[3] = {fi[0], fi[1], fi[2]};
      cl4[d] -= 1;
      const int ir = cr4[0] + 4 * (cr4[1] + 4 * cr4[2]), il = cl4[0] + 4 * (cl4[1] + 4 * cl4[2]);
#pragma unroll
      for (int v = 0; v < NV; v++) fx[v] = fx[v] + A.dt * div1 * (L.uc[ir][v] - L.uc[il][v]);
    }
    // stencil coordinates of the two cells of the face
    int cl[3];
    cl[d] = a + 1; cl[t0] = b + 2; cl[t1] = c + 2;
    const int sl = sidx(cl[0], cl[1], cl[2]);
    const int stride = d == 0 ? 1 : (d == 1 ? 6 : 36);
    const bool zero = L.ok[sl] || L.ok[sl + stride];
#pragma unroll
    for (int v = 0; v < NV; v++) L.f.fl[d][a][b][c][v] = zero ? 0.0 : fx[v];
    L.f.tp[d][a][b][c][0] = zero ? 0.0 : tp2[0];
    L.f.tp[d][a][b][c][1] = zero ? 0.0 : tp2[1];
  }
  wave_sync();

  // ---- (F) conservative update of the oct's 8 cells ----------------------------
  if (lane < 8) {
    const int ic[3] = {lane & 1, (lane >> 1) & 1, lane >> 2};
    const long cell = A.ncoarse + (long)lane * A.ngridmax + g;
#pragma unroll
    for (int v = 0; v < NV; v++) {
      double un = A.unew[(long)v * ncell + cell - 1];
#pragma unroll
      for (int d = 0; d < 3; d++) {
        const int t0 = d == 0 ? 1 : 0, t1 = d == 2 ? 1 : 2;
        const int a = ic[d], b = ic[t0], c = ic[t1];
        un = un + (L.f.fl[d][a][b][c][v] - L.f.fl[d][a + 1][b][c][v]);
      }
      A.unew[(long)v * ncell + cell - 1] = un;
    }
    if (A.divu) {
      // pressure_fix: velocity divergence and internal energy (hydro/godunov_fine.f90:771-786)
      double dv = A.divu[cell - 1], en = A.enew[cell - 1];
#pragma unroll
      for (int d = 0; d < 3; d++) {
        const int t0 = d == 0 ? 1 : 0, t1 = d == 2 ? 1 : 2;
        const int a = ic[d], b = ic[t0], c = ic[t1];
        dv = dv + (L.f.tp[d][a][b][c][0] - L.f.tp[d][a + 1][b][c][0]);
        en = en + (L.f.tp[d][a][b][c][1] - L.f.tp[d][a + 1][b][c][1]);
      }
      A.divu[cell - 1] = dv;
      A.enew[cell - 1] = en;
    }
  }
  // ---- (G) fluxes owed to coarse neighbour cells --------------------------------
  if (lane >= 8 && lane < 14) {
    const int f = lane - 8, d = f >> 1, side = f & 1;
    const int nb = A.nbor[(long)f * A.ngridmax + g - 1];
    const bool coarse = A.son[nb - 1] == 0;
    A.corr_tgt[(long)io * 6 + f] = coarse ? nb : 0;
    if (coarse) {
      const int a = side ? 2 : 0;
      constexpr int CV = NV + 2;                       // fluxes + the two pressure_fix quantities
      double *dst = A.corr + ((long)io * 6 + f) * 4 * CV;
#pragma unroll
      for (int q = 0; q < 4; q++) {
#pragma unroll
        for (int v = 0; v < NV; v++) dst[q * CV + v] = L.f.fl[d][a][q & 1][q >> 1][v];
        dst[q * CV + NV] = L.f.tp[d][a][q & 1][q >> 1][0];
        dst[q * CV + NV + 1] = L.f.tp[d][a][q & 1][q >> 1][1];
      }
    }
  }
}


// ===========================================================================
// Grouped variant: one WORKGROUP per father oct.  The (up to) eight son octs of one level-(l-1) oct
// are updated together from ONE 8^3 stencil (the 4^3 father cells around the father oct): 512
// gathered / interpolated cells, 6^3 traces and 240 interface fluxes for 64 cells, where eight
// single-oct waves gather 8 x 216 cells, trace 8 x 64 and solve 8 x 36 interfaces.  Every cell,
// slope, traced state and flux is the same function of the same stencil values as in the
// single-oct kernel (and in the reference), so the results are bit-identical; octs of the group
// that are not in the call's list (posof < 0) are left alone.  pressure_fix and difmag ride along as template flags
// (PFIX, DIFMAG); only scheme='plmde' with pressure_fix takes the single-oct kernel.
// ===========================================================================
constexpr int GRP_THREADS = 256;
// workgroups per CU the register allocation must allow (256 threads: waves per SIMD).  The kernel is latency-bound: measured
// on the 256^3 tree (profiles/r02_amr_probe_walk.txt) 3 groups (148 VGPRs, no spills) 4.46 ms, 4 (128 VGPRs, 13 spilled)
// 3.44 ms, 5 (96 VGPRs, 144 spilled to scratch) 3.08 ms, 6 3.37 ms, 7 3.10 ms per sweep
// (tried and dropped: unew of the updated cells fetched by the idle fourth wavefront behind the gather and parked in LDS --
// 3.02 -> 3.30 ms, the extra live registers spill)
constexpr int GRP_MINWAVES = 5;

// z stride of the 8^3 stencil in LDS.  With 64 the 4 x 4 x 4 inner cells a wave traces sit on x + 8j (mod 32 doubles = the
// 64 four-byte banks) whatever their plane: 16 bank pairs for 64 lanes, every stencil read a 4-way conflict
// (profiles/r03_amr_sweep_pmc.txt: 4.0e8 conflict cycles of 7.3e8 LDS-active ones).  68 puts plane k on x + 8j + 4k: each
// bank pair twice, the minimum for 64 eight-byte lanes.
constexpr int GRP_ZS = 68;
constexpr int GRP_STENCIL = 7 * GRP_ZS + 64;
// LDS layout of the stencil and the face arrays: variable-major (consecutive lanes touch consecutive doubles of one
// variable; cell-major, the NV values of a cell together, measured the same: 3.795 vs 3.778 ms)
template <int NV>
struct GrpFaces {
  double qm[3][NV][80];   // traced state on the +d face of the low cell of face (a = 0..4, 4x4 transverse); then the flux
  double qp[3][NV][80];   // traced state on the -d face of the high cell
};
// pressure_fix: cmpflxm's tmp per face (normal velocity and internal-energy flux, hydro/umuscl.f90:714-856)
template <bool PFIX>
struct GrpTmp { double tp[3][2][80]; };
template <>
struct GrpTmp<false> {};
// difmag (cmpdivu + consup, hydro/uplmde.f90:702-866): the conserved variables of the 160 traced cells and the velocity
// divergence at the 5^3 cell corners of the updated block
template <int NV, bool DIFMAG>
struct GrpDif { double uc[NV][160]; double divc[125]; };
template <int NV>
struct GrpDif<NV, false> {};
// slot of cell (i, j, k) of the 6^3 trace block among the 160 traced cells (the thread that traces it), -1: an edge / corner
// cell.  Threads 0..63 own the inner 4^3, threads 64 + 16 f + (pb - 1) + 4 (pc - 1) the shell of face f = 2 axis + high.
__device__ __forceinline__ int trace_slot(int i, int j, int k) {
  const bool ei = i == 0 || i == 5, ej = j == 0 || j == 5, ek = k == 0 || k == 5;
  const int ne = (ei ? 1 : 0) + (ej ? 1 : 0) + (ek ? 1 : 0);
  if (ne == 0) return (i - 1) + 4 * (j - 1) + 16 * (k - 1);
  if (ne > 1) return -1;
  if (ei) return 64 + 16 * (0 + (i == 5 ? 1 : 0)) + (j - 1) + 4 * (k - 1);
  if (ej) return 64 + 16 * (2 + (j == 5 ? 1 : 0)) + (i - 1) + 4 * (k - 1);
  return 64 + 16 * (4 + (k == 5 ? 1 : 0)) + (i - 1) + 4 * (j - 1);
}
#define GU(s, v) u[v][s]
#define GF(arr, d, r, v) f.arr[d][v][r]
template <int NV, bool PFIX, bool DIFMAG>
struct GrpLds : GrpTmp<PFIX>, GrpDif<NV, DIFMAG> {
  union {
    double u[NV][GRP_STENCIL]; // primitive variables of the 8^3 stencil (until the traces are done)
    GrpFaces<NV> f;
  };
  // tab[0..63]    fc: the 4^3 father cells (1-based cell index, 0: not there)
  // tab[64..127]  ex: their son oct (0: not refined)
  // tab[128..191] px: position of the son oct of each father cell in the call's list (-1: none / not in the list)
  int tab[192];
  int io[8];               // position of each son of the father oct in the call's list (-1: not active)
  unsigned char ok[GRP_STENCIL];   // cell is refined
};
__device__ __forceinline__ int gsidx(int i, int j, int k) { return i + 8 * j + GRP_ZS * k; }
__device__ __forceinline__ int gface(int a, int b, int c) { return a * 16 + b + 4 * c; }

// Stencil cells of a father cell c0 that has no son oct (thread t of the 4^3 father cells): interpol_hydro of the father
// cell with its 2*ndim neighbours (hydro/interpol_hydro.f90:268-444, getnborfather's coarser fallback), then ctoprim.
// u = the stencil [NV][GRP_STENCIL] in LDS.
template <int NV, bool GRAV>
__device__ __forceinline__ void grp_fill_missing(const AmrSweepArgs &A, int c0, int t, double *__restrict__ u,
                                                          unsigned char *__restrict__ ok, double *__restrict__ uc) {
  const HydroConst &P = A.P;
  const long ncell = A.ncell;
  const int i0 = 2 * (t & 3), j0 = 2 * ((t >> 2) & 3), k0 = 2 * (t >> 4);
  if (c0 > 0) {
    double u1[7][NV], u2[8][NV];
#pragma unroll
    for (int j = 0; j < 7; j++) {
      int c = c0;
      if (j > 0) {
        c = nbor_cell(c0, j - 1, A);
        if (c < 0) c = -c;
      }
#pragma unroll
      for (int v = 0; v < NV; v++) u1[j][v] = A.uold[(long)v * ncell + c - 1];
    }
    interpol_hydro_cell<NV>(u1, u2, A.interpol_var, A.interpol_type, P.smallr);
    double gz[3] = {0.0, 0.0, 0.0};
    if (GRAV) {
#pragma unroll
      for (int d = 0; d < 3; d++) gz[d] = A.grav[(long)d * ncell + c0 - 1];
    }
#pragma unroll
    for (int ind = 0; ind < 8; ind++) {
      const int s = gsidx(i0 + (ind & 1), j0 + ((ind >> 1) & 1), k0 + (ind >> 2));
      double q[NV];
      ctoprim_cell<NV, GRAV>(u2[ind], gz, A.dt * 0.5, P, q);
#pragma unroll
      for (int v = 0; v < NV; v++) u[v * GRP_STENCIL + s] = q[v];
      ok[s] = 0;
      if (uc) {
        // difmag: the interpolated conserved variables of the traced cells (the reference's uloc)
        const int i6 = i0 + (ind & 1) - 1, j6 = j0 + ((ind >> 1) & 1) - 1, k6 = k0 + (ind >> 2) - 1;
        if (i6 >= 0 && i6 <= 5 && j6 >= 0 && j6 <= 5 && k6 >= 0 && k6 <= 5) {
          const int sl = trace_slot(i6, j6, k6);
          if (sl >= 0) {
#pragma unroll
            for (int v = 0; v < NV; v++) uc[v * 160 + sl] = u2[ind][v];
          }
        }
      }
    }
  } else {
    // no father cell: nothing that is stored depends on these cells; keep them finite
#pragma unroll
    for (int ind = 0; ind < 8; ind++) {
      const int s = gsidx(i0 + (ind & 1), j0 + ((ind >> 1) & 1), k0 + (ind >> 2));
#pragma unroll
      for (int v = 0; v < NV; v++) u[v * GRP_STENCIL + s] = 1.0;
      ok[s] = 0;
    }
  }
}


// The walk of every group in a pass of its own (one thread per father cell, nothing but dependent loads: the
// latency hides behind thousands of waves here, not behind the four groups a CU holds in the sweep kernel):
// walk[g*192 + t] = father cell, [+64] its son oct, [+128] the son's position in the call's list
__device__ __forceinline__ void group_walk_block(const AmrSweepArgs &A, int block, const int *__restrict__ groups, int ngroups,
                                                 const int *__restrict__ posof, int *__restrict__ walk) {
  const long e = (long)block * 256 + threadIdx.x;
  if (e >= (long)ngroups * 64) return;
  const int g = (int)(e >> 6), t = (int)(e & 63);
  const int c = group_father_cell(A, groups[g], t);
  const int og = c > 0 ? A.son[c - 1] : 0;
  int *w = walk + (long)g * 192;
  w[t] = c;
  w[64 + t] = og;
  w[128 + t] = og > 0 ? posof[og - 1] : -1;
}
#ifndef AMR_SWEEP_ST
__global__ __launch_bounds__(256) void amr_group_walk_kernel(AmrSweepArgs A, const int *__restrict__ groups, int ngroups,
                                                             const int *__restrict__ posof, int *__restrict__ walk) {
  group_walk_block(A, blockIdx.x, groups, ngroups, posof, walk);
}
#endif

template <int ST, int RS, bool GRAV, int NV, int SCHEME, bool PFIX, bool DIFMAG>
__global__ __launch_bounds__(GRP_THREADS, GRP_MINWAVES) void amr_group_kernel(AmrSweepArgs A, const int *__restrict__ groups,
                                                                 const int *__restrict__ posof, const int *__restrict__ walk) {
  __shared__ GrpLds<NV, PFIX, DIFMAG> L;
  const int t = threadIdx.x;
  const HydroConst &P = A.P;
  // Workgroup b runs on XCD b % 8 (each with its own L2).  Neighbouring father octs read the same ghost octs: give every
  // XCD a contiguous run of the list (on a tree numbered along a Z-order curve: one octant of the level), so that those
  // re-reads hit its L2 (profiles/r03_amr_sweep_pmc.txt: 9.0 GB fetched per sweep with the round-robin order = every
  // record from HBM by every group that reads it).
  int bid = blockIdx.x;
  {
    const int nblk = gridDim.x, per = nblk >> 3, rem = nblk & 7, x = bid & 7;
    bid = x * per + (x < rem ? x : rem) + (bid >> 3);
  }
  const int gF = groups[bid];                 // the father oct (level l-1)
  const long ncell = A.ncell;

  // ---- (A) the 4^3 father cells around the father oct --------------------------------
  if (walk) {
    // the pre-pass has walked: one coalesced 768-byte read (fc, ex, px are contiguous)
    if (t < 192) L.tab[t] = walk[(long)bid * 192 + t];
  } else if (t < 64) {
    const int c = group_father_cell(A, gF, t);
    L.tab[t] = c;
    const int og_t = c > 0 ? A.son[c - 1] : 0;
    L.tab[64 + t] = og_t;
    L.tab[128 + t] = og_t > 0 ? posof[og_t - 1] : -1;
  }
  __syncthreads();
  if (t < 8) L.io[t] = L.tab[128 + (1 + (t & 1)) + 4 * ((1 + ((t >> 1) & 1)) + 4 * (1 + (t >> 2)))];

  // ---- (B)+(C) gather the 8^3 stencil, convert to primitive variables ------------------
  const double dtxhalf = A.dt * 0.5;
  const bool packed = A.packed != nullptr;
  for (int e = t; e < 512; e += GRP_THREADS) {
    // cell vectors: lanes run over the father cells first (the same octant of sibling octs is contiguous);
    // packed records: over the octants first (the 8 values of a variable are 64 contiguous bytes, sibling octs follow)
    const int ind = packed ? (e & 7) : (e >> 6), f = packed ? (e >> 3) : (e & 63);
    const int og = L.tab[64 + f];
    if (og > 0) {
      const int i3 = 2 * (f & 3) + (ind & 1), j3 = 2 * ((f >> 2) & 3) + ((ind >> 1) & 1), k3 = 2 * (f >> 4) + (ind >> 2);
      const int s = gsidx(i3, j3, k3);
      double u[NV], q[NV], gz[3] = {0.0, 0.0, 0.0};
      bool refined;
      const int px = L.tab[128 + f];
      if (packed && px >= 0) {
        // the record holds the cell's primitive variables (ctoprim ran once per cell in amr_pack_kernel)
        const double *__restrict__ r = A.packed + (long)px * A.rec;
#pragma unroll
        for (int v = 0; v < NV; v++) L.GU(s, v) = r[v * 8 + ind];
        L.ok[s] = reinterpret_cast<const int *>(r + 8 * NV)[ind] != 0;
        continue;
      } else {
        const long cell = A.ncoarse + (long)ind * A.ngridmax + og;   // 1-based
#pragma unroll
        for (int v = 0; v < NV; v++) u[v] = A.uold[(long)v * ncell + cell - 1];
        if (GRAV) {
#pragma unroll
          for (int d = 0; d < 3; d++) gz[d] = A.grav[(long)d * ncell + cell - 1];
        }
        refined = A.son[cell - 1] > 0;
      }
      ctoprim_cell<NV, GRAV>(u, gz, dtxhalf, P, q);
#pragma unroll
      for (int v = 0; v < NV; v++) L.GU(s, v) = q[v];
      L.ok[s] = refined;
    }
  }
  // father cells without a son oct (a level's edge): interpolated from the father level.  (The 75 doubles of
  // interpol_hydro's stencil cost the kernel 136 spilled registers; without the branch it runs 5 % faster, 2.33 vs 2.46 ms
  // on the 256^3 tree -- and as a call to a function of its own 3.4x SLOWER, 7.98 ms: the callee's private arrays become a
  // stack frame every wave of the kernel has to reserve.)
  {
    double *ucp = nullptr;
    if constexpr (DIFMAG) ucp = &L.uc[0][0];
    if (t < 64 && L.tab[64 + t] == 0) grp_fill_missing<NV, GRAV>(A, L.tab[t], t, &L.u[0][0], L.ok, ucp);
  }
  __syncthreads();

  // a father cell that an active son needs and that does not exist: the tree breaks the refinement rules
  if (t < 64 && L.tab[t] == 0) {
    const int i = t & 3, j = (t >> 2) & 3, k = t >> 4;
    for (int so = 0; so < 8; so++) {
      const int dxs = i - (1 + (so & 1)), dys = j - (1 + ((so >> 1) & 1)), dzs = k - (1 + (so >> 2));
      if (L.io[so] >= 0 && dxs >= -1 && dxs <= 1 && dys >= -1 && dys <= 1 && dzs >= -1 && dzs <= 1) atomicAdd(A.err, 1);
    }
  }

  // ---- (D) slopes + trace of the inner 6^3 cells -----------------------------------------
  const double dtdx = A.dt / A.dx;
  double qm[3][NV], qp[3][NV];
  // Of the 6^3 cells around the updated 4^3 only 160 feed an interface: the inner 4^3 (all three directions) and the 6 x 16
  // cells of the face shells (their one face towards the block); the 56 edge and corner cells feed none.  Threads 0..63
  // trace the inner cells, 64..159 the shells.
  const bool tracer = t < 160;
  int ti, tj, tk;
  if (t < 64) {
    ti = 1 + (t & 3); tj = 1 + ((t >> 2) & 3); tk = 1 + (t >> 4);
  } else {
    const int e = t - 64, f = e >> 4, pb = 1 + (e & 3), pc = 1 + ((e >> 2) & 3), pa = (f & 1) ? 5 : 0;
    ti = (f >> 1) == 0 ? pa : pb;
    tj = (f >> 1) == 1 ? pa : ((f >> 1) == 0 ? pb : pc);
    tk = (f >> 1) == 2 ? pa : pc;
  }
  if constexpr (DIFMAG) {
    // conserved variables of the traced cells of existing octs (missing octs: filled by grp_fill_missing)
    if (tracer) {
      const int i3 = ti + 1, j3 = tj + 1, k3 = tk + 1;
      const int f = (i3 >> 1) + 4 * ((j3 >> 1) + 4 * (k3 >> 1)), ind = (i3 & 1) + 2 * (j3 & 1) + 4 * (k3 & 1);
      const int og = L.tab[64 + f];
      if (og > 0) {
        const long cell = A.ncoarse + (long)ind * A.ngridmax + og;
#pragma unroll
        for (int v = 0; v < NV; v++) L.uc[v][t] = A.uold[(long)v * ncell + cell - 1];
      }
    }
    // cmpdivu (hydro/uplmde.f90:702-764): velocity divergence at the 5^3 corners (i,j,k = 1..5 in the reference's flux
    // indexing; corner i sits on the low side of stencil cell i + 1), by the threads that do not trace
    const int tcn = t - (GRP_THREADS - 125);
    if (tcn >= 0) {
      const int ci = tcn % 5 + 2, cj = (tcn / 5) % 5 + 2, ck = tcn / 25 + 2;
      const double fct = 0.25 / A.dx;
      auto Q = [&](int di, int dj, int dk, int v) { return L.GU(gsidx(ci + di, cj + dj, ck + dk), v); };
      double ux = 0.0, vy = 0.0, wz = 0.0;
      ux = ux + fct * (Q(0, 0, 0, 1) - Q(-1, 0, 0, 1));
      ux = ux + fct * (Q(0, -1, 0, 1) - Q(-1, -1, 0, 1));
      vy = vy + fct * (Q(0, 0, 0, 2) - Q(0, -1, 0, 2) + Q(-1, 0, 0, 2) - Q(-1, -1, 0, 2));
      ux = ux + fct * (Q(0, 0, -1, 1) - Q(-1, 0, -1, 1) + Q(0, -1, -1, 1) - Q(-1, -1, -1, 1));
      vy = vy + fct * (Q(0, 0, -1, 2) - Q(0, -1, -1, 2) + Q(-1, 0, -1, 2) - Q(-1, -1, -1, 2));
      wz = wz + fct * (Q(0, 0, 0, 3) - Q(0, 0, -1, 3) + Q(0, -1, 0, 3) - Q(0, -1, -1, 3) + Q(-1, 0, 0, 3) - Q(-1, 0, -1, 3) +
                       Q(-1, -1, 0, 3) - Q(-1, -1, -1, 3));
      L.divc[tcn] = ux + vy + wz;
    }
  }
  if (tracer) {
    const int s = gsidx(ti + 1, tj + 1, tk + 1);
    double qb[NV], dq[3][NV];
#pragma unroll
    for (int v = 0; v < NV; v++) {
      qb[v] = L.GU(s, v);
      if constexpr (ST == 3) {
        double nb[27], d3[3];
#pragma unroll
        for (int n = 0; n < 27; n++) nb[n] = L.GU(s + (n % 3 - 1) + 8 * ((n / 3) % 3 - 1) + GRP_ZS * (n / 9 - 1), v);
        slope3_var(nb, d3);
        dq[0][v] = d3[0]; dq[1][v] = d3[1]; dq[2][v] = d3[2];
      } else {
        dq[0][v] = slope1<ST>(L.GU(s - 1, v), qb[v], L.GU(s + 1, v), P);
        dq[1][v] = slope1<ST>(L.GU(s - 8, v), qb[v], L.GU(s + 8, v), P);
        dq[2][v] = slope1<ST>(L.GU(s - GRP_ZS, v), qb[v], L.GU(s + GRP_ZS, v), P);
      }
    }
    if constexpr (SCHEME == 0) {
      trace3d_cell<NV>(qb, dq, dtdx, dtdx, dtdx, P, qm, qp);
    } else {
      const double cs = ctoprim_sound(qb[0], qb[4], P);
      tracexyz_cell<NV>(qb, dq, cs, dtdx, dtdx, dtdx, P, qm, qp);
    }
  }
  __syncthreads();            // every thread has read its stencil values: the memory is reused below
  if (tracer) {
    const int tc[3] = {ti, tj, tk};
#pragma unroll
    for (int d = 0; d < 3; d++) {
      const int t0 = d == 0 ? 1 : 0, t1 = d == 2 ? 1 : 2;     // transverse axes, increasing
      const int a = tc[d], b = tc[t0] - 1, c = tc[t1] - 1;
      if (b >= 0 && b < 4 && c >= 0 && c < 4) {
        if (a <= 4) {
#pragma unroll
          for (int v = 0; v < NV; v++) L.GF(qm, d, gface(a, b, c), v) = qm[d][v];
        }
        if (a >= 1) {
#pragma unroll
          for (int v = 0; v < NV; v++) L.GF(qp, d, gface(a - 1, b, c), v) = qp[d][v];
        }
      }
    }
  }
  __syncthreads();

  // ---- (E) the 240 interface fluxes, zeroed at refined interfaces ---------------------------
  if (t < 240) {
    const int d = t / 80, r = t % 80, a = r >> 4, b = r & 3, c = (r >> 2) & 3;
    double qL[NV], qR[NV], fx[NV];
#pragma unroll
    for (int v = 0; v < NV; v++) { qL[v] = L.GF(qm, d, r, v); qR[v] = L.GF(qp, d, r, v); }
    const bool pow2 = A.pow2 != 0;
    double tp2[2] = {0.0, 0.0};
    if constexpr (PFIX) {
      if (d == 0) scaled_interface_flux_tmp<RS, NV, 0>(qL, qR, P, A.dt, A.dx, A.rdx, pow2, fx, tp2);
      else if (d == 1) scaled_interface_flux_tmp<RS, NV, 1>(qL, qR, P, A.dt, A.dx, A.rdx, pow2, fx, tp2);
      else scaled_interface_flux_tmp<RS, NV, 2>(qL, qR, P, A.dt, A.dx, A.rdx, pow2, fx, tp2);
    } else {
      if (d == 0) scaled_interface_flux<RS, NV, 0>(qL, qR, P, A.dt, A.dx, A.rdx, dtdx, pow2, fx);
      else if (d == 1) scaled_interface_flux<RS, NV, 1>(qL, qR, P, A.dt, A.dx, A.rdx, dtdx, pow2, fx);
      else scaled_interface_flux<RS, NV, 2>(qL, qR, P, A.dt, A.dx, A.rdx, dtdx, pow2, fx);
    }
    const int t0 = d == 0 ? 1 : 0, t1 = d == 2 ? 1 : 2;
    if constexpr (DIFMAG) {
      // consup (hydro/uplmde.f90:769-866): the face's flux index along d is a + 1, the own cells' transverse indices are
      // b + 1, c + 1; corner (i,j,k) lives at divc[(i-1) + 5 (j-1) + 25 (k-1)]
      int fi[3];
      fi[d] = a + 1; fi[t0] = b + 1; fi[t1] = c + 1;
      auto DV = [&](int i, int j, int k) { return L.divc[(i - 1) + 5 * (j - 1) + 25 * (k - 1)]; };
      const int i = fi[0], j = fi[1], k = fi[2];
      const double factor = 0.25;
      double div1;
      if (d == 0) {
        div1 = factor * DV(i, j, k);
        div1 = div1 + factor * DV(i, j + 1, k);
        div1 = div1 + factor * (DV(i, j, k + 1) + DV(i, j + 1, k + 1));
      } else if (d == 1) {
        div1 = 0.0;
        div1 = div1 + factor * (DV(i, j, k) + DV(i + 1, j, k));
        div1 = div1 + factor * (DV(i, j, k + 1) + DV(i + 1, j, k + 1));
      } else {
        div1 = factor * (DV(i, j, k) + DV(i + 1, j, k) + DV(i, j + 1, k) + DV(i + 1, j + 1, k));
      }
      div1 = A.difmag * __builtin_fmin(0.0, div1);
      // conserved variables of the two cells of the face (coordinates in the 6^3 trace block = the reference's cell index)
      int cr6[3] = {fi[0], fi[1], fi[2]}, cl6[3] = {fi[0], fi[1], fi[2]};
      cl6[d] -= 1;
      const int ir = trace_slot(cr6[0], cr6[1], cr6[2]), il = trace_slot(cl6[0], cl6[1], cl6[2]);
#pragma unroll
      for (int v = 0; v < NV; v++) fx[v] = fx[v] + A.dt * div1 * (L.uc[v][ir] - L.uc[v][il]);
    }
    int cl[3];
    cl[d] = a + 1; cl[t0] = b + 2; cl[t1] = c + 2;            // stencil coordinates of the low cell of the face
    const int sl = gsidx(cl[0], cl[1], cl[2]);
    const int stride = d == 0 ? 1 : (d == 1 ? 8 : GRP_ZS);
    const bool zero = L.ok[sl] || L.ok[sl + stride];
#pragma unroll
    for (int v = 0; v < NV; v++) L.GF(qm, d, r, v) = zero ? 0.0 : fx[v];
    if constexpr (PFIX) {
      L.tp[d][0][r] = zero ? 0.0 : tp2[0];
      L.tp[d][1][r] = zero ? 0.0 : tp2[1];
    }
  }
  __syncthreads();

  // ---- (F) conservative update of the cells of the active sons -------------------------------
  if (t < 64) {
    const int x = t & 3, y = (t >> 2) & 3, z = t >> 4;
    const int so = (x >> 1) + 2 * (y >> 1) + 4 * (z >> 1);
    if (L.io[so] >= 0) {
      const int og = L.tab[64 + (1 + (x >> 1)) + 4 * ((1 + (y >> 1)) + 4 * (1 + (z >> 1)))];
      const int ind = (x & 1) + 2 * (y & 1) + 4 * (z & 1);
      const long cell = A.ncoarse + (long)ind * A.ngridmax + og;
      const int ic[3] = {x, y, z};
#pragma unroll
      for (int v = 0; v < NV; v++) {
        double un = A.unew[(long)v * ncell + cell - 1];
#pragma unroll
        for (int d = 0; d < 3; d++) {
          const int t0 = d == 0 ? 1 : 0, t1 = d == 2 ? 1 : 2;
          const int a = ic[d], b = ic[t0], c = ic[t1];
          un = un + (L.GF(qm, d, gface(a, b, c), v) - L.GF(qm, d, gface(a + 1, b, c), v));
        }
        A.unew[(long)v * ncell + cell - 1] = un;
      }
      if constexpr (PFIX) {
        // pressure_fix: velocity divergence and internal energy (hydro/godunov_fine.f90:771-786)
        double dv = A.divu[cell - 1], en = A.enew[cell - 1];
#pragma unroll
        for (int d = 0; d < 3; d++) {
          const int t0 = d == 0 ? 1 : 0, t1 = d == 2 ? 1 : 2;
          const int a = ic[d], b = ic[t0], c = ic[t1];
          dv = dv + (L.tp[d][0][gface(a, b, c)] - L.tp[d][0][gface(a + 1, b, c)]);
          en = en + (L.tp[d][1][gface(a, b, c)] - L.tp[d][1][gface(a + 1, b, c)]);
        }
        A.divu[cell - 1] = dv;
        A.enew[cell - 1] = en;
      }
    }
  }
  // ---- (G) fluxes owed to coarse neighbour cells (same records as the single-oct kernel) --------
  if (t >= 64 && t < 64 + 48) {
    const int e = t - 64, so = e / 6, f = e % 6, d = f >> 1, side = f & 1;
    const int io = L.io[so];
    if (io >= 0) {
      const int sc[3] = {so & 1, (so >> 1) & 1, so >> 2};
      const int og = L.tab[64 + (1 + sc[0]) + 4 * ((1 + sc[1]) + 4 * (1 + sc[2]))];
      const int nb = A.nbor[(long)f * A.ngridmax + og - 1];
      const bool coarse = A.son[nb - 1] == 0;
      A.corr_tgt[(long)io * 6 + f] = coarse ? nb : 0;
      if (coarse) {
        const int t0 = d == 0 ? 1 : 0, t1 = d == 2 ? 1 : 2;
        const int a = 2 * sc[d] + (side ? 2 : 0);
        constexpr int CV = NV + 2;
        double *dst = A.corr + ((long)io * 6 + f) * 4 * CV;
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const int fi = gface(a, 2 * sc[t0] + (q & 1), 2 * sc[t1] + (q >> 1));
#pragma unroll
          for (int v = 0; v < NV; v++) dst[q * CV + v] = L.GF(qm, d, fi, v);
          if constexpr (PFIX) {
            dst[q * CV + NV] = L.tp[d][0][fi];
            dst[q * CV + NV + 1] = L.tp[d][1][fi];
          } else {
            dst[q * CV + NV] = 0.0;
            dst[q * CV + NV + 1] = 0.0;
          }
        }
      }
    }
  }
}

#undef GU
#undef GF

// The octs of the call's list repacked as contiguous records, record i = oct igrid[i]:
//   [8 x q(:,1)] ... [8 x q(:,nvar)] [8 ints: cell is refined], padded to 128 bytes
// with q = the PRIMITIVE variables of the cell (ctoprim with the gravity predictor, hydro/umuscl.f90:862-954): a cell is read
// by up to 27 father-oct groups, and converting it here makes that once per cell instead of once per reader (the same
// function of the same values: bit-identical).  The grouped kernel then reads a father cell's son oct as one 3-line burst
// instead of 8 x nvar eight-byte gathers at stride ngridmax (profiles/r02_amr_sweep_pmc.txt: those gathers moved 24x the
// algorithmic traffic).  32 octs per workgroup, one cell per thread: cell-vector reads run over consecutive octs, the
// records leave through LDS in record order.
constexpr int PACK_OCTS = 32;
template <int NV, bool GRAV>
__device__ __forceinline__ void pack_block(const AmrSweepArgs &A, int block, double *__restrict__ out, int rec) {
  __shared__ double tile[PACK_OCTS][AMR_PACK_REC_MAX + 1];
  const int base = block * PACK_OCTS;
  const int n = min(PACK_OCTS, A.ngrid - base);
  constexpr int nval = 8 * NV;                            // doubles of data per record
  {
    const int o = threadIdx.x % PACK_OCTS, ind = threadIdx.x / PACK_OCTS;       // consecutive lanes: consecutive octs
    if (o < n) {
      const int g = A.igrid[base + o];
      const long cell = A.ncoarse + (long)ind * A.ngridmax + g - 1;
      double u[NV], q[NV], gz[3] = {0.0, 0.0, 0.0};
#pragma unroll
      for (int v = 0; v < NV; v++) u[v] = A.uold[(long)v * A.ncell + cell];
      if (GRAV) {
#pragma unroll
        for (int d = 0; d < 3; d++) gz[d] = A.grav[(long)d * A.ncell + cell];
      }
      ctoprim_cell<NV, GRAV>(u, gz, A.dt * 0.5, A.P, q);
#pragma unroll
      for (int v = 0; v < NV; v++) tile[o][v * 8 + ind] = q[v];
      reinterpret_cast<int *>(&tile[o][nval])[ind] = A.son[cell] > 0 ? 1 : 0;
    }
  }
  __syncthreads();
  for (int e = threadIdx.x; e < n * rec; e += 256) {
    const int o = e / rec, k = e % rec;
    out[(long)(base + o) * rec + k] = k < nval + 4 ? tile[o][k] : 0.0;
  }
}
// The records and the father-cell walk in ONE launch, their workgroups alternating: the walk is nothing but dependent
// loads (latency), the pack streams the level (bandwidth); side by side on a CU they overlap (walk_blocks = 0: pack only).
template <int NV, bool GRAV>
__global__ __launch_bounds__(256) void amr_prep_kernel(AmrSweepArgs A, double *__restrict__ out, int rec, int pack_blocks,
                                                       int walk_blocks, const int *__restrict__ groups, int ngroups,
                                                       const int *__restrict__ posof, int *__restrict__ walk) {
  const int b = blockIdx.x, both = 2 * min(pack_blocks, walk_blocks);
  bool is_pack;
  int idx;
  if (b < both) { is_pack = (b & 1) == 0; idx = b >> 1; }
  else { is_pack = pack_blocks > walk_blocks; idx = b - both + (both >> 1); }
  if (is_pack) pack_block<NV, GRAV>(A, idx, out, rec);
  else group_walk_block(A, idx, groups, ngroups, posof, walk);
}

// groups[] = the father octs that have at least one son in the call's list, each once: the son at the lowest
// octant position enters it
#ifndef AMR_SWEEP_ST
__global__ __launch_bounds__(1024) void amr_group_build_kernel(AmrSweepArgs A, const int *posof, int *groups, int *count) {
  const int io = blockIdx.x * blockDim.x + threadIdx.x;
  bool lead = false;
  int gF = 0;
  if (io < A.ngrid) {
    const int g = A.igrid[io];
    const int c = A.father[g - 1];
    int pos;
    cell_split(c, A.ncoarse, A.ngridmax, pos, gF);
    lead = true;
    for (int p = 0; p < pos; p++) {
      const int s = A.son[A.ncoarse + (long)p * A.ngridmax + gF - 1];
      if (s > 0 && posof[s - 1] >= 0) { lead = false; break; }
    }
  }
  // one atomic per 1024-thread workgroup (atomics on one address cost ~11 ns each on the 256^3 tree, whether 262144
  // single ones or 32768 per wavefront); the groups of a workgroup keep the order of the list, which keeps
  // neighbouring father octs together
  __shared__ int wcount[16], wbase[16];
  const unsigned long long m = __ballot(lead);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwave = (blockDim.x + 63) >> 6;
  if (lane == 0) wcount[wave] = __popcll(m);
  __syncthreads();
  if (threadIdx.x == 0) {
    int tot = 0;
    for (int w = 0; w < nwave; w++) { wbase[w] = tot; tot += wcount[w]; }
    const int base = tot > 0 ? atomicAdd(count, tot) : 0;
    for (int w = 0; w < nwave; w++) wbase[w] += base;
  }
  __syncthreads();
  if (lead) groups[wbase[wave] + __popcll(m & ((1ull << lane) - 1ull))] = gF;
}
#endif

// posof[oct-1] = position (0-based) of the oct in the active list
#ifndef AMR_SWEEP_ST
__global__ void amr_posof_kernel(const int *igrid, int ngrid, int *posof) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < ngrid) posof[igrid[i] - 1] = i;
}
#endif

// Conservative update at level ilevel-1 (hydro/godunov_fine.f90:798-908): every
// (oct, face) whose neighbouring father cell is a leaf owes it 4 fluxes.  A coarse
// cell has at most 6 such creditors; the thread of the creditor that comes first in
// the reference's loop order (batch of nvector octs, idim, left before right)
// replays all of them sequentially.
#ifndef AMR_SWEEP_ST
__global__ __launch_bounds__(256) void amr_coarse_update_kernel(AmrSweepArgs A, const int *posof, int nvector) {
  const int NV = A.nvar;
  const long ev = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (ev >= (long)A.ngrid * 6) return;
  const int C = A.corr_tgt[ev];
  if (C <= 0) return;
  const int io = (int)(ev / 6), f = (int)(ev % 6);
  const long mykey = ((long)(io / nvector) * 3 + (f >> 1)) * 2 + (f & 1);
  long key[6];
  long src[6];
  int n = 0;
  bool first = true;
  for (int e = 0; e < 6; e++) {
    // the cell on side e of C; if it is refined, its oct borders C with face e^1
    int pos, gC;
    int Ne;
    if (C > A.ncoarse) {
      Ne = nbor_cell(C, e, A);
    } else {
      Ne = -1;   // level-1 father cells: not handled (the launcher refuses ilevel < 3)
    }
    (void)pos; (void)gC;
    if (Ne <= 0) continue;
    const int g2 = A.son[Ne - 1];
    if (g2 == 0) continue;
    const int p2 = posof[g2 - 1];
    if (p2 < 0) continue;                       // not an active oct of this level (cannot happen on one rank)
    const int f2 = e ^ 1;
    if (A.corr_tgt[(long)p2 * 6 + f2] != C) continue;
    const long k = ((long)(p2 / nvector) * 3 + (f2 >> 1)) * 2 + (f2 & 1);
    key[n] = k; src[n] = (long)p2 * 6 + f2; n++;
    if (k < mykey) first = false;
  }
  if (!first) return;
  // insertion sort of <= 6 creditors
  for (int i = 1; i < n; i++) {
    const long k = key[i], s = src[i];
    int j = i - 1;
    while (j >= 0 && key[j] > k) { key[j + 1] = key[j]; src[j + 1] = src[j]; j--; }
    key[j + 1] = k; src[j + 1] = s;
  }
  const double oneontwotondim = 1.0 / 8.0;
  const int CV = NV + 2;
  const int nacc = A.divu ? NV + 2 : NV;       // unew(1:nvar) [, divu, enew]
  for (int v = 0; v < nacc; v++) {
    double *tgt = v < NV ? A.unew + (long)v * A.ncell : (v == NV ? A.divu : A.enew);
    double val = tgt[C - 1];
    for (int i = 0; i < n; i++) {
      const double *c = A.corr + src[i] * 4 * CV;
      const bool left = ((src[i] % 6) & 1) == 0;
      for (int q = 0; q < 4; q++) {
        const double t = c[q * CV + v] * oneontwotondim;
        val = left ? val - t : val + t;
      }
    }
    tgt[C - 1] = val;
  }
}
#endif

template <int ST, int RS, int NV>
static hipError_t launch3(const AmrSweepArgs &A, const int *groups, int ngroups, const int *posof, const int *walk, hipStream_t s) {
  if (ngroups > 0) {
    const dim3 grid(ngroups), block(GRP_THREADS);
    const bool dif = A.difmag > 0.0;
#define RAMSES_AMD_GRP(SCH, PF, DF)                                                                                           \
  do {                                                                                                                        \
    if (A.grav) hipLaunchKernelGGL((amr_group_kernel<ST, RS, true, NV, SCH, PF, DF>), grid, block, 0, s, A, groups, posof, walk);  \
    else hipLaunchKernelGGL((amr_group_kernel<ST, RS, false, NV, SCH, PF, DF>), grid, block, 0, s, A, groups, posof, walk);        \
    return hipGetLastError();                                                                                                 \
  } while (0)
    if (A.scheme == 1) {
      if constexpr (NV == 5) {
        if (dif) RAMSES_AMD_GRP(1, false, true);
        RAMSES_AMD_GRP(1, false, false);
      } else {
        return hipErrorInvalidValue;
      }
    }
    if (A.divu) {
      // pressure_fix (the launcher sends plmde + pressure_fix to the single-oct kernel)
      if (dif) RAMSES_AMD_GRP(0, true, true);
      RAMSES_AMD_GRP(0, true, false);
    }
    if (dif) RAMSES_AMD_GRP(0, false, true);
    RAMSES_AMD_GRP(0, false, false);
#undef RAMSES_AMD_GRP
  }
  const int blocks = (A.ngrid + OCTS_PER_BLOCK - 1) / OCTS_PER_BLOCK;
  const dim3 grid(blocks), block(64 * OCTS_PER_BLOCK);
  if (A.scheme == 1) {
    // scheme='plmde' (tracexyz): hydro variables only, like the dense sweep
    if constexpr (NV == 5) {
      if (A.grav) hipLaunchKernelGGL((amr_godunov_kernel<ST, RS, true, NV, 1>), grid, block, 0, s, A);
      else hipLaunchKernelGGL((amr_godunov_kernel<ST, RS, false, NV, 1>), grid, block, 0, s, A);
      return hipGetLastError();
    } else {
      return hipErrorInvalidValue;
    }
  }
  if (A.grav) hipLaunchKernelGGL((amr_godunov_kernel<ST, RS, true, NV, 0>), grid, block, 0, s, A);
  else hipLaunchKernelGGL((amr_godunov_kernel<ST, RS, false, NV, 0>), grid, block, 0, s, A);
  return hipGetLastError();
}
template <int ST, int RS>
static hipError_t launch2(const AmrSweepArgs &A, const int *groups, int ngroups, const int *posof, const int *walk, hipStream_t s) {
  switch (A.nvar) {
    case 5: return launch3<ST, RS, 5>(A, groups, ngroups, posof, walk, s);
#ifndef RAMSES_AMD_AMR_DEV
    case 6: return launch3<ST, RS, 6>(A, groups, ngroups, posof, walk, s);
    case 7: return launch3<ST, RS, 7>(A, groups, ngroups, posof, walk, s);
#endif
  }
  return hipErrorInvalidValue;
}

// One translation unit per slope type (ramses_amd/build.py compiles this file once without AMR_SWEEP_ST -- the shared kernels and
// the dispatcher -- and once per slope type with -DAMR_SWEEP_ST=<type>: the 1080 instantiations of the option matrix build side
// by side instead of in one 14-minute compile)
template <int ST>
hipError_t launch1(const AmrSweepArgs &A, int rs, const int *groups, int ngroups, const int *posof, const int *walk, hipStream_t s) {
  switch (rs) {
    case RIEMANN_LLF: return launch2<ST, RIEMANN_LLF>(A, groups, ngroups, posof, walk, s);
#ifndef RAMSES_AMD_AMR_DEV     // development build (kernel tuning with scripts/amr_probe.py): minmod + LLF + NVAR=5 only
    case RIEMANN_HLLC: return launch2<ST, RIEMANN_HLLC>(A, groups, ngroups, posof, walk, s);
    case RIEMANN_HLL: return launch2<ST, RIEMANN_HLL>(A, groups, ngroups, posof, walk, s);
    case RIEMANN_ACOUSTIC: return launch2<ST, RIEMANN_ACOUSTIC>(A, groups, ngroups, posof, walk, s);
    case RIEMANN_EXACT: return launch2<ST, RIEMANN_EXACT>(A, groups, ngroups, posof, walk, s);
#endif
    default: return hipErrorInvalidValue;
  }
}

#ifdef AMR_SWEEP_ST
template hipError_t launch1<AMR_SWEEP_ST>(const AmrSweepArgs &, int, const int *, int, const int *, const int *, hipStream_t);
#else
#define RAMSES_AMD_EXTERN_ST(K) extern template hipError_t launch1<K>(const AmrSweepArgs &, int, const int *, int, const int *, const int *, hipStream_t);
RAMSES_AMD_EXTERN_ST(1)
#ifndef RAMSES_AMD_AMR_DEV
RAMSES_AMD_EXTERN_ST(0) RAMSES_AMD_EXTERN_ST(2) RAMSES_AMD_EXTERN_ST(3) RAMSES_AMD_EXTERN_ST(7) RAMSES_AMD_EXTERN_ST(8)
#endif
#undef RAMSES_AMD_EXTERN_ST
#endif

}  // namespace amrsweep

#ifndef AMR_SWEEP_ST
hipError_t launch_amr_godunov(const AmrSweepArgs &A_in, int slope_type, int riemann, int *posof, int nvector,
                              hipStream_t s, double *pack_area, int *walk_area) {
  using namespace amrsweep;
  AmrSweepArgs A = A_in;
  A.packed = nullptr; A.rec = 0;
  if (A.ngrid <= 0) return hipSuccess;
  hipError_t e;
  e = hipMemsetAsync(posof, 0xff, sizeof(int) * A.ngridmax, s);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(amr_posof_kernel, dim3((A.ngrid + 255) / 256), dim3(256), 0, s, A.igrid, A.ngrid, posof);
  // One workgroup per father oct (its sons share one stencil) unless an option only the single-oct kernel
  // carries is on (scheme='plmde' with pressure_fix).
  int *groups = posof + A.ngridmax, *count = groups + A.ngrid;     // workspace tail (ramses_amd_godunov_fine_amr_workspace)
  int ngroups = 0;
  const int *walk = nullptr;
  // A level of a few thousand octs is launch latency, not work: the one-oct-per-wavefront kernel needs no groups, no father-cell
  // pre-pass and -- what counts -- no answer from the device before it can be launched (the group count is a blocking copy).
  // (It carries neither the artificial diffusion nor, outside plmde, pressure_fix: those keep the grouped kernel.)
  const bool small_level = A.ngrid <= AMR_SMALL_LEVEL_OCTS && A.difmag <= 0.0 && A.divu == nullptr;
  if (!(A.divu != nullptr && A.scheme == 1) && !small_level) {
    e = hipMemsetAsync(count, 0, sizeof(int), s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(amr_group_build_kernel, dim3((A.ngrid + 1023) / 1024), dim3(1024), 0, s, A, posof, groups, count);
    e = hipMemcpyAsync(&ngroups, count, sizeof(int), hipMemcpyDeviceToHost, s);
    if (e != hipSuccess) return e;
    e = hipStreamSynchronize(s);
    if (e != hipSuccess) return e;
    // the grouped kernel reads the level from packed oct records, the father-cell walk of all groups runs in the same
    // pre-pass (a caller without the two scratch areas gets the gather from the cell vectors / the walk inside the kernel)
    const bool do_walk = walk_area && ngroups > 0, do_pack = pack_area && ngroups > 0;
    const int walk_blocks = do_walk ? (int)(((long)ngroups * 64 + 255) / 256) : 0;
    if (do_pack) {
      // records of primitive variables + (alternating workgroups of the same launch) the father-cell walk of every group:
      // 768 bytes per father oct, carved out of the caller's workspace like every other scratch area of the call
      // (two calls on two streams with two workspaces do not share anything)
      const int rec = amr_pack_rec(A.nvar);
      const int pack_blocks = (A.ngrid + PACK_OCTS - 1) / PACK_OCTS;
      const dim3 pgrid(pack_blocks + walk_blocks), pblock(256);
      switch (A.nvar * 2 + (A.grav ? 1 : 0)) {
#define RAMSES_AMD_PREP(NV, G) hipLaunchKernelGGL((amr_prep_kernel<NV, G>), pgrid, pblock, 0, s, A, pack_area, rec, pack_blocks, \
                                                   walk_blocks, groups, ngroups, posof, walk_area)
        case 10: RAMSES_AMD_PREP(5, false); break;
        case 11: RAMSES_AMD_PREP(5, true); break;
#ifndef RAMSES_AMD_AMR_DEV
        case 12: RAMSES_AMD_PREP(6, false); break;
        case 13: RAMSES_AMD_PREP(6, true); break;
        case 14: RAMSES_AMD_PREP(7, false); break;
        case 15: RAMSES_AMD_PREP(7, true); break;
#endif
#undef RAMSES_AMD_PREP
        default: return hipErrorInvalidValue;
      }
      A.packed = pack_area; A.rec = rec;
    } else if (do_walk) {
      hipLaunchKernelGGL(amr_group_walk_kernel, dim3(walk_blocks), dim3(256), 0, s, A, groups, ngroups, posof, walk_area);
    }
    if (do_walk) walk = walk_area;
  }
  switch (slope_type) {
    case 1: e = launch1<1>(A, riemann, groups, ngroups, posof, walk, s); break;
#ifndef RAMSES_AMD_AMR_DEV
    case 0: e = launch1<0>(A, riemann, groups, ngroups, posof, walk, s); break;
    case 2: e = launch1<2>(A, riemann, groups, ngroups, posof, walk, s); break;
    case 3: e = launch1<3>(A, riemann, groups, ngroups, posof, walk, s); break;
    case 7: e = launch1<7>(A, riemann, groups, ngroups, posof, walk, s); break;
    case 8: e = launch1<8>(A, riemann, groups, ngroups, posof, walk, s); break;
#endif
    default: return hipErrorInvalidValue;
  }
  if (e != hipSuccess) return e;
  const long nev = (long)A.ngrid * 6;
  hipLaunchKernelGGL(amr_coarse_update_kernel, dim3((int)((nev + 255) / 256)), dim3(256), 0, s, A, posof, nvector);
  return hipGetLastError();
}

// the replay alone: for a caller whose own sweep filed the records (the dense sweep of a level in tiles, csrc/capi_amr.hip)
hipError_t launch_amr_coarse_update(const AmrSweepArgs &A, const int *posof, int nvector, hipStream_t s) {
  if (A.ngrid <= 0) return hipSuccess;
  const long nev = (long)A.ngrid * 6;
  hipLaunchKernelGGL(amrsweep::amr_coarse_update_kernel, dim3((int)((nev + 255) / 256)), dim3(256), 0, s, A, posof, nvector);
  return hipGetLastError();
}

#endif   // AMR_SWEEP_ST

}  // namespace ramses_amd

#include "warm.hpp"
#ifndef AMR_SWEEP_ST
RAMSES_AMD_TU_WARM(amr_sweep)
#else
#define RAMSES_AMD_WARM_CAT2(a, b) RAMSES_AMD_TU_WARM(a##b)
#define RAMSES_AMD_WARM_CAT(a, b) RAMSES_AMD_WARM_CAT2(a, b)
RAMSES_AMD_WARM_CAT(amr_sweep_st, AMR_SWEEP_ST)
#endif
