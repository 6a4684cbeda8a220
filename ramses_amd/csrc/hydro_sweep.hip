// hydro_sweep.hip -- the Godunov sweep of one fully refined level brick on
// MI355X (gfx950): set_unew + godunov_fine/godfine1 + unsplit fused in one
// kernel (reference: hydro/godunov_fine.f90:5-130,486-911, hydro/umuscl.f90).
//
// Design (see DESIGN.md "Godunov sweep kernel"):
//  * a workgroup owns a (64-4) x (BY-4) column tile in (x,y) and MARCHES along
//    z; one wavefront = one y row of 64 x-columns, so every HBM access of a
//    wave is a contiguous 512 B row segment;
//  * a thread keeps its own column's z-neighbours (3 primitive planes, the
//    previous z traced state and z flux) in registers; only the in-plane
//    neighbours go through LDS: the primitive plane (double buffered), the
//    +x/+y traced states and the x/y interface fluxes;
//  * every cell is converted to primitives once, traced once, and every
//    interface flux is computed once per tile (the reference recomputes a 6^3
//    stencil per 2^3 oct: 27x load and ~8x flop redundancy);
//  * halo rows/lanes exit early by role (wave-uniform for rows), so the 2-cell
//    ghost ring costs ctoprim+trace only;
//  * uold is read once (+ tile halo from L2) and unew written once: 80 B per
//    cell update algorithmic HBM traffic.
//
// Compiled twice: strict (-ffp-contract=off, bit-identical to the reference)
// and fast (-DRAMSES_AMD_FAST, FMA contraction allowed).
#include <hip/hip_runtime.h>

#include "hydro_core.hpp"
#include "sweep_args.hpp"

namespace ramses_amd {

#ifdef RAMSES_AMD_FAST
#define SWEEP_NS fastmode
#else
#define SWEEP_NS strictmode
#endif

namespace SWEEP_NS {

constexpr int BX = 64;   // lanes along x = one wavefront
constexpr int NV = 5;    // rho, u, v, w, P

// LDS plane of NV doubles per column: [n][ty][tx]
template <int BY>
struct Plane {
  double v[NV][BY][BX];
};

template <int ST, int RS, int BY, bool GRAV, bool DXPOW2>
__global__ __launch_bounds__(BX *BY) void godunov_sweep_kernel(SweepArgs A) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  Plane<BY> *qbuf = reinterpret_cast<Plane<BY> *>(smem_raw);  // [2] primitives of plane c / c+1
  Plane<BY> *smx = qbuf + 2;                                   // qm along x (state on +x face)
  Plane<BY> *smy = qbuf + 3;                                   // qm along y
  Plane<BY> *fxb = qbuf + 4;                                   // flux through the -x face
  Plane<BY> *fyb = qbuf + 5;                                   // flux through the -y face

  const int tx = threadIdx.x, ty = threadIdx.y;
  const HydroConst &P = A.P;

  // ---- tile decode (XCD-aware: consecutive tiles of one z-chunk column
  // share an XCD's L2 for their halo re-reads) ------------------------------
  int bid = blockIdx.x;
  {
    const int nblk = gridDim.x;
    const int nxcd = 8;
    if (nblk % nxcd == 0) {
      const int per = nblk / nxcd;
      bid = (bid % nxcd) * per + bid / nxcd;
    }
  }
  const int tix = bid % A.ntx;
  const int tiy = (bid / A.ntx) % A.nty;
  const int tiz = bid / (A.ntx * A.nty);
  const int x0 = tix * (BX - 4);
  const int y0 = tiy * (BY - 4);
  const int z0 = tiz * A.zchunk;
  const int z1 = min(z0 + A.zchunk, A.nz);

  // ---- this thread's column ------------------------------------------------
  const int xu = x0 - 2 + tx;  // unwrapped interior coordinate
  const int yu = y0 - 2 + ty;
  int xi, yi;
  if (A.ng == 0) {
    xi = xu < 0 ? xu + A.nx : (xu >= A.nx ? xu - A.nx : xu);
    xi = xi >= A.nx ? xi % A.nx : xi;
    yi = yu < 0 ? yu + A.ny : (yu >= A.ny ? yu - A.ny : yu);
    yi = yi >= A.ny ? yi % A.ny : yi;
  } else {
    xi = min(max(xu, -A.ng), A.nx + A.ng - 1) + A.ng;
    yi = min(max(yu, -A.ng), A.ny + A.ng - 1) + A.ng;
  }
  const long col = (long)xi + (long)yi * A.pitch_y;
  const double *__restrict__ uold = A.uold;
  double *__restrict__ unew = A.unew;
  const double *__restrict__ grav = A.grav;

  // roles
  const bool r_trace = (ty >= 1) && (ty <= BY - 2);
  const bool r_fy = (ty >= 2) && (ty <= BY - 2);
  const bool r_fxz = (ty >= 2) && (ty <= BY - 3);
  const bool r_upd = r_fxz && (tx >= 2) && (tx <= BX - 3) && (xu < A.nx) && (yu < A.ny);

  const double dtdx = A.dt / A.dx;
  const double dtxhalf = A.dt * 0.5;

  auto plane_off = [&](int p) -> long {
    int pz;
    if (A.ng == 0) { pz = p < 0 ? p + A.nz : (p >= A.nz ? p - A.nz : p); }
    else { pz = p + A.ng; }
    return col + (long)pz * A.pitch_z;
  };
  auto load_u = [&](int p, double (&u)[NV]) {
    const long o = plane_off(p);
#pragma unroll
    for (int n = 0; n < NV; n++) u[n] = uold[o + (long)n * A.pitch_var];
  };
  auto load_g = [&](int p, double (&g)[3]) {
    if (GRAV) {
      const long o = plane_off(p);
#pragma unroll
      for (int d = 0; d < 3; d++) g[d] = grav[o + (long)d * A.pitch_var];
    } else {
      g[0] = g[1] = g[2] = 0.0;
    }
  };

  // ---- register state carried along z --------------------------------------
  double qa[NV], qb[NV], qc[NV];  // primitives of planes c-1, c, c+1
  double ucur[NV];                // conservative u of plane c
  double unxt[NV];                // conservative u of plane c+1
  double qmz[NV];                 // qm along z of plane c-1 (state on its +z face)
  double part[NV];                // u + x and y flux differences of plane c-1
  double fzlo[NV];                // z flux through the -z face of plane c-1
  double upre[NV], gpre[3];       // prefetch of plane c+2

  // prologue: planes z0-2, z0-1, z0  (c starts at z0-1)
  {
    double u[NV], g[3];
    load_u(z0 - 2, u); load_g(z0 - 2, g);
    ctoprim_cell<NV, GRAV>(u, g, dtxhalf, P, qa);
    load_u(z0 - 1, ucur); load_g(z0 - 1, g);
    ctoprim_cell<NV, GRAV>(ucur, g, dtxhalf, P, qb);
    load_u(z0, upre); load_g(z0, gpre);
#pragma unroll
    for (int n = 0; n < NV; n++) { qmz[n] = 0.0; part[n] = 0.0; fzlo[n] = 0.0; }
    // primitives of plane c = z0-1 into LDS
    Plane<BY> &qs = qbuf[(z0 - 1) & 1];
#pragma unroll
    for (int n = 0; n < NV; n++) qs.v[n][ty][tx] = qb[n];
  }
  __syncthreads();

  const int txm = max(tx - 1, 0), txp = min(tx + 1, BX - 1);
  const int tym = max(ty - 1, 0), typ = min(ty + 1, BY - 1);

  for (int c = z0 - 1; c <= z1; c++) {
    // ---- plane c+1 arrives; prefetch plane c+2 ------------------------------
    {
      double g[3];
#pragma unroll
      for (int n = 0; n < NV; n++) unxt[n] = upre[n];
#pragma unroll
      for (int d = 0; d < 3; d++) g[d] = gpre[d];
      if (c + 2 <= z1 + 1) { load_u(c + 2, upre); load_g(c + 2, gpre); }
      ctoprim_cell<NV, GRAV>(unxt, g, dtxhalf, P, qc);
      Plane<BY> &qn = qbuf[(c + 1) & 1];
#pragma unroll
      for (int n = 0; n < NV; n++) qn.v[n][ty][tx] = qc[n];
    }
    // No barrier here: plane c's primitives were written one iteration ago
    // (two barriers back); qbuf[(c+1)&1] was last read before B2 of c-1.

    const bool do_xy = (c >= z0) && (c < z1);
    double qpx[NV], qpy[NV], qpz[NV], qmz_new[NV];
    if (r_trace) {
      const Plane<BY> &qs = qbuf[c & 1];
      double dq[3][NV];
#pragma unroll
      for (int n = 0; n < NV; n++) {
        const double q0 = qb[n];
        dq[0][n] = slope1<ST>(qs.v[n][ty][txm], q0, qs.v[n][ty][txp], P);
        dq[1][n] = slope1<ST>(qs.v[n][tym][tx], q0, qs.v[n][typ][tx], P);
        dq[2][n] = slope1<ST>(qa[n], q0, qc[n], P);
      }
      double qm[3][NV], qp[3][NV];
      trace3d_cell<NV>(qb, dq, dtdx, dtdx, dtdx, P, qm, qp);
#pragma unroll
      for (int n = 0; n < NV; n++) {
        smx->v[n][ty][tx] = qm[0][n];
        smy->v[n][ty][tx] = qm[1][n];
        qpx[n] = qp[0][n]; qpy[n] = qp[1][n]; qpz[n] = qp[2][n];
        qmz_new[n] = qm[2][n];
      }
    }
    __syncthreads();  // (B2) traced states visible

    double fx[NV], fy[NV], fz[NV];
#pragma unroll
    for (int n = 0; n < NV; n++) { fx[n] = 0.0; fy[n] = 0.0; fz[n] = 0.0; }
    double un_, ef_;
    if (r_fy && do_xy) {
      double qL[NV];
#pragma unroll
      for (int n = 0; n < NV; n++) qL[n] = smy->v[n][tym][tx];
      interface_flux<RS, NV, 1>(qL, qpy, P, fy, un_, ef_);
#pragma unroll
      for (int n = 0; n < NV; n++) {
        fy[n] = DXPOW2 ? fy[n] * A.dt * A.rdx : fy[n] * A.dt / A.dx;
        fyb->v[n][ty][tx] = fy[n];
      }
    }
    if (r_fxz) {
      if (do_xy) {
        double qL[NV];
#pragma unroll
        for (int n = 0; n < NV; n++) qL[n] = smx->v[n][ty][txm];
        interface_flux<RS, NV, 0>(qL, qpx, P, fx, un_, ef_);
#pragma unroll
        for (int n = 0; n < NV; n++) {
          fx[n] = DXPOW2 ? fx[n] * A.dt * A.rdx : fx[n] * A.dt / A.dx;
          fxb->v[n][ty][tx] = fx[n];
        }
      }
      if (c >= z0) {
        // z flux through the face between planes c-1 and c
        interface_flux<RS, NV, 2>(qmz, qpz, P, fz, un_, ef_);
#pragma unroll
        for (int n = 0; n < NV; n++)
          fz[n] = DXPOW2 ? fz[n] * A.dt * A.rdx : fz[n] * A.dt / A.dx;
      }
    }
    __syncthreads();  // (B3) x/y fluxes visible

    if (r_fxz) {
      // finish plane c-1: its +z face flux is fz
      if (c >= z0 + 1) {
        if (r_upd) {
          const long o = plane_off(c - 1);
#pragma unroll
          for (int n = 0; n < NV; n++)
            unew[o + (long)n * A.pitch_var] = part[n] + (fzlo[n] - fz[n]);
        }
      }
      if (do_xy) {
#pragma unroll
        for (int n = 0; n < NV; n++) {
          double t = ucur[n] + (fx[n] - fxb->v[n][ty][txp]);
          part[n] = t + (fy[n] - fyb->v[n][typ][tx]);
        }
      }
#pragma unroll
      for (int n = 0; n < NV; n++) fzlo[n] = fz[n];
    }
    // rotate the z window
#pragma unroll
    for (int n = 0; n < NV; n++) {
      qa[n] = qb[n]; qb[n] = qc[n];
      ucur[n] = unxt[n];
      qmz[n] = qmz_new[n];
    }
  }
}

// ---------------------------------------------------------------------------
// host-side dispatch
// ---------------------------------------------------------------------------
template <int ST, int RS, int BY, bool GRAV>
static hipError_t launch2(const SweepArgs &A, bool pow2, hipStream_t s) {
  const size_t lds = 6 * sizeof(Plane<BY>);
  dim3 block(BX, BY);
  dim3 grid(A.ntx * A.nty * A.ntz);
  hipError_t e;
  if (pow2) {
    auto k = godunov_sweep_kernel<ST, RS, BY, GRAV, true>;
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k, grid, block, lds, s, A);
  } else {
    auto k = godunov_sweep_kernel<ST, RS, BY, GRAV, false>;
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k, grid, block, lds, s, A);
  }
  return hipGetLastError();
}

template <int ST, int RS>
static hipError_t launch1(SweepArgs &A, int by, bool grav, bool pow2, hipStream_t s) {
  A.ntx = (A.nx + (BX - 4) - 1) / (BX - 4);
  A.nty = (A.ny + (by - 4) - 1) / (by - 4);
  A.ntz = (A.nz + A.zchunk - 1) / A.zchunk;
  if (by == 8) return grav ? launch2<ST, RS, 8, true>(A, pow2, s) : launch2<ST, RS, 8, false>(A, pow2, s);
  return hipErrorInvalidValue;
}

template <int ST>
static hipError_t launch0(SweepArgs &A, int rs, int by, bool grav, bool pow2, hipStream_t s) {
  switch (rs) {
    case RIEMANN_LLF: return launch1<ST, RIEMANN_LLF>(A, by, grav, pow2, s);
    case RIEMANN_HLLC: return launch1<ST, RIEMANN_HLLC>(A, by, grav, pow2, s);
    case RIEMANN_HLL: return launch1<ST, RIEMANN_HLL>(A, by, grav, pow2, s);
    case RIEMANN_ACOUSTIC: return launch1<ST, RIEMANN_ACOUSTIC>(A, by, grav, pow2, s);
    case RIEMANN_EXACT: return launch1<ST, RIEMANN_EXACT>(A, by, grav, pow2, s);
  }
  return hipErrorInvalidValue;
}

hipError_t launch_godunov_sweep(SweepArgs &A, int slope_type, int riemann, int by,
                                bool grav, bool pow2, hipStream_t s) {
  switch (slope_type) {
    case 0: return launch0<0>(A, riemann, by, grav, pow2, s);
    case 1: return launch0<1>(A, riemann, by, grav, pow2, s);
    case 2: return launch0<2>(A, riemann, by, grav, pow2, s);
    case 7: return launch0<7>(A, riemann, by, grav, pow2, s);
    case 8: return launch0<8>(A, riemann, by, grav, pow2, s);
  }
  return hipErrorInvalidValue;
}

}  // namespace SWEEP_NS
}  // namespace ramses_amd
