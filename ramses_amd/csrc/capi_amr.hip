// capi_amr.hip -- residency for AMR runs: the reference's own cell vectors
// uold/unew(1:ncell,1:nvar) and tree arrays (son, nbor, father) stay on the device between the routines of
// amr_step that touch the hydro state, instead of crossing PCIe around every call:
//   set_unew      hydro/godunov_fine.f90:40-130        lvl_copy_kernel
//   godunov_fine  hydro/godunov_fine.f90:5-35,486-911  the tree-walking sweep (amr_sweep.hip) on the resident arrays
//   set_uold      hydro/godunov_fine.f90:135-232       lvl_set_uold_kernel (incl. the passive-scalar floor fix :176-190)
//   upload_fine   hydro/interpol_hydro.f90:5-263       lvl_upload_kernel (upl: restriction, interpol_var 0/1/2)
//   courant_fine  hydro/courant_fine.f90:1-159         lvl_courant_kernel (cmpdt on the leaf cells)
//   hydro_flag    hydro/hydro_flag.f90:1-178 + hydro_refine hydro/godunov_utils.f90:125-263   lvl_flag_kernel
// The mesh itself stays the reference's host code: before refine_fine rebuilds levels the shim brings the
// levels it reads back to the host (sync_level), and afterwards the tree and the rebuilt levels are sent
// again (tree, load_level) -- the other levels never leave the device.  NDIM=3, hydro only.
// The octs carry the DEVICE's own numbers here (csrc/amr_layout.hpp: levels in tiles of 32 x 4 x 4 octs, so that the dense
// sweep reads them in 256-byte runs): `ngridmax` / `ncell` of the kernels below are the device's, oct lists are translated as
// they arrive (set_level, upload_list), the host arrays are addressed with R.ngh / R.ncell_h.
// Compiled with -ffp-contract=off: the reference's operation order.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/ramses_amd.h"
#include "amr_core.hpp"
#include "amr_layout.hpp"
#include "amr_sweep_args.hpp"
#include "amr_tree.hpp"
#include "hydro_core.hpp"
#include "rho_args.hpp"
#include "sweep_args.hpp"

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

struct LvlArgs {
  double *uold, *unew;
  const int *son, *nbor;
  const int *igrid;
  int ngrid, nvar;
  long ncell, ncoarse, ngridmax;
};

// dst(cells of the level's octs, 1:nvar) = src(...)
__global__ __launch_bounds__(256) void lvl_copy_kernel(LvlArgs A, double *__restrict__ dst, const double *__restrict__ src) {
  const long total = (long)A.ngrid * 8;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    const int ind = (int)(t / A.ngrid), i = (int)(t % A.ngrid);
    const long c = A.ncoarse + (long)ind * A.ngridmax + A.igrid[i] - 1;
    for (int v = 0; v < A.nvar; v++) dst[c + (long)v * A.ncell] = src[c + (long)v * A.ncell];
  }
}

// set_uold: the passive-scalar fix near the density floor (:176-190), then uold = unew
__global__ __launch_bounds__(256) void lvl_set_uold_kernel(LvlArgs A, double smallr) {
  const long total = (long)A.ngrid * 8;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    const int ind = (int)(t / A.ngrid), i = (int)(t % A.ngrid);
    const long c = A.ncoarse + (long)ind * A.ngridmax + A.igrid[i] - 1;
    if (A.nvar > 5) {
      const double ro = A.uold[c], rn = A.unew[c];
      if (ro < smallr && rn > ro) {
        for (int v = 5; v < A.nvar; v++) A.unew[c + (long)v * A.ncell] = A.uold[c + (long)v * A.ncell] * __builtin_fmax(rn, smallr) / smallr;
      } else if (rn < smallr && ro > rn) {
        for (int v = 5; v < A.nvar; v++) A.unew[c + (long)v * A.ncell] = A.uold[c + (long)v * A.ncell] * smallr / __builtin_fmax(ro, smallr);
      }
    }
    for (int v = 0; v < A.nvar; v++) A.uold[c + (long)v * A.ncell] = A.unew[c + (long)v * A.ncell];
  }
}

// upload_fine / upl: every split cell of the level = mean of its 8 children (density floored; internal-energy
// averaging with interpol_var 1|2)
template <int NV>
__global__ __launch_bounds__(256) void lvl_upload_kernel(LvlArgs A, int interpol_var, double smallr) {
  const long total = (long)A.ngrid * 8;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    const int ind = (int)(t / A.ngrid), i = (int)(t % A.ngrid);
    const long c = A.ncoarse + (long)ind * A.ngridmax + A.igrid[i] - 1;
    const int gs = A.son[c];
    if (gs <= 0) continue;
    double ch[8][NV];
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const long cs = A.ncoarse + (long)k * A.ngridmax + gs - 1;
#pragma unroll
      for (int v = 0; v < NV; v++) ch[k][v] = A.uold[cs + (long)v * A.ncell];
    }
    double pa[NV];
    upl_cell<NV>(ch, interpol_var, smallr, pa);
#pragma unroll
    for (int v = 0; v < NV; v++) A.uold[c + (long)v * A.ncell] = pa[v];
  }
}

__device__ __forceinline__ double wsum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}
__device__ __forceinline__ double wmin(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = __builtin_fmin(v, __shfl_down(v, off, 64));
  return v;
}

// courant_fine: cmpdt over the leaf cells of the level (dt exact: a minimum), mass / energy sums (diagnostics)
template <bool GRAV>
__global__ __launch_bounds__(256) void lvl_courant_kernel(LvlArgs A, const double *__restrict__ f, HydroConst P, double dx, double vol,
                                                          double courant_factor, double dt_init, double *__restrict__ out) {
  double dtmin = dt_init, mass = 0.0, etot = 0.0, eint = 0.0;
  const long total = (long)A.ngrid * 8;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    const int ind = (int)(t / A.ngrid), i = (int)(t % A.ngrid);
    const long c = A.ncoarse + (long)ind * A.ngridmax + A.igrid[i] - 1;
    if (A.son[c] != 0) continue;
    double u[5], g[3] = {0.0, 0.0, 0.0};
#pragma unroll
    for (int v = 0; v < 5; v++) u[v] = A.uold[c + (long)v * A.ncell];
    if (GRAV) {
#pragma unroll
      for (int k = 0; k < 3; k++) g[k] = f[c + (long)k * A.ncell];
    }
    dtmin = __builtin_fmin(dtmin, cmpdt_cell<5, GRAV>(u, g, dx, courant_factor, P, 3.0));
    mass += u[0] * vol;
    etot += u[4] * vol;
    double ei = u[4] * vol;
    const double rho = __builtin_fmax(u[0], P.smallr);
    ei -= 0.5 * (u[1] * u[1]) / rho * vol;
    ei -= 0.5 * (u[2] * u[2]) / rho * vol;
    ei -= 0.5 * (u[3] * u[3]) / rho * vol;
    eint += ei;
  }
  dtmin = wmin(dtmin); mass = wsum(mass); etot = wsum(etot); eint = wsum(eint);
  __shared__ double red[4][4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { red[wave][0] = dtmin; red[wave][1] = mass; red[wave][2] = etot; red[wave][3] = eint; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double d = red[0][0], m = red[0][1], e = red[0][2], ei = red[0][3];
    for (int w = 1; w < 4; w++) { d = __builtin_fmin(d, red[w][0]); m += red[w][1]; e += red[w][2]; ei += red[w][3]; }
    atomicMin(reinterpret_cast<unsigned long long *>(out), (unsigned long long)__double_as_longlong(d));
    // the diagnostics: one partial per workgroup, added in workgroup order by lvl_courant_final_kernel (reproducible
    // prints of mcons / econs; the grid is fixed by the level's size, so the same run gives the same digits)
    double *part = out + 4 + 3 * (long)blockIdx.x;
    part[0] = m; part[1] = e; part[2] = ei;
  }
}
__global__ void lvl_courant_init_kernel(double *out, double dt_init) { out[0] = dt_init; out[1] = out[2] = out[3] = 0.0; }
__global__ void lvl_courant_final_kernel(double *out, int nblocks) {
  const int k = threadIdx.x;     // 0: mass, 1: total energy, 2: internal energy
  if (k >= 3) return;
  double s = 0.0;
  for (int b = 0; b < nblocks; b++) s += out[4 + 3 * (long)b + k];
  out[1 + k] = s;
}

// hydro_flag's gradient criteria (hydro_refine): ok[ind*ngrid+i] = 1 when the cell is to be refined
struct FlagCrit { double err_grad_d, err_grad_p, err_grad_u, floor_d, floor_p, floor_u, gamma, smallr; };
__device__ __forceinline__ void refine_prim(const LvlArgs &A, long c, const FlagCrit &F, double (&q)[5]) {
  // conservative -> (rho, u, v, w, P) as hydro_refine does (:150-190)
  double u[5];
#pragma unroll
  for (int v = 0; v < 5; v++) u[v] = A.uold[c + (long)v * A.ncell];
  q[0] = __builtin_fmax(u[0], F.smallr);
  double ek = 0.0;
#pragma unroll
  for (int d = 0; d < 3; d++) { q[1 + d] = u[1 + d] / q[0]; }
#pragma unroll
  for (int d = 0; d < 3; d++) ek = ek + 0.5 * q[0] * (q[1 + d] * q[1 + d]);
  q[4] = (F.gamma - 1.0) * (u[4] - ek);
}
// out: the cells that ask for refinement as a compact list of 1-based cell indices (one atomic per wavefront; the order
// is whatever the waves arrive in -- the host sorts the few that come back)
__global__ __launch_bounds__(256) void lvl_flag_kernel(LvlArgs A, FlagCrit F, int *__restrict__ list, int *__restrict__ count) {
  const long total = (long)A.ngrid * 8;
  const long span = (long)gridDim.x * blockDim.x;
  for (long t0 = (long)blockIdx.x * blockDim.x; t0 < total; t0 += span) {
    const long t = t0 + threadIdx.x;
    bool flag = false;
    long c = 0;
    if (t < total) {
    const int ind = (int)(t / A.ngrid), i = (int)(t % A.ngrid);
    const int g = A.igrid[i];
    c = A.ncoarse + (long)ind * A.ngridmax + g - 1;
    double qm[5];
    refine_prim(A, c, F, qm);
#pragma unroll
    for (int d = 0; d < 3; d++) {
      // neighbour cells in direction d (getnborcells); a missing one is replaced by the neighbouring father cell
      long cn[2];
#pragma unroll
      for (int side = 0; side < 2; side++) {
        const int bit = (ind >> d) & 1;
        if (bit != side) {
          cn[side] = c + (side ? 1 : -1) * ((long)(1 << d) * A.ngridmax);
        } else {
          const int nb = A.nbor[(long)(2 * d + side) * A.ngridmax + g - 1];
          const int g2 = A.son[nb - 1];
          cn[side] = g2 > 0 ? A.ncoarse + (long)(ind ^ (1 << d)) * A.ngridmax + g2 - 1 : (long)nb - 1;
        }
      }
      double qg[5], qd[5];
      refine_prim(A, cn[0], F, qg);
      refine_prim(A, cn[1], F, qd);
      if (F.err_grad_d >= 0.0) {
        const double e = 2.0 * __builtin_fmax(__builtin_fabs((qd[0] - qm[0]) / (qd[0] + qm[0] + F.floor_d)),
                                              __builtin_fabs((qm[0] - qg[0]) / (qm[0] + qg[0] + F.floor_d)));
        flag = flag || e > F.err_grad_d;
      }
      if (F.err_grad_p >= 0.0) {
        const double e = 2.0 * __builtin_fmax(__builtin_fabs((qd[4] - qm[4]) / (qd[4] + qm[4] + F.floor_p)),
                                              __builtin_fabs((qm[4] - qg[4]) / (qm[4] + qg[4] + F.floor_p)));
        flag = flag || e > F.err_grad_p;
      }
      if (F.err_grad_u >= 0.0) {
        const double cg = __builtin_sqrt(__builtin_fmax(F.gamma * qg[4] / qg[0], F.floor_u * F.floor_u));
        const double cm = __builtin_sqrt(__builtin_fmax(F.gamma * qm[4] / qm[0], F.floor_u * F.floor_u));
        const double cd = __builtin_sqrt(__builtin_fmax(F.gamma * qd[4] / qd[0], F.floor_u * F.floor_u));
#pragma unroll
        for (int k = 0; k < 3; k++) {
          const double vg = qg[1 + k], vm = qm[1 + k], vd = qd[1 + k];
          const double e = 2.0 * __builtin_fmax(__builtin_fabs((vd - vm) / (cd + cm + __builtin_fabs(vd) + __builtin_fabs(vm) + F.floor_u)),
                                                __builtin_fabs((vm - vg) / (cm + cg + __builtin_fabs(vm) + __builtin_fabs(vg) + F.floor_u)));
          flag = flag || e > F.err_grad_u;
        }
      }
    }
    }
    const unsigned long long m = __ballot(flag);
    if (m) {
      const int lane = threadIdx.x & 63;
      int base = 0;
      if (lane == 0) base = atomicAdd(count, __popcll(m));
      base = __shfl(base, 0, 64);
      if (flag) list[base + __popcll(m & ((1ull << lane) - 1ull))] = (int)(c + 1);
    }
  }
}

// compact buffer buf[v][ind*ngrid+i] <-> cell vector (sync_level / load_level)
template <bool GATHER>
__global__ __launch_bounds__(256) void lvl_pack_kernel(LvlArgs A, double *__restrict__ buf) {
  const long total = (long)A.ngrid * 8;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    const int ind = (int)(t / A.ngrid), i = (int)(t % A.ngrid);
    const long c = A.ncoarse + (long)ind * A.ngridmax + A.igrid[i] - 1;
    for (int v = 0; v < A.nvar; v++) {
      if (GATHER) buf[(long)v * total + t] = A.uold[c + (long)v * A.ncell];
      else A.uold[c + (long)v * A.ncell] = buf[(long)v * total + t];
    }
  }
}


// synchro_hydro_fine (hydro/synchro_hydro_fine.f90:5-136): momentum kick d*f*dteff with the kinetic energy taken out of
// and put back into the total energy, on the level's cells
__global__ __launch_bounds__(256) void lvl_synchro_kernel(LvlArgs A, const double *__restrict__ f, double dteff, double smallr) {
  const long total = (long)A.ngrid * 8;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    const int ind = (int)(t / A.ngrid), i = (int)(t % A.ngrid);
    const long c = A.ncoarse + (long)ind * A.ngridmax + A.igrid[i] - 1;
    const double d = __builtin_fmax(A.uold[c], smallr);
    double m[3] = {A.uold[c + A.ncell], A.uold[c + 2 * A.ncell], A.uold[c + 3 * A.ncell]};
    double pp = A.uold[c + 4 * A.ncell];
#pragma unroll
    for (int k = 0; k < 3; k++) pp = pp - 0.5 * (m[k] * m[k]) / d;
#pragma unroll
    for (int k = 0; k < 3; k++) {
      m[k] = m[k] + d * f[c + (long)k * A.ncell] * dteff;
      A.uold[c + (long)(k + 1) * A.ncell] = m[k];
    }
#pragma unroll
    for (int k = 0; k < 3; k++) pp = pp + 0.5 * (m[k] * m[k]) / d;
    A.uold[c + 4 * A.ncell] = pp;
  }
}

// add_gravity_source_terms (hydro/godunov_fine.f90:237-289) on unew of the level's cells (strict_equilibrium = 0)
__global__ __launch_bounds__(256) void lvl_gravity_source_kernel(LvlArgs A, const double *__restrict__ f, double dt, double smallr) {
  const long total = (long)A.ngrid * 8;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    const int ind = (int)(t / A.ngrid), i = (int)(t % A.ngrid);
    const long c = A.ncoarse + (long)ind * A.ngridmax + A.igrid[i] - 1;
    const long N = A.ncell;
    const double d = __builtin_fmax(A.unew[c], smallr);
    double u = A.unew[c + N] / d, v = A.unew[c + 2 * N] / d, w = A.unew[c + 3 * N] / d;
    double e_kin = 0.5 * d * (u * u + v * v + w * w);
    const double e_prim = A.unew[c + 4 * N] - e_kin;
    const double d_old = __builtin_fmax(A.uold[c], smallr);
    const double req = 0.0;
    const double fact = (d_old - req) / d * 0.5 * dt;
    u = u + f[c] * fact;
    A.unew[c + N] = d * u;
    v = v + f[c + N] * fact;
    A.unew[c + 2 * N] = d * v;
    w = w + f[c + 2 * N] * fact;
    A.unew[c + 3 * N] = d * w;
    e_kin = 0.5 * d * (u * u + v * v + w * w);
    A.unew[c + 4 * N] = e_prim + e_kin;
  }
}

// one variable (or ncomp of them, stride ncell) of the level's cells <-> packed [ncomp][8*ngrid]
template <bool GATHER>
__global__ void lvl_pack_comp_kernel(double *vec, double *buf, const int *igrid, int ngrid, int ncomp, long ncell, long ncoarse, long ngridmax) {
  const long total = (long)ngrid * 8;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    const long c = ncoarse + (long)(t / ngrid) * ngridmax + igrid[t % ngrid] - 1;
    for (int k = 0; k < ncomp; k++) {
      if (GATHER) buf[(long)k * total + t] = vec[c + (long)k * ncell];
      else vec[c + (long)k * ncell] = buf[(long)k * total + t];
    }
  }
}

// make_virtual_reverse_dp's scatter (amr/virtual_boundaries.f90:857-867): vec(emission cells) += what the peer's virtual cells held
__global__ void lvl_acc_comp_kernel(double *vec, const double *buf, const int *igrid, int ngrid, int ncomp, long ncell, long ncoarse, long ngridmax) {
  const long total = (long)ngrid * 8;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    const long c = ncoarse + (long)(t / ngrid) * ngridmax + igrid[t % ngrid] - 1;
    for (int k = 0; k < ncomp; k++) vec[c + (long)k * ncell] = vec[c + (long)k * ncell] + buf[(long)k * total + t];
  }
}
// set_unew's loop over the virtual octs (hydro/godunov_fine.f90:92-122): vec(cells of the octs, 1:ncomp) = 0
__global__ void lvl_zero_comp_kernel(double *vec, const int *igrid, int ngrid, int ncomp, long ncell, long ncoarse, long ngridmax) {
  const long total = (long)ngrid * 8;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    const long c = ncoarse + (long)(t / ngrid) * ngridmax + igrid[t % ngrid] - 1;
    for (int k = 0; k < ncomp; k++) vec[c + (long)k * ncell] = 0.0;
  }
}

// ---- pressure_fix (hydro/godunov_fine.f90:66-83, 203-227, 294-481) -------------------------------------------------------
// set_unew: divu = 0, enew = internal energy of uold
__global__ __launch_bounds__(256) void lvl_pfix_init_kernel(LvlArgs A, double *__restrict__ divu, double *__restrict__ enew, double smallr) {
  const long total = (long)A.ngrid * 8;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    const long c = A.ncoarse + (long)(t / A.ngrid) * A.ngridmax + A.igrid[t % A.ngrid] - 1;
    const long N = A.ncell;
    const double d = __builtin_fmax(A.uold[c], smallr);
    const double u = A.uold[c + N] / d, v = A.uold[c + 2 * N] / d, w = A.uold[c + 3 * N] / d;
    divu[c] = 0.0;
    enew[c] = A.uold[c + 4 * N] - 0.5 * d * (u * u + v * v + w * w);
  }
}
// add_pdv_source_terms: enew -= (gamma-1) e_old div(u) dt, div(u) from the normal velocities of the two neighbours per
// direction (a neighbour oct that does not exist: the father's neighbour cell at 1.5 dx)
__global__ __launch_bounds__(256) void lvl_pdv_kernel(LvlArgs A, double *__restrict__ enew, double dx_loc, double dt, double gamma, double smallr) {
  const long total = (long)A.ngrid * 8;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    const int ind = (int)(t / A.ngrid), i = (int)(t % A.ngrid);
    const int g = A.igrid[i];
    const long N = A.ncell;
    const long c = A.ncoarse + (long)ind * A.ngridmax + g - 1;
    double divu_loc = 0.0;
#pragma unroll
    for (int d = 0; d < 3; d++) {
      const int bit = (ind >> d) & 1, jnd = ind ^ (1 << d);
      double vel[2], dxs[2];
#pragma unroll
      for (int side = 0; side < 2; side++) {
        long cn;
        dxs[side] = dx_loc;
        if (bit != side) {
          cn = A.ncoarse + (long)jnd * A.ngridmax + g - 1;             // inside the oct
        } else {
          const int nb = A.nbor[(long)(2 * d + side) * A.ngridmax + g - 1];
          const int g2 = A.son[nb - 1];
          if (g2 > 0) cn = A.ncoarse + (long)jnd * A.ngridmax + g2 - 1;
          else { cn = nb - 1; dxs[side] = dx_loc * 1.5; }
        }
        vel[side] = A.uold[cn + (long)(d + 1) * N] / __builtin_fmax(A.uold[cn], smallr);
      }
      divu_loc = divu_loc + (vel[1] - vel[0]) / (dxs[0] + dxs[1]);
    }
    const double dd = __builtin_fmax(A.uold[c], smallr);
    const double u = A.uold[c + N] / dd, v = A.uold[c + 2 * N] / dd, w = A.uold[c + 3 * N] / dd;
    const double eold = A.uold[c + 4 * N] - 0.5 * dd * (u * u + v * v + w * w);
    enew[c] = enew[c] - (gamma - 1.0) * eold * divu_loc * dt;
  }
}
// set_uold's energy switch, after uold = unew
__global__ __launch_bounds__(256) void lvl_pfix_switch_kernel(LvlArgs A, const double *__restrict__ divu, const double *__restrict__ enew, double dx_loc,
                                                              double dt, double beta_fix, double hexp, double smallr) {
  const long total = (long)A.ngrid * 8;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    const long c = A.ncoarse + (long)(t / A.ngrid) * A.ngridmax + A.igrid[t % A.ngrid] - 1;
    const long N = A.ncell;
    const double d = __builtin_fmax(A.uold[c], smallr);
    const double u = A.uold[c + N] / d, v = A.uold[c + 2 * N] / d, w = A.uold[c + 3 * N] / d;
    const double e_kin = 0.5 * d * (u * u + v * v + w * w);
    const double e_cons = A.uold[c + 4 * N] - e_kin;
    const double e_prim = enew[c];
    const double div = __builtin_fabs(divu[c]) * dx_loc / dt;
    const double m = __builtin_fmax(div, 3.0 * hexp * dx_loc);
    const double e_trunc = beta_fix * d * (m * m);
    if (e_cons < e_trunc) A.uold[c + 4 * N] = e_prim + e_kin;
  }
}

// ---- godunov_fine of a level stored in TILES (csrc/amr_layout.hpp) through the dense sweep -------------------------------
// The tree-walking sweep recomputes an 8^3 stencil per father oct; the dense z-marching sweep (csrc/hydro_sweep.hip) converts,
// traces and solves every cell and face once.  It runs IN PLACE on the device's cell vectors through the level's tile
// directory; what it needs beyond them is prepared once per (tree, oct list) -- the PLAN of the level:
//   * the status byte of every cell: refined (set with the tree), OWNED = in the call's list (updated), GHOST;
//   * the GHOST octs: the positions next to a listed oct (its 26 neighbours) where the level has no oct.  The reference
//     interpolates those from the father cell and its six neighbours for every oct that needs them (hydro/godunov_fine.f90:
//     563-600, interpol_hydro); here a pre-pass writes them once per sweep into the free slots of the tiles (plan_ghost_fill_kernel:
//     the same function of the same values), the acceleration of the father cell with them (:616-626);
//   * the work list: the 60 x 8-column tiles and plane ranges that hold listed cells;
//   * the flux records' targets (the leaf cell of the coarser level behind each oct face, :798-908) and the position of every
//     listed oct in the list (the replay kernel's order).
struct PlanArgs {
  const int *son, *nbor, *father, *iperm;
  unsigned char *stat;
  int *octpos;
  const int *ig;               // the call's list, device indices
  int n;
  long ncell, ncoarse, ngd;
  // the level's tiles
  const int *dir, *tileid;
  long base;                   // first device index of the level (1-based)
  int no, ntx, nty, ntz;
};
// coordinates of a device oct of a level in tiles
__device__ __forceinline__ void plan_oct_pos(const PlanArgs &A, int d, int &x, int &y, int &z) {
  const long r = (long)d - A.base;
  const int t = A.tileid[r / TILE_OCTS], l = (int)(r % TILE_OCTS);
  const int tx = t % A.ntx, ty = (t / A.ntx) % A.nty, tz = t / (A.ntx * A.nty);
  x = tx * TILE_OX + l % TILE_OX; y = ty * TILE_OY + (l / TILE_OX) % TILE_OY; z = tz * TILE_OZ + l / (TILE_OX * TILE_OY);
}
// device oct index (1-based) at a position, 0: no tile there
__device__ __forceinline__ int plan_oct_at(const PlanArgs &A, int x, int y, int z) {
  const int t = (x / TILE_OX) + A.ntx * ((y / TILE_OY) + A.nty * (z / TILE_OZ));
  const int c0 = A.dir[t];
  if (c0 < 0) return 0;
  return (int)(c0 - A.ncoarse + 1) + (x % TILE_OX) + TILE_OX * ((y % TILE_OY) + TILE_OY * (z % TILE_OZ));
}
// a range of device octs of every octant position: status bits cleared, list positions forgotten
__global__ __launch_bounds__(256) void plan_clear_kernel(unsigned char *stat, int *octpos, long ncoarse, long ngd, long base, long cap, int *gfather) {
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < cap * 8; t += (long)gridDim.x * blockDim.x) {
    const int ind = (int)(t / cap);
    const long o = t % cap;
    const long c = ncoarse + (long)ind * ngd + base - 1 + o;
    stat[c] &= (unsigned char)CELL_REFINED;
    if (ind == 0) { octpos[base - 1 + o] = -1; if (gfather) gfather[o] = 0; }
  }
}
__global__ __launch_bounds__(256) void plan_owned_kernel(PlanArgs A) {
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < (long)A.n * 8; t += (long)gridDim.x * blockDim.x) {
    const int ind = (int)(t / A.n), i = (int)(t % A.n);
    const int d = A.ig[i];
    A.stat[A.ncoarse + (long)ind * A.ngd + d - 1] |= (unsigned char)CELL_OWNED;
    if (ind == 0) A.octpos[d - 1] = i;
  }
}
// the ghost octs of the listed octs: slot (0-based device oct index) and father cell, each once
__global__ __launch_bounds__(256) void plan_ghost_kernel(PlanArgs A, int *gfather, int *gslot, int *gcell, int *count, int cap, int *err) {
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < (long)A.n * 26; t += (long)gridDim.x * blockDim.x) {
    const int i = (int)(t / 26);
    int o = (int)(t % 26);
    if (o >= 13) o++;                                   // skip the oct itself
    const int sx = o % 3 - 1, sy = (o / 3) % 3 - 1, sz = o / 9 - 1;
    const int d = A.ig[i];
    int x, y, z;
    plan_oct_pos(A, d, x, y, z);
    const int m = A.no - 1;
    const int dn = plan_oct_at(A, (x + sx) & m, (y + sy) & m, (z + sz) & m);
    if (dn == 0) { atomicAdd(err, 1); continue; }       // (the layout gave every neighbour position a tile)
    if (A.iperm[dn - 1] != 0) continue;                  // an oct of the tree: nothing to interpolate
    // its father cell: from the oct's own father cell, x, then y, then z steps through son(nbor(...)) (getnborfather)
    AmrSweepArgs T;
    T.son = A.son; T.nbor = A.nbor; T.ncoarse = A.ncoarse; T.ngridmax = A.ngd;
    int c = A.father[d - 1];
    const int st[3] = {sx, sy, sz};
#pragma unroll
    for (int axis = 0; axis < 3; axis++)
      if (st[axis] != 0 && c > A.ncoarse) c = amrsweep::nbor_cell(c, 2 * axis + (st[axis] > 0 ? 1 : 0), T);
    if (c <= A.ncoarse || A.son[c - 1] != 0) { atomicAdd(err, 1); continue; }
    if (atomicCAS(&gfather[dn - A.base], 0, c) != 0) continue;
    const int pos = atomicAdd(count, 1);
    if (pos < cap) { gslot[pos] = dn - 1; gcell[pos] = c; }
    for (int ind = 0; ind < 8; ind++) A.stat[A.ncoarse + (long)ind * A.ngd + dn - 1] |= (unsigned char)CELL_GHOST;
  }
}
// the leaf cell of the coarser level behind each (listed oct, face), 0: the neighbour oct exists; indexed by the oct's slot in
// the level's index range (device index - base), like the flux records
__global__ __launch_bounds__(256) void plan_target_kernel(PlanArgs A, int *corr_tgt) {
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < (long)A.n * 6; t += (long)gridDim.x * blockDim.x) {
    const int i = (int)(t / 6), f = (int)(t % 6);
    const int d = A.ig[i];
    const int nb = A.nbor[(long)f * A.ngd + d - 1];
    corr_tgt[((long)d - A.base) * 6 + f] = (nb > 0 && A.son[nb - 1] == 0) ? nb : 0;
  }
}
// Conservative update at level ilevel-1 (hydro/godunov_fine.f90:798-908) from the records the dense sweep filed: every (oct,
// face) whose neighbouring father cell is a leaf owes it 4 fluxes.  A coarse cell has at most 6 such creditors; the thread of
// the creditor that comes first in the reference's loop order (batch of nvector octs of the LIST, idim, left before right)
// replays all of them sequentially: floating-point addition is not associative.  (The twin of amr_coarse_update_kernel,
// csrc/amr_sweep.hip, with the records indexed by device oct instead of list position.)
// the (list position, face) pairs that owe something: io * 6 + f, in whatever order the waves arrive (the replay picks the
// first creditor of a coarse cell by the reference's key, not by this order)
__global__ __launch_bounds__(256) void plan_events_kernel(PlanArgs A, const int *__restrict__ corr_tgt, int *__restrict__ events, int *__restrict__ count,
                                                          int *__restrict__ evt_of) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  bool have = false;
  long sf = 0;
  if (t < (long)A.n * 6) { sf = ((long)A.ig[t / 6] - A.base) * 6 + t % 6; have = corr_tgt[sf] > 0; }
  const unsigned long long m = __ballot(have);
  if (m) {
    const int lane = threadIdx.x & 63;
    int b = 0;
    if (lane == 0) b = atomicAdd(count, __popcll(m));
    b = __shfl(b, 0, 64);
    if (have) events[b + __popcll(m & ((1ull << lane) - 1ull))] = (int)t;      // (sorted and indexed by the caller: evt_of)
  }
}
// the events by (face, device oct): neighbouring threads of the surface pass then read neighbouring octs of a tile row -- its
// gathers are isolated 8-byte words otherwise, a cache line each (profiles/r06_tile_sweep_pmc.txt: 3 GB fetched for 35 MB of records)
// (order 0: (face, oct); 1: (tile of 512 octs, face, oct in the tile); 2: (oct, face) -- all keys below 2^35)
__global__ __launch_bounds__(256) void plan_event_keys_kernel(PlanArgs A, const int *__restrict__ events, int nevent, unsigned long long *__restrict__ keys, int order) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= nevent) return;
  const int ev = events[e];
  const unsigned long long ig = (unsigned)A.ig[ev / 6], f = (unsigned)(ev % 6);
  keys[e] = order == 0 ? (f << 32) | ig : (order == 1 ? ((ig >> 9) << 12) | (f << 9) | (ig & 511) : (ig << 3) | f);
}
__global__ __launch_bounds__(256) void plan_event_index_kernel(PlanArgs A, const int *__restrict__ events, int nevent, int *__restrict__ evt_of) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= nevent) return;
  const int ev = events[e];
  evt_of[((long)A.ig[ev / 6] - A.base) * 6 + ev % 6] = e;              // where the surface pass files this (oct, face)'s four fluxes
}
__global__ __launch_bounds__(256) void tile_coarse_update_kernel(PlanArgs A, double *__restrict__ unew, const double *__restrict__ corr,
                                                                 const int *__restrict__ corr_tgt, const int *__restrict__ evt_of,
                                                                 const int *__restrict__ events, int nevent, int nvector, int NV) {
  const long e0 = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e0 >= nevent) return;
  const long ev = events[e0];
  const int io = (int)(ev / 6), f = (int)(ev % 6);
  const long slot = (long)A.ig[io] - A.base;
  const int C = corr_tgt[slot * 6 + f];
  if (C <= A.ncoarse) return;              // (0: nothing owed; a level-1 father cell: the launcher refuses ilevel < 3)
  AmrSweepArgs T;
  T.son = A.son; T.nbor = A.nbor; T.ncoarse = A.ncoarse; T.ngridmax = A.ngd;
  const long mykey = ((long)(io / nvector) * 3 + (f >> 1)) * 2 + (f & 1);
  long key[6], src[6];
  int n = 0;
  bool first = true;
  for (int e = 0; e < 6; e++) {
    // the cell on side e of C; if it is refined, its oct borders C with face e^1
    const int Ne = amrsweep::nbor_cell(C, e, T);
    if (Ne <= 0) continue;
    const int g2 = A.son[Ne - 1];
    if (g2 == 0) continue;
    const int p2 = A.octpos[g2 - 1];
    if (p2 < 0) continue;                       // not an oct of the call's list
    const int f2 = e ^ 1;
    const long s2 = ((long)g2 - A.base) * 6 + f2;
    if (corr_tgt[s2] != C) continue;
    const long k = ((long)(p2 / nvector) * 3 + (f2 >> 1)) * 2 + (f2 & 1);
    key[n] = k; src[n] = s2; n++;
    if (k < mykey) first = false;
  }
  if (!first) return;
  for (int i = 1; i < n; i++) {                 // insertion sort of <= 6 creditors
    const long k = key[i], sv = src[i];
    int j = i - 1;
    while (j >= 0 && key[j] > k) { key[j + 1] = key[j]; src[j + 1] = src[j]; j--; }
    key[j + 1] = k; src[j + 1] = sv;
  }
  const double oneontwotondim = 1.0 / 8.0;
  const int CV = NV + 2;
  for (int v = 0; v < NV; v++) {
    double val = unew[(long)v * A.ncell + C - 1];
    for (int i = 0; i < n; i++) {
      const double *c = corr + (long)evt_of[src[i]] * 4 * CV;
      const bool left = ((src[i] % 6) & 1) == 0;
      for (int q = 0; q < 4; q++) {
        const double t = c[q * CV + v] * oneontwotondim;
        val = left ? val - t : val + t;
      }
    }
    unew[(long)v * A.ncell + C - 1] = val;
  }
}
// which (tile column of 60 x 8 cells, chunk of 8 planes) hold listed cells
__global__ __launch_bounds__(256) void plan_work_kernel(PlanArgs A, int wtx, int rows, int wz, unsigned char *flag) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < A.n; i += (long)gridDim.x * blockDim.x) {
    int x, y, z;
    plan_oct_pos(A, A.ig[i], x, y, z);
    flag[((long)((2 * y) / rows) * wtx + (2 * x) / 60) * wz + (2 * z) / 8] = 1;
  }
}
// the seven cells interpol_hydro reads for a ghost oct: the father cell and its -x,+x,-y,+y,-z,+z neighbours, a neighbour that
// does not exist replaced by the cell of the coarser level there (getnborfather's fallback) -- found once per plan, so that the
// fill kernel of every sweep is loads and arithmetic only (the walk is nothing but dependent nbor -> son loads)
__global__ __launch_bounds__(256) void plan_ghost_stencil_kernel(const int *__restrict__ son, const int *__restrict__ nbor, const int *__restrict__ gcell,
                                                                 int nghost, long ncoarse, long ngd, int *__restrict__ gsten) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long)nghost * 7) return;
  const int j = (int)(t / nghost), i = (int)(t % nghost);
  AmrSweepArgs T;
  T.son = son; T.nbor = nbor; T.ncoarse = ncoarse; T.ngridmax = ngd;
  int c = gcell[i];
  if (j > 0) {
    c = amrsweep::nbor_cell(c, j - 1, T);
    if (c < 0) c = -c;
  }
  gsten[(long)j * nghost + i] = c;
}
// the ghost octs' cells: interpol_hydro of the father cell with its six neighbours, the father cell's acceleration
// (hydro/godunov_fine.f90:563-626) -- into the free slots of the level's tiles
template <int NV>
__global__ __launch_bounds__(128) void plan_ghost_fill_kernel(double *__restrict__ uold, double *__restrict__ grav, const int *__restrict__ gslot,
                                                              const int *__restrict__ gsten, int nghost, long ncell, long ncoarse, long ngd,
                                                              int interpol_var, int interpol_type, double smallr) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nghost) return;
  double u1[7][NV], u2[8][NV];
#pragma unroll
  for (int j = 0; j < 7; j++) {
    const int c = gsten[(long)j * nghost + i];
#pragma unroll
    for (int v = 0; v < NV; v++) u1[j][v] = uold[(long)v * ncell + c - 1];
  }
  interpol_hydro_cell<NV>(u1, u2, interpol_var, interpol_type, smallr);
  const long o = ncoarse + gslot[i];
#pragma unroll
  for (int ind = 0; ind < 8; ind++)
#pragma unroll
    for (int v = 0; v < NV; v++) uold[(long)v * ncell + o + (long)ind * ngd] = u2[ind][v];
  if (grav) {
    const int c0 = gsten[i];
#pragma unroll
    for (int k = 0; k < 3; k++) {
      const double g = grav[(long)k * ncell + c0 - 1];
#pragma unroll
      for (int ind = 0; ind < 8; ind++) grav[(long)k * ncell + o + (long)ind * ngd] = g;
    }
  }
}

// ---- make_boundary_hydro (hydro/hydro_boundary.f90:5-269) on the resident cell vectors ------------------------------
// One physical boundary region of one level: every cell of a boundary oct takes the state of its reference cell -- the
// cell ind_ref(ind) of the oct son(nbor(oct, inbor)) next to it towards the box -- mirrored (reflexive walls: the normal
// momentum changes sign) or copied with the optional no_inflow clamp (free boundaries; the kinetic energy is taken out
// before and put back after, as the reference does).  The reference walks the region's oct list in chunks of NVECTOR octs
// and, inside a chunk, cell index by cell index -- gather the reference cells of the whole chunk, then scatter (:119-262) --
// writing in place.  A boundary oct whose reference oct is itself an oct of the SAME region (regions two octs deep)
// therefore reads that oct's NEW state if the oct sits in an earlier chunk, or in the same chunk and the reference cell
// has a smaller cell index than the cell being filled; its OLD state otherwise.  The device reproduces exactly that:
// pos[] holds the position of every oct of the region in the list, a thread follows the chain of reference cells while
// they are NEW by that rule, reads the state the region had on entry at the end of the chain and applies the boundary
// rule once per link.
struct BndArgs {
  double *uold, *tmp;
  const int *son, *nbor, *list;
  int *pos;
  int n, nvar, type, no_inflow, nvector;
  long ncell, ncoarse, ngridmax;
  double smallr;
};
__global__ __launch_bounds__(256) void bnd_mark_kernel(BndArgs A, int clear) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < A.n; i += gridDim.x * blockDim.x) A.pos[A.list[i] - 1] = clear ? 0 : i + 1;
}
__global__ __launch_bounds__(256) void bnd_compute_kernel(BndArgs A) {
  const int dir = A.type % 10, kind = A.type / 10;          // boundary_dir 1..6; 0 reflexive, 1 free
  const int axis = (dir - 1) >> 1, high = (dir - 1) & 1;    // the wall's normal; low (x < 0 side) or high wall
  const int inbor = high ? 2 * axis + 1 : 2 * axis + 2;     // :59-64: towards the box
  const int bit = 1 << axis;
  const long total = (long)A.n * 8;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    const int ind = (int)(t / A.n), i = (int)(t % A.n);
    int oct = A.list[i], cind = ind, napply = 0, ref = 0, rind = 0;
    for (;;) {
      const int fc = A.nbor[(long)(inbor - 1) * A.ngridmax + oct - 1];
      ref = fc > 0 ? A.son[fc - 1] : 0;
      // ind_ref (:66-87): reflexive walls mirror the octant across the wall, free boundaries take the reference oct's layer next to the wall
      rind = kind == 0 ? (cind ^ bit) : (high ? (cind | bit) : (cind & ~bit));
      napply++;
      if (ref > 0 && napply < 64) {
        const int pr = A.pos[ref - 1];
        if (pr != 0) {
          const int chunk_ref = (pr - 1) / A.nvector, chunk_oct = (A.pos[oct - 1] - 1) / A.nvector;
          if (chunk_ref < chunk_oct || (chunk_ref == chunk_oct && rind < cind)) { oct = ref; cind = rind; continue; }
        }
      }
      break;
    }
    const long cref = A.ncoarse + (long)rind * A.ngridmax + ref;          // 1-based, as the reference computes it
    double uu[8];
    for (int v = 0; v < A.nvar; v++) uu[v] = A.uold[(long)v * A.ncell + cref - 1];
    for (int a = 0; a < napply; a++) {
      if (kind == 0) {
        uu[1 + axis] = uu[1 + axis] * -1.0;
      } else {
        double ekin = 0.0;
        double d = __builtin_fmax(uu[0], A.smallr);
        for (int k = 0; k < 3; k++) { const double vel = uu[1 + k] / d; ekin = ekin + 0.5 * d * (vel * vel); }
        uu[4] = uu[4] - ekin;
        if (A.no_inflow) uu[1 + axis] = high ? __builtin_fmax(0.0, uu[1 + axis]) : __builtin_fmin(0.0, uu[1 + axis]);
        ekin = 0.0;
        d = __builtin_fmax(uu[0], A.smallr);
        for (int k = 0; k < 3; k++) { const double vel = uu[1 + k] / d; ekin = ekin + 0.5 * d * (vel * vel); }
        uu[4] = uu[4] + ekin;
      }
    }
    for (int v = 0; v < A.nvar; v++) A.tmp[(long)v * total + t] = uu[v];
  }
}
__global__ __launch_bounds__(256) void bnd_store_kernel(BndArgs A) {
  const long total = (long)A.n * 8;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    const int ind = (int)(t / A.n), i = (int)(t % A.n);
    const long c = A.ncoarse + (long)ind * A.ngridmax + A.list[i] - 1;
    for (int v = 0; v < A.nvar; v++) A.uold[(long)v * A.ncell + c] = A.tmp[(long)v * total + t];
  }
}

using amrlayout::Buf;

struct PinBuf {
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

// MPI: the communicators of one level as build_comm left them (amr/amr_commons.f90:170-179, emission(icpu,l)%igrid and
// reception(icpu,l)%igrid concatenated in icpu order)
struct CommLevel {
  int epoch = -1, ncpu = 0;
  std::vector<int> em_first, rc_first;        // [ncpu+1] positions in the concatenated lists
  Buf em_ig, rc_ig;                           // device indices (translated for the layout `serial`)
  Buf em_raw, rc_raw;                         // the lists as build_comm left them (host indices)
  int serial = -1;
};

// what the dense sweep of a level in tiles needs beyond the cell vectors (see plan_* above), valid for one layout and one list
#ifndef EVENT_ORDER_DEFAULT
#define EVENT_ORDER_DEFAULT 2      // (oct, face): measured on the shell level, profiles/r06_event_order.txt
#endif
#ifndef EVENT_QMINOR_DEFAULT
#define EVENT_QMINOR_DEFAULT 1      // RAMSES_AMD_EVENT_LANES=q: the four fine faces of an event in neighbouring lanes; e: 64 events of one fine face
#endif
struct LevelPlan {
  int version = -1, ngrid = -1, ig_first = 0, ig_last = 0;      // the layout version of the level and the list the plan was made for
  int ig_sample[10] = {0};                                        // (ten entries of it, spread over the list: the fingerprint of the list cache)
  int nghost = 0, nwork = 0, nevent = 0;
  int rows = 0;                                                   // interior rows of its work items (8, or 4 for the 8-row kernels)
  Buf gfather, gslot, gcell, gsten, work, corr, corr_tgt, evt_of, flag, events;
  void release() {
    for (Buf *b : {&gfather, &gslot, &gcell, &gsten, &work, &corr, &corr_tgt, &evt_of, &flag, &events}) { if (b->p) (void)hipFree(b->p); b->p = nullptr; b->cap = 0; }
    version = -1;
  }
};

struct AmrRes {
  std::vector<CommLevel> comm;                 // by level
  Buf sendbuf, recvbuf;
  PinBuf h_send, h_recv;
  std::vector<int64_t> f_send_off, f_recv_off; // [ncpu+1], doubles, of the exchange that is under way
  int halo_level = 0, halo_dir = -1;           // the exchange halo_stage_out has opened (0: none)
  bool valid = false;
  int nvar = 0;
  long ncell = 0, ncoarse = 0, ngridmax = 0;   // of the DEVICE's cell vectors (ngridmax = map.ngd)
  long ngh = 0, ncell_h = 0;                   // of the host's
  amrlayout::DevMap map;
  Buf stat, octpos, bad;                       // status byte per device cell; device oct -> position in the list of its level's plan; bad-index counter
  // the oct lists of the levels as they last arrived, translated: a level's list (active(ilevel)%igrid) comes down with every
  // routine of a step and changes only with the tree
  struct ListSlot { const int *h = nullptr; int n = 0, serial = -1; int sample[10] = {0}; Buf dev, sorted, tmp; bool has_sorted = false; long stamp = 0; };
  ListSlot *cur_slot = nullptr;
  ListSlot lcache[16];
  long lstamp = 0;
  int *cur_ig = nullptr;                       // the list of the routine under way (device indices)
  std::vector<LevelPlan> plan;
  long tile_sweeps = 0, tree_sweeps = 0;
  int64_t f_up_bytes = 0, f_down_bytes = 0;    // bytes of the acceleration that crossed PCIe (ramses_amd_amrres_f_traffic)
  int64_t rho_down_bytes = 0;                  // bytes of the density deposit that went back to the host vector
  bool rho_keep = false;                       // ramses_amd_amrres_rho_keep: rho_fine's deposit stays on the device
  long relayouts = 0;                          // regrids that had to lay the kept levels out again (tiles in the way of the finer levels)
  int err_pending = 0;                         // a tree-walking sweep has run since R.err was last read (the finest level it swept)
  bool announced = false;
  const double *h_uold = nullptr;
  Buf uold, unew, son, nbor, father, work, err, red, okbuf, pack;
  Buf xg;                // xg(1:ngridmax,1:3) (rho_fine's deposit needs the oct centres); sent with the tree when gravity is on
  bool xg_valid = false;
  Buf mp, rho, posof, mpscratch, lists;   // rho_fine: multipoles (4, ncell), the deposit (ncell), oct -> list position, scan scratch
  Buf hkeys, hvals;                       // rho_fine with several ranks: the own octs of the level by position
  long covered_sweeps = 0;                     // how many sweeps of fully refined levels took the dense path (tests, ramses_amd_amrres_covered_sweeps)
  int rl_level = 0, rl_nown = 0, rl_nall = 0;   // the level ramses_amd_amrres_rho_mpi_multipole opened (its list is in `lists`)
  Buf f;                 // f(1:ncell,1:3), a copy of the host array refreshed after force_fine and after regrids
  bool grav = false;
  Buf bnd_list, bnd_pos, bnd_tmp;   // make_boundary_hydro: the regions' oct lists, oct -> position in its region, the new states
  bool bnd_pos_clean = false;
  Buf divu, enew;        // pressure_fix: the reference's divu / enew work vectors (device only: scratch of one step)
  bool pfix = false;
  std::vector<double> hpack;
};
AmrRes g_ar;

inline int grid_for(long n) {
  long g = (n + 255) / 256;
  if (g < 1) g = 1;
  if (g > 8192) g = 8192;
  return (int)g;
}

// RAMSES_AMD_EVENT_ORDER (A/B of the order the surface pass visits the (oct, face) events in): face | tile | oct
int event_order() {
  const char *e = getenv("RAMSES_AMD_EVENT_ORDER");
  if (!e) return EVENT_ORDER_DEFAULT;
  return e[0] == 'f' ? 0 : (e[0] == 't' ? 1 : 2);
}
bool env_on(const char *name) {      // (read on every sweep: the A/B tests flip the switches inside one process)
  const char *e = getenv(name);
  return !(e && e[0] == '0');
}
// an oct list of the host on the device, in device indices (amr_layout.hpp); an index that is not in the tree is an error of
// the caller, reported by the next routine that synchronises anyway (check_lists)
int upload_list(AmrRes &R, Buf &dst, const int *h, int n) {
  HCHK(dst.ensure(sizeof(int) * (size_t)(n > 0 ? n : 1)), "hipMalloc oct list");
  if (n <= 0) return 0;
  HCHK(hipMemcpyAsync(dst.p, h, sizeof(int) * (size_t)n, hipMemcpyHostToDevice, nullptr), "H2D oct list");
  if (R.map.on) {
    hipLaunchKernelGGL(amrlayout::xlate_list_kernel, dim3(amrlayout::grid1(n)), dim3(256), 0, nullptr, dst.as<int>(), n, R.map.perm.as<int>(), R.ngh, R.bad.as<int>());
    HCHK(hipGetLastError(), "oct list translation");
  }
  return 0;
}
int check_lists(AmrRes &R, const char *where) {
  // (the tree-walking sweeps count the father cells they did not find into R.err and do not wait for the answer: a sweep of a
  //  1000-oct level is a handful of launches, a blocking copy per call would double it.  Whoever reads something back anyway --
  //  courant_fine, hydro_flag, a level going home -- asks for both counters.)
  if (R.err_pending) {
    int miss = 0;
    HCHK(hipMemcpy(&miss, R.err.p, sizeof(int), hipMemcpyDeviceToHost), "D2H");
    const int lev = R.err_pending;
    R.err_pending = 0;
    if (miss) {
      HCHK(hipMemset(R.err.p, 0, sizeof(int)), "memset");
      return failf(RAMSES_AMD_EINVAL, "%s: %d father cells needed by an oct do not exist (tree inconsistent; godunov_fine up to level %d since the last check)", where, miss, lev);
    }
  }
  if (!R.map.on) return 0;
  int bad = 0;
  HCHK(hipMemcpy(&bad, R.bad.p, sizeof(int), hipMemcpyDeviceToHost), "D2H");
  if (bad) {
    HCHK(hipMemset(R.bad.p, 0, sizeof(int)), "memset");
    return failf(RAMSES_AMD_EINVAL, "%s: %d octs of a list are not in the tree the device holds", where, bad);
  }
  return 0;
}
// The list of the routine under way in ascending DEVICE order, for the kernels whose result does not depend on the order of
// the list (copies, source terms, restriction): neighbouring threads then touch neighbouring octs of a tile -- the host's
// order is the order of creation, which the device numbering scatters.
int sorted_list(AmrRes &R, LvlArgs &A) {
  AmrRes::ListSlot *S = R.cur_slot;
  if (!R.map.on || !S || S->n < 2 || !env_on("RAMSES_AMD_SORTED_LISTS")) return 0;
  if (!S->has_sorted) {
    HCHK(S->sorted.ensure(sizeof(int) * (size_t)S->n), "hipMalloc");
    size_t bytes = 0;
    HCHK(hipcub::DeviceRadixSort::SortKeys(nullptr, bytes, S->dev.as<int>(), S->sorted.as<int>(), S->n, 0, 32, (hipStream_t) nullptr), "sort");
    HCHK(S->tmp.ensure(bytes), "hipMalloc");
    HCHK(hipcub::DeviceRadixSort::SortKeys(S->tmp.p, bytes, S->dev.as<int>(), S->sorted.as<int>(), S->n, 0, 32, (hipStream_t) nullptr), "sort");
    S->has_sorted = true;
  }
  A.igrid = S->sorted.as<int>();
  return 0;
}
int set_level(AmrRes &R, int ngrid, const int *igrid, LvlArgs &A) {
  if (!R.valid) return failf(RAMSES_AMD_EINVAL, "no resident AMR state (ramses_amd_amrres_load)");
  if (ngrid < 0 || (ngrid > 0 && !igrid)) return failf(RAMSES_AMD_EINVAL, "bad oct list");
  {
    // the same list as before (same array, length, layout; first, last and eight entries in between equal): already there.
    // (Contract of the C ABI: an oct list does not change between two calls of ramses_amd_amrres_tree other than as a whole.)
    AmrRes::ListSlot *hit = nullptr, *lru = &R.lcache[0];
    int sample[10] = {0};
    if (ngrid > 0) for (int k = 0; k < 10; k++) sample[k] = igrid[(long)k * (ngrid - 1) / 9];
    for (AmrRes::ListSlot &S : R.lcache) {
      if (S.h == igrid && S.n == ngrid && S.serial == R.map.serial && ngrid > 0 && memcmp(S.sample, sample, sizeof(sample)) == 0) { hit = &S; break; }
      if (S.stamp < lru->stamp) lru = &S;
    }
    if (!hit || !env_on("RAMSES_AMD_LIST_CACHE")) {
      hit = lru;
      hit->serial = -1;
      if (int rc = upload_list(R, hit->dev, igrid, ngrid)) return rc;
      hit->h = igrid; hit->n = ngrid; hit->serial = R.map.serial; memcpy(hit->sample, sample, sizeof(sample));
      hit->has_sorted = false;
    }
    hit->stamp = ++R.lstamp;
    R.cur_ig = hit->dev.as<int>();
    R.cur_slot = hit;
  }
  A.uold = R.uold.as<double>(); A.unew = R.unew.as<double>();
  A.son = R.son.as<int>(); A.nbor = R.nbor.as<int>(); A.igrid = R.cur_ig;
  A.ngrid = ngrid; A.nvar = R.nvar; A.ncell = R.ncell; A.ncoarse = R.ncoarse; A.ngridmax = R.ngridmax;
  return 0;
}
}  // namespace

static HydroConst make_const_amr(const ramses_amd_hydro_params *p) {
  HydroConst P;
  P.gamma = p->gamma; P.smallr = p->smallr; P.smallc = p->smallc;
  P.smallc2 = p->smallc * p->smallc;
  P.smallp = P.smallc2 / p->gamma;
  P.smalle = P.smallc2 / p->gamma / (p->gamma - 1.0);
  P.entho = 1.0 / (p->gamma - 1.0);
  P.gm1 = p->gamma - 1.0;
  P.gamma6 = (p->gamma + 1.0) / (2.0 * p->gamma);
  P.smallpp = p->smallr * P.smallp;
  P.oneovergamma = 1.0 / p->gamma;
  P.slope_theta = p->slope_theta;
  P.niter_riemann = p->niter_riemann;
  return P;
}

extern "C" {

extern "C" int ramses_amd_mgdist_traffic(int64_t *out2);
// RAMSES_AMD_STATS=1: one line at exit with how godunov_fine of the AMR levels ran (tests and timing scripts read it)
static void amrres_report(void) {
  const char *e = getenv("RAMSES_AMD_STATS");
  if (!e || e[0] == '0') return;
  if (g_ar.f_up_bytes + g_ar.f_down_bytes > 0)
    fprintf(stdout, " ramses_amd: acceleration f over PCIe: %lld bytes to the device, %lld bytes back\n", (long long)g_ar.f_up_bytes, (long long)g_ar.f_down_bytes);
  {
    int64_t t[2] = {0, 0};
    if (ramses_amd_mgdist_traffic(t) == 0 && t[0] + t[1] > 0)
      fprintf(stdout, " ramses_amd: distributed multigrid over PCIe: rho %lld bytes to the device, phi %lld bytes back\n", (long long)t[0], (long long)t[1]);
  }
  if (g_ar.rho_down_bytes > 0) fprintf(stdout, " ramses_amd: density deposit rho over PCIe: %lld bytes back\n", (long long)g_ar.rho_down_bytes);
  if (g_ar.tile_sweeps + g_ar.tree_sweeps == 0) { fflush(stdout); return; }
  fprintf(stdout, " ramses_amd: godunov_fine of AMR levels: %ld sweeps through the dense kernel on tiles (%ld of them fully refined levels), %ld through the tree-walking kernel; %ld levels in tiles at the end\n",
          g_ar.tile_sweeps, g_ar.covered_sweeps, g_ar.tree_sweeps, g_ar.map.on ? g_ar.map.tiles_levels : 0L);
  fflush(stdout);
}

int ramses_amd_amrres_active(void) { return g_ar.valid ? 1 : 0; }

// the tree arrays again (after refine_fine): son(1:ncell), nbor(1:ngridmax,1:6), father(1:ngridmax) of the host.  The levels
// whose octs changed (and every finer one) are laid out again in device numbers; the others keep theirs, and their data.
int ramses_amd_amrres_tree(const int *son, const int *nbor, const int *father) {
  AmrRes &R = g_ar;
  if (!R.valid) return failf(RAMSES_AMD_EINVAL, "no resident AMR state");
  if (!son || !nbor || !father) return failf(RAMSES_AMD_EINVAL, "NULL argument");
  hipError_t e = R.map.build(son, nbor, father, R.son.as<int>(), R.nbor.as<int>(), R.father.as<int>(), R.stat.as<unsigned char>(), nullptr);
  if (e == hipErrorInvalidValue && R.map.on && R.map.overflow) {
    // A legitimate regrid: the finer levels grew, and the free tile slots of the levels that kept their layout are in the way
    // (their fit was checked against the finer levels of the time they were laid out).  Those levels' state lives on the device
    // only: it is parked in the host's numbering, every level is laid out again -- tiles where they fit NOW -- and the state
    // goes back to the new indices.  The rebuilt levels are the caller's to reload, as after any regrid.
    (void)hipGetLastError();
    const int kept = R.map.first_changed - 1;
    const int nvec = 2 * R.nvar + (R.grav ? 3 : 0);
    Buf park;
    HCHK(park.ensure(sizeof(double) * (size_t)nvec * (size_t)R.ncell_h), "hipMalloc (parking the kept levels)");
    auto vec_of = [&](int v, double *&dev) {          // device vector v of the parked set
      if (v < R.nvar) dev = R.uold.as<double>() + (long)v * R.ncell;
      else if (v < 2 * R.nvar) dev = R.unew.as<double>() + (long)(v - R.nvar) * R.ncell;
      else dev = R.f.as<double>() + (long)(v - 2 * R.nvar) * R.ncell;
    };
    for (int l = 1; l <= kept; l++) {
      amrlayout::LevelMap &L = R.map.lev[l];
      for (int v = 0; v < nvec; v++) {
        double *dev; vec_of(v, dev);
        hipLaunchKernelGGL(amrlayout::move_var_kernel<false>, dim3(amrlayout::grid1((long)L.n * 8)), dim3(256), 0, nullptr, L.hoct.as<int>(), L.doct.as<int>(), L.n,
                           R.ncoarse, R.ngh, R.ngridmax, park.as<double>() + (long)v * R.ncell_h, dev);
      }
    }
    HCHK(hipGetLastError(), "parking the kept levels");
    HCHK(hipDeviceSynchronize(), "sync");
    R.map.forget();
    e = R.map.build(son, nbor, father, R.son.as<int>(), R.nbor.as<int>(), R.father.as<int>(), R.stat.as<unsigned char>(), nullptr);
    if (e == hipSuccess) {
      for (int l = 1; l <= kept && l <= R.map.nlev; l++) {
        amrlayout::LevelMap &L = R.map.lev[l];
        for (int v = 0; v < nvec; v++) {
          double *dev; vec_of(v, dev);
          hipLaunchKernelGGL(amrlayout::move_var_kernel<true>, dim3(amrlayout::grid1((long)L.n * 8)), dim3(256), 0, nullptr, L.hoct.as<int>(), L.doct.as<int>(), L.n,
                             R.ncoarse, R.ngh, R.ngridmax, park.as<double>() + (long)v * R.ncell_h, dev);
        }
      }
      HCHK(hipGetLastError(), "restoring the kept levels");
      HCHK(hipDeviceSynchronize(), "sync");
      R.relayouts++;
    }
    if (park.p) (void)hipFree(park.p);
  }
  if (e != hipSuccess) return failf(e == hipErrorInvalidValue ? RAMSES_AMD_EINVAL : RAMSES_AMD_EHIP, "tree layout on the device: %s", e == hipErrorInvalidValue ? R.map.why_not : hipGetErrorString(e));
  if (R.map.on) {
    // (list positions of the levels that were laid out again; the kept levels keep their plans)
    const long k = R.map.kept_end;
    HCHK(hipMemsetAsync(R.octpos.as<int>() + (k - 1), 0xff, sizeof(int) * (size_t)(R.ngridmax - k + 1), nullptr), "memset");
    R.xg_valid = false;                    // the oct centres are indexed by device oct: rho_fine's shim sends them with every tree epoch
    if (!R.announced) {
      R.announced = true;
      if (getenv("RAMSES_AMD_VERBOSE")) fprintf(stderr, "ramses_amd: %d levels on the device, %ld of them in tiles of 32x4x4 octs\n", R.map.nlev, R.map.tiles_levels);
    }
  }
  R.bnd_pos_clean = false;
  return 0;
}

// everything: the hydro state uold(1:ncell,1:nvar) and the tree
int ramses_amd_amrres_load(int nvar, int64_t ngridmax, int64_t ncoarse, const double *uold, const int *son, const int *nbor,
                           const int *father) {
  if (!uold || !son || !nbor || !father) return failf(RAMSES_AMD_EINVAL, "NULL argument");
  if (nvar < 5 || nvar > 7 || ngridmax < 1 || ncoarse < 1) return failf(RAMSES_AMD_EUNSUPPORTED, "AMR residency implements NVAR=5..7");
  AmrRes &R = g_ar;
  { static bool registered = false; if (!registered) { registered = true; atexit(amrres_report); } }
  R.valid = false; R.grav = false; R.pfix = false;
  R.xg_valid = false;          // the oct centres belong to the tree that is loaded below: rho_fine's shim sends them again
  R.nvar = nvar; R.ngh = ngridmax; R.ncoarse = ncoarse; R.ncell_h = ncoarse + 8 * ngridmax;
  // the device's index space: the host's ngridmax times RAMSES_AMD_DEVICE_OCTS (default 1; tiles that are not full cost indices)
  double fac = 1.0;
  if (const char *e = getenv("RAMSES_AMD_DEVICE_OCTS")) { fac = atof(e); if (!(fac >= 1.0 && fac <= 16.0)) fac = 1.0; }
  long ngd = (long)((double)ngridmax * fac);
  if ((unsigned long)(ncoarse + 8 * ngd) >= (1ul << 31)) ngd = ngridmax;      // cell indices are 32-bit ints
  R.map.reset(ngridmax, ngd, ncoarse, son[0] > 0);      // (no oct in the coarse cell: nothing to lay out -- the host's numbering)
  R.ngridmax = R.map.ngd; R.ncell = ncoarse + 8 * R.ngridmax;
  for (LevelPlan &P : R.plan) P.release();        // (flux records of a big level are gigabytes)
  R.plan.clear();
  for (CommLevel &L : R.comm) L.epoch = -1;
  R.h_uold = uold;
  const size_t vb = sizeof(double) * (size_t)nvar * (size_t)R.ncell;
  HCHK(R.uold.ensure(vb), "hipMalloc uold"); HCHK(R.unew.ensure(vb), "hipMalloc unew");
  HCHK(R.son.ensure(sizeof(int) * (size_t)R.ncell), "hipMalloc son");
  HCHK(R.nbor.ensure(sizeof(int) * 6 * (size_t)R.ngridmax), "hipMalloc nbor");
  HCHK(R.father.ensure(sizeof(int) * (size_t)R.ngridmax), "hipMalloc father");
  HCHK(R.stat.ensure((size_t)R.ncell), "hipMalloc stat"); HCHK(R.octpos.ensure(sizeof(int) * (size_t)R.ngridmax), "hipMalloc octpos");
  HCHK(R.bad.ensure(sizeof(int)), "hipMalloc"); HCHK(hipMemsetAsync(R.bad.p, 0, sizeof(int), nullptr), "memset");
  HCHK(R.err.ensure(sizeof(int)), "hipMalloc"); HCHK(R.red.ensure(sizeof(double) * (4 + 3 * 2048)), "hipMalloc");
  HCHK(hipMemsetAsync(R.unew.p, 0, vb, nullptr), "memset unew");
  R.valid = true;
  if (int rc = ramses_amd_amrres_tree(son, nbor, father)) { R.valid = false; return rc; }
  if (!R.map.on) {
    HCHK(hipMemcpyAsync(R.uold.p, uold, vb, hipMemcpyHostToDevice, nullptr), "H2D uold");
  } else {
    // variable by variable through a staging vector in the host's numbering: the octs of the tree into their device slots
    HCHK(hipMemsetAsync(R.uold.p, 0, vb, nullptr), "memset uold");
    HCHK(R.work.ensure(sizeof(double) * (size_t)R.ncell_h), "hipMalloc staging");
    for (int v = 0; v < nvar; v++) {
      HCHK(hipMemcpyAsync(R.work.p, uold + (size_t)v * R.ncell_h, sizeof(double) * (size_t)R.ncell_h, hipMemcpyHostToDevice, nullptr), "H2D uold");
      HCHK(hipMemcpyAsync(R.uold.as<double>() + (size_t)v * R.ncell, R.work.p, sizeof(double) * (size_t)ncoarse, hipMemcpyDeviceToDevice, nullptr), "coarse cells");
      for (int l = 1; l <= R.map.nlev; l++) {
        amrlayout::LevelMap &L = R.map.lev[l];
        hipLaunchKernelGGL(amrlayout::move_var_kernel<true>, dim3(amrlayout::grid1((long)L.n * 8)), dim3(256), 0, nullptr, L.hoct.as<int>(), L.doct.as<int>(), L.n,
                           R.ncoarse, R.ngh, R.ngridmax, R.work.as<double>(), R.uold.as<double>() + (size_t)v * R.ncell);
      }
    }
    HCHK(hipGetLastError(), "uold into the device's numbering");
  }
  HCHK(hipStreamSynchronize(nullptr), "sync");
  return 0;
}

int ramses_amd_amrres_invalidate(void) {
  g_ar.valid = false;
  for (LevelPlan &P : g_ar.plan) P.release();
  return 0;
}

// uold of one level's cells back into the host array (before refine_fine reads it)
int ramses_amd_amrres_sync_level(int ngrid, const int *igrid, double *uold) {
  AmrRes &R = g_ar;
  LvlArgs A;
  if (int rc = set_level(R, ngrid, igrid, A)) return rc;
  if (uold != R.h_uold) return failf(RAMSES_AMD_EINVAL, "sync_level: not the array the state was loaded from");
  if (ngrid == 0) return 0;
  const size_t n = (size_t)ngrid * 8 * R.nvar;
  HCHK(R.pack.ensure(sizeof(double) * n), "hipMalloc");
  hipLaunchKernelGGL(lvl_pack_kernel<true>, dim3(grid_for((long)ngrid * 8)), dim3(256), 0, nullptr, A, R.pack.as<double>());
  HCHK(hipGetLastError(), "pack launch");
  R.hpack.resize(n);
  HCHK(hipMemcpy(R.hpack.data(), R.pack.p, sizeof(double) * n, hipMemcpyDeviceToHost), "D2H level");
  const long tot = (long)ngrid * 8;
  for (int v = 0; v < R.nvar; v++)
    for (int ind = 0; ind < 8; ind++) {
      double *dst = uold + (size_t)v * R.ncell_h + R.ncoarse + (size_t)ind * R.ngh - 1;
      const double *src = R.hpack.data() + (size_t)v * tot + (size_t)ind * ngrid;
      for (int i = 0; i < ngrid; i++) dst[igrid[i]] = src[i];
    }
  return check_lists(R, "sync_level");
}

// uold of one level's cells from the host array (after refine_fine rebuilt the level)
int ramses_amd_amrres_load_level(int ngrid, const int *igrid, const double *uold) {
  AmrRes &R = g_ar;
  LvlArgs A;
  if (int rc = set_level(R, ngrid, igrid, A)) return rc;
  if (ngrid == 0) return 0;
  const size_t n = (size_t)ngrid * 8 * R.nvar;
  const long tot = (long)ngrid * 8;
  R.hpack.resize(n);
  for (int v = 0; v < R.nvar; v++)
    for (int ind = 0; ind < 8; ind++) {
      const double *src = uold + (size_t)v * R.ncell_h + R.ncoarse + (size_t)ind * R.ngh - 1;
      double *dst = R.hpack.data() + (size_t)v * tot + (size_t)ind * ngrid;
      for (int i = 0; i < ngrid; i++) dst[i] = src[igrid[i]];
    }
  HCHK(R.pack.ensure(sizeof(double) * n), "hipMalloc");
  HCHK(hipMemcpy(R.pack.p, R.hpack.data(), sizeof(double) * n, hipMemcpyHostToDevice), "H2D level");
  hipLaunchKernelGGL(lvl_pack_kernel<false>, dim3(grid_for(tot)), dim3(256), 0, nullptr, A, R.pack.as<double>());
  HCHK(hipGetLastError(), "unpack launch");
  HCHK(hipStreamSynchronize(nullptr), "sync");
  return 0;
}

// the whole hydro state back (backup_hydro)
int ramses_amd_amrres_sync_all(double *uold) {
  AmrRes &R = g_ar;
  if (!R.valid) return 0;
  if (uold != R.h_uold) return failf(RAMSES_AMD_EINVAL, "sync_all: not the array the state was loaded from");
  if (!R.map.on) {
    HCHK(hipMemcpy(uold, R.uold.p, sizeof(double) * (size_t)R.nvar * (size_t)R.ncell, hipMemcpyDeviceToHost), "D2H uold");
    return 0;
  }
  // the cells of the tree's octs, variable by variable through a staging vector in the host's numbering (cells of octs that
  // are not in the tree keep what the host array holds)
  HCHK(R.work.ensure(sizeof(double) * (size_t)R.ncell_h), "hipMalloc staging");
  for (int v = 0; v < R.nvar; v++) {
    double *hv = uold + (size_t)v * R.ncell_h;
    HCHK(hipMemcpyAsync(R.work.p, hv, sizeof(double) * (size_t)R.ncell_h, hipMemcpyHostToDevice, nullptr), "H2D staging");
    HCHK(hipMemcpyAsync(R.work.p, R.uold.as<double>() + (size_t)v * R.ncell, sizeof(double) * (size_t)R.ncoarse, hipMemcpyDeviceToDevice, nullptr), "coarse cells");
    for (int l = 1; l <= R.map.nlev; l++) {
      amrlayout::LevelMap &L = R.map.lev[l];
      hipLaunchKernelGGL(amrlayout::move_var_kernel<false>, dim3(amrlayout::grid1((long)L.n * 8)), dim3(256), 0, nullptr, L.hoct.as<int>(), L.doct.as<int>(), L.n,
                         R.ncoarse, R.ngh, R.ngridmax, R.work.as<double>(), R.uold.as<double>() + (size_t)v * R.ncell);
    }
    HCHK(hipGetLastError(), "uold into the host's numbering");
    HCHK(hipMemcpy(hv, R.work.p, sizeof(double) * (size_t)R.ncell_h, hipMemcpyDeviceToHost), "D2H uold");
  }
  return 0;
}

int ramses_amd_amrres_set_unew(int ngrid, const int *igrid) {
  LvlArgs A;
  if (int rc = set_level(g_ar, ngrid, igrid, A)) return rc;
  if (ngrid == 0) return 0;
  if (int rc = sorted_list(g_ar, A)) return rc;
  hipLaunchKernelGGL(lvl_copy_kernel, dim3(grid_for((long)ngrid * 8)), dim3(256), 0, nullptr, A, A.unew, A.uold);
  HCHK(hipGetLastError(), "set_unew launch");
  return 0;
}

int ramses_amd_amrres_set_uold(const ramses_amd_hydro_params *p, int ngrid, const int *igrid) {
  if (!p) return failf(RAMSES_AMD_EINVAL, "NULL argument");
  LvlArgs A;
  if (int rc = set_level(g_ar, ngrid, igrid, A)) return rc;
  if (ngrid == 0) return 0;
  if (int rc = sorted_list(g_ar, A)) return rc;
  hipLaunchKernelGGL(lvl_set_uold_kernel, dim3(grid_for((long)ngrid * 8)), dim3(256), 0, nullptr, A, p->smallr);
  HCHK(hipGetLastError(), "set_uold launch");
  return 0;
}

int ramses_amd_amrres_upload_fine(const ramses_amd_hydro_params *p, int ngrid, const int *igrid, int interpol_var) {
  if (!p) return failf(RAMSES_AMD_EINVAL, "NULL argument");
  if (interpol_var < 0 || interpol_var > 2) return failf(RAMSES_AMD_EINVAL, "interpol_var must be 0, 1 or 2");
  LvlArgs A;
  if (int rc = set_level(g_ar, ngrid, igrid, A)) return rc;
  if (ngrid == 0) return 0;
  if (int rc = sorted_list(g_ar, A)) return rc;
  const dim3 g(grid_for((long)ngrid * 8)), b(256);
  switch (A.nvar) {
    case 5: hipLaunchKernelGGL(lvl_upload_kernel<5>, g, b, 0, nullptr, A, interpol_var, p->smallr); break;
    case 6: hipLaunchKernelGGL(lvl_upload_kernel<6>, g, b, 0, nullptr, A, interpol_var, p->smallr); break;
    default: hipLaunchKernelGGL(lvl_upload_kernel<7>, g, b, 0, nullptr, A, interpol_var, p->smallr); break;
  }
  HCHK(hipGetLastError(), "upload_fine launch");
  return 0;
}

// out4 = {dt_loc (min with dt_in), mass_loc, sum(E*vol), eint_loc} over the leaf cells of the level
int ramses_amd_amrres_courant(const ramses_amd_hydro_params *p, int ngrid, const int *igrid, double dx, double dt_in, double *out4) {
  if (!p || !out4) return failf(RAMSES_AMD_EINVAL, "NULL argument");
  LvlArgs A;
  if (int rc = set_level(g_ar, ngrid, igrid, A)) return rc;
  AmrRes &R = g_ar;
  // (dt is a minimum; the diagnostic sums are added per workgroup in workgroup order either way, not in the reference's)
  if (int rc = sorted_list(R, A)) return rc;
  const double dt0 = p->courant_factor * dx / p->smallc;
  hipLaunchKernelGGL(lvl_courant_init_kernel, dim3(1), dim3(1), 0, nullptr, R.red.as<double>(), dt0);
  if (ngrid > 0) {
    int g = grid_for((long)ngrid * 8);
    if (g > 2048) g = 2048;
    if (R.grav) hipLaunchKernelGGL(lvl_courant_kernel<true>, dim3(g), dim3(256), 0, nullptr, A, R.f.as<double>(), make_const_amr(p), dx, dx * dx * dx, p->courant_factor, dt0, R.red.as<double>());
    else hipLaunchKernelGGL(lvl_courant_kernel<false>, dim3(g), dim3(256), 0, nullptr, A, (const double *)nullptr, make_const_amr(p), dx, dx * dx * dx, p->courant_factor, dt0, R.red.as<double>());
    hipLaunchKernelGGL(lvl_courant_final_kernel, dim3(1), dim3(64), 0, nullptr, R.red.as<double>(), g);
  }
  HCHK(hipGetLastError(), "courant launch");
  HCHK(hipMemcpy(out4, R.red.p, sizeof(double) * 4, hipMemcpyDeviceToHost), "D2H courant");
  if (dt_in < out4[0]) out4[0] = dt_in;
  return check_lists(R, "courant_fine");
}

// hydro_flag's gradient criteria: the cells (1-based indices into the cell vectors, ascending) where hydro_refine asks
// for refinement come back as a compact list -- cells[0..*ncells), capacity 8*ngrid -- and nothing else crosses PCIe
int ramses_amd_amrres_hydro_flag(const ramses_amd_hydro_params *p, int ngrid, const int *igrid, double err_grad_d, double err_grad_p,
                                 double err_grad_u, double floor_d, double floor_p, double floor_u, int *cells, int *ncells) {
  if (!p || !cells || !ncells) return failf(RAMSES_AMD_EINVAL, "NULL argument");
  *ncells = 0;
  LvlArgs A;
  if (int rc = set_level(g_ar, ngrid, igrid, A)) return rc;
  if (ngrid == 0) return 0;
  if (A.nvar < 5) return failf(RAMSES_AMD_EUNSUPPORTED, "NVAR");
  AmrRes &R = g_ar;
  if (int rc = sorted_list(R, A)) return rc;          // (the flagged cells come back sorted anyway)
  HCHK(R.okbuf.ensure(sizeof(int) * (8 * (size_t)ngrid + 1)), "hipMalloc");
  int *d_count = R.okbuf.as<int>(), *d_list = d_count + 1;
  HCHK(hipMemsetAsync(d_count, 0, sizeof(int), nullptr), "memset");
  FlagCrit F = {err_grad_d, err_grad_p, err_grad_u, floor_d, floor_p, floor_u, p->gamma, p->smallr};
  hipLaunchKernelGGL(lvl_flag_kernel, dim3(grid_for((long)ngrid * 8)), dim3(256), 0, nullptr, A, F, d_list, d_count);
  HCHK(hipGetLastError(), "hydro_flag launch");
  int n = 0;
  HCHK(hipMemcpy(&n, d_count, sizeof(int), hipMemcpyDeviceToHost), "D2H flag count");
  if (n < 0 || (long)n > 8L * ngrid) return failf(RAMSES_AMD_EHIP, "hydro_flag: bad flag count %d", n);
  if (n > 0) {
    if (R.map.on) hipLaunchKernelGGL(amrlayout::cells_d2h_kernel, dim3(amrlayout::grid1(n)), dim3(256), 0, nullptr, d_list, n, R.map.iperm.as<int>(), R.ncoarse, R.ngh, R.ngridmax);
    HCHK(hipMemcpy(cells, d_list, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost), "D2H flagged cells");
    std::sort(cells, cells + n);
  }
  *ncells = n;
  return check_lists(R, "hydro_flag");
}

namespace {

// the plan of a level in tiles for the list of the call (R.cur_ig; see plan_* above); rebuilt when the layout or the list changed
int build_plan(AmrRes &R, int ilevel, int ngrid, const int *h_igrid, int rows, LevelPlan &P) {
  amrlayout::LevelMap &L = R.map.lev[ilevel];
  hipStream_t s = nullptr;
  P.version = -1;
  PlanArgs A;
  A.son = R.son.as<int>(); A.nbor = R.nbor.as<int>(); A.father = R.father.as<int>(); A.iperm = R.map.iperm.as<int>();
  A.stat = R.stat.as<unsigned char>(); A.octpos = R.octpos.as<int>(); A.ig = R.cur_ig; A.n = ngrid;
  A.ncell = R.ncell; A.ncoarse = R.ncoarse; A.ngd = R.ngridmax;
  A.dir = L.dir.as<int>(); A.tileid = L.tileid.as<int>(); A.base = L.base; A.no = L.no; A.ntx = L.ntx; A.nty = L.nty; A.ntz = L.ntz;
  HCHK(P.gfather.ensure(sizeof(int) * (size_t)L.cap), "hipMalloc");
  const int gcap = (int)std::min<long>(L.cap - L.n, (long)ngrid * 26);
  HCHK(P.gslot.ensure(sizeof(int) * (size_t)(gcap > 0 ? gcap : 1)), "hipMalloc"); HCHK(P.gcell.ensure(sizeof(int) * (size_t)(gcap > 0 ? gcap : 1)), "hipMalloc");
  // per slot of the level's index range and face: the leaf cell of the coarser level behind it, and the event that carries its
  // four flux records (events = the (oct of the list, face) pairs with such a cell, compacted: the records take
  // nevent x 4 x (nvar + 2) doubles -- a level without a surface has none)
  HCHK(P.corr_tgt.ensure(sizeof(int) * (size_t)L.cap * 6), "hipMalloc");
  HCHK(hipMemsetAsync(P.corr_tgt.p, 0, sizeof(int) * (size_t)L.cap * 6, s), "memset");
  HCHK(P.evt_of.ensure(sizeof(int) * (size_t)L.cap * 6), "hipMalloc");
  HCHK(R.okbuf.ensure(sizeof(int) * 2), "hipMalloc");
  int *cnt = R.okbuf.as<int>();
  HCHK(hipMemsetAsync(cnt, 0, sizeof(int) * 2, s), "memset");
  hipLaunchKernelGGL(plan_clear_kernel, dim3(grid_for(L.cap * 8)), dim3(256), 0, s, A.stat, A.octpos, R.ncoarse, R.ngridmax, L.base, L.cap, P.gfather.as<int>());
  hipLaunchKernelGGL(plan_owned_kernel, dim3(grid_for((long)ngrid * 8)), dim3(256), 0, s, A);
  // (a level that holds every oct of the periodic box has no neighbour position without an oct)
  if ((long)L.n < (long)L.no * L.no * L.no)
    hipLaunchKernelGGL(plan_ghost_kernel, dim3(grid_for((long)ngrid * 26)), dim3(256), 0, s, A, P.gfather.as<int>(), P.gslot.as<int>(), P.gcell.as<int>(), cnt, gcap, cnt + 1);
  hipLaunchKernelGGL(plan_target_kernel, dim3(grid_for((long)ngrid * 6)), dim3(256), 0, s, A, P.corr_tgt.as<int>());
  HCHK(P.events.ensure(sizeof(int) * ((size_t)ngrid * 6 + 1)), "hipMalloc");
  HCHK(hipMemsetAsync(P.events.p, 0, sizeof(int), s), "memset");
  hipLaunchKernelGGL(plan_events_kernel, dim3((unsigned)(((long)ngrid * 6 + 255) / 256)), dim3(256), 0, s, A, P.corr_tgt.as<int>(), P.events.as<int>() + 1, P.events.as<int>(), P.evt_of.as<int>());
  // work items: columns of 60 x 8 cells, runs of 8-plane chunks up to 128 planes
  // (rows: interior rows of a work item -- 8, or 4 for the kernels of 8 rows; even: an oct never straddles two)
  const int n = 2 * L.no, wtx = (n + 59) / 60, wty = n / rows, wz = n / 8;
  const size_t nflag = (size_t)wtx * wty * wz;
  HCHK(P.flag.ensure(nflag), "hipMalloc");
  HCHK(hipMemsetAsync(P.flag.p, 0, nflag, s), "memset");
  hipLaunchKernelGGL(plan_work_kernel, dim3(grid_for(ngrid)), dim3(256), 0, s, A, wtx, rows, wz, P.flag.as<unsigned char>());
  HCHK(hipGetLastError(), "plan launch");
  int hc[2] = {0, 0};
  std::vector<unsigned char> flag(nflag);
  HCHK(hipMemcpyAsync(&P.nevent, P.events.p, sizeof(int), hipMemcpyDeviceToHost, s), "D2H");
  HCHK(hipMemcpyAsync(hc, cnt, sizeof(int) * 2, hipMemcpyDeviceToHost, s), "D2H");
  HCHK(hipMemcpyAsync(flag.data(), P.flag.p, nflag, hipMemcpyDeviceToHost, s), "D2H");
  HCHK(hipStreamSynchronize(s), "sync");
  HCHK(P.corr.ensure(sizeof(double) * 4 * (size_t)(R.nvar + 2) * (size_t)(P.nevent > 0 ? P.nevent : 1)), "hipMalloc flux records");
  if (P.nevent > 1) {
    // (once per plan) the events sorted by face and device oct, their index table after the sort
    amrlayout::Buf &k1 = P.corr, &k2 = P.gfather, &v2 = P.flag;        // (free here: the records are written by the first sweep, the ghost table and the flags have done their job)
    HCHK(k1.ensure(sizeof(unsigned long long) * (size_t)P.nevent), "hipMalloc"); HCHK(k2.ensure(sizeof(unsigned long long) * (size_t)P.nevent), "hipMalloc");
    HCHK(v2.ensure(sizeof(int) * (size_t)P.nevent), "hipMalloc");
    hipLaunchKernelGGL(plan_event_keys_kernel, dim3((P.nevent + 255) / 256), dim3(256), 0, s, A, P.events.as<int>() + 1, P.nevent, k1.as<unsigned long long>(), event_order());
    size_t bytes = 0;
    HCHK(hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, k1.as<unsigned long long>(), k2.as<unsigned long long>(), P.events.as<int>() + 1, v2.as<int>(), P.nevent, 0, 35, s), "sort");
    HCHK(R.work.ensure(bytes), "hipMalloc");
    HCHK(hipcub::DeviceRadixSort::SortPairs(R.work.p, bytes, k1.as<unsigned long long>(), k2.as<unsigned long long>(), P.events.as<int>() + 1, v2.as<int>(), P.nevent, 0, 35, s), "sort");
    HCHK(hipMemcpyAsync(P.events.as<int>() + 1, v2.p, sizeof(int) * (size_t)P.nevent, hipMemcpyDeviceToDevice, s), "copy");
    HCHK(hipStreamSynchronize(s), "sync");
    HCHK(P.corr.ensure(sizeof(double) * 4 * (size_t)(R.nvar + 2) * (size_t)P.nevent), "hipMalloc flux records");
  }
  if (P.nevent > 0) {
    hipLaunchKernelGGL(plan_event_index_kernel, dim3((P.nevent + 255) / 256), dim3(256), 0, s, A, P.events.as<int>() + 1, P.nevent, P.evt_of.as<int>());
    HCHK(hipGetLastError(), "event index");
  }
  if (hc[1]) return failf(RAMSES_AMD_EINVAL, "level %d: %d neighbour positions of an oct have no father cell or no tile (tree inconsistent)", ilevel, hc[1]);
  if (hc[0] > gcap) return failf(RAMSES_AMD_EINVAL, "level %d: more ghost octs (%d) than free slots in the level's tiles (%d)", ilevel, hc[0], gcap);
  P.nghost = hc[0];
  if (P.nghost > 1) {
    // the ghost octs in slot order: neighbouring threads of the fill kernel then write neighbouring octs of a tile (the search
    // appended them in whatever order its waves arrived)
    amrlayout::Buf &k2 = P.gfather, &v2 = P.flag;          // (both free from here on: the ghost table's job is done, the flags are on the host)
    HCHK(k2.ensure(sizeof(int) * (size_t)P.nghost), "hipMalloc"); HCHK(v2.ensure(sizeof(int) * (size_t)P.nghost), "hipMalloc");
    size_t bytes = 0;
    HCHK(hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, P.gslot.as<int>(), k2.as<int>(), P.gcell.as<int>(), v2.as<int>(), P.nghost, 0, 32, s), "sort");
    HCHK(R.work.ensure(bytes), "hipMalloc");
    HCHK(hipcub::DeviceRadixSort::SortPairs(R.work.p, bytes, P.gslot.as<int>(), k2.as<int>(), P.gcell.as<int>(), v2.as<int>(), P.nghost, 0, 32, s), "sort");
    HCHK(hipMemcpyAsync(P.gslot.p, k2.p, sizeof(int) * (size_t)P.nghost, hipMemcpyDeviceToDevice, s), "copy");
    HCHK(hipMemcpyAsync(P.gcell.p, v2.p, sizeof(int) * (size_t)P.nghost, hipMemcpyDeviceToDevice, s), "copy");
    HCHK(hipStreamSynchronize(s), "sync");
  }
  if (P.nghost > 0) {
    HCHK(P.gsten.ensure(sizeof(int) * 7 * (size_t)P.nghost), "hipMalloc");
    hipLaunchKernelGGL(plan_ghost_stencil_kernel, dim3(grid_for((long)P.nghost * 7)), dim3(256), 0, s, R.son.as<int>(), R.nbor.as<int>(), P.gcell.as<int>(), P.nghost,
                       R.ncoarse, R.ngridmax, P.gsten.as<int>());
    HCHK(hipGetLastError(), "ghost stencil");
  }
  // Work items = runs of flagged 8-plane chunks of a column, cut to at most `zrun` planes.  A workgroup fills a CU (LDS), so a
  // launch proceeds in rounds of ncu items, each costing its planes + 3 (the pipeline's prologue): take the cut that minimises
  // rounds x (planes + 3) -- long items for big levels (least redundant work), short ones when a level has few columns
  // (a 256^3 level is 320 columns of 128 planes: two rounds of 131 planes with the longest cut, five of 35 with 32)
  static int ncu = 0;
  if (ncu == 0) {
    ncu = 256;
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ncu = v;
  }
  std::vector<int> runs;                       // (tx, ty, first chunk, chunks)
  for (int ty = 0; ty < wty; ty++)
    for (int tx = 0; tx < wtx; tx++) {
      const unsigned char *col = flag.data() + ((size_t)ty * wtx + tx) * wz;
      for (int z = 0; z < wz;) {
        if (!col[z]) { z++; continue; }
        int z1 = z;
        while (z1 < wz && col[z1]) z1++;
        runs.push_back(tx); runs.push_back(ty); runs.push_back(z); runs.push_back(z1 - z);
        z = z1;
      }
    }
  int zrun = 0;
  if (const char *e = getenv("RAMSES_AMD_TILE_ZRUN")) { const int v = atoi(e); if (v >= 8 && v <= 4096) zrun = v / 8 * 8; }
  if (zrun == 0) {
    double best = 0.0;
    for (int cand = 16; cand <= 256; cand *= 2) {
      long nitems = 0;
      for (size_t r = 0; r < runs.size(); r += 4) nitems += (runs[r + 3] * 8 + cand - 1) / cand;
      const double cost = (double)((nitems + ncu - 1) / ncu) * (cand + 3);
      if (zrun == 0 || cost < best) { best = cost; zrun = cand; }
    }
  }
  std::vector<int> items;
  for (size_t r = 0; r < runs.size(); r += 4) {
    const int za = runs[r + 2] * 8, zb = za + runs[r + 3] * 8;
    for (int z = za; z < zb; z += zrun) {
      items.push_back(runs[r] * 60); items.push_back(runs[r + 1] * rows); items.push_back(z); items.push_back(std::min(z + zrun, zb));
    }
  }
  const int nw = (int)(items.size() / 4);
  {
    // the items of one slab of planes next to each other -- x first, then y, then the slabs (RAMSES_AMD_TILE_ORDER=column keeps a
    // column's pieces together: the round-5 order): the 32 workgroups an XCD runs at a time are then neighbours in x AND y
    // marching the same planes, and the four halo rows of a 12-row tile are asked of the L2 their neighbour filled
    const char *e = getenv("RAMSES_AMD_TILE_ORDER");
    if (!(e && e[0] == 'c')) {
      std::vector<int> idx((size_t)nw), sorted(items.size());
      for (int k = 0; k < nw; k++) idx[(size_t)k] = k;
      std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) {
        const int *A = &items[(size_t)a * 4], *B = &items[(size_t)b * 4];
        if (A[2] != B[2]) return A[2] < B[2];
        if (A[1] != B[1]) return A[1] < B[1];
        return A[0] < B[0];
      });
      for (int k = 0; k < nw; k++) for (int q = 0; q < 4; q++) sorted[(size_t)k * 4 + q] = items[(size_t)idx[(size_t)k] * 4 + q];
      items.swap(sorted);
    }
  }
  // workgroup b runs on XCD b mod 8: give each XCD a contiguous run of the list (neighbouring columns re-read each other's
  // halo from ONE L2)
  std::vector<int> order((size_t)nw * 4);
  {
    const int per = (nw + 7) / 8;
    int b = 0;
    for (int k = 0; k < per; k++)
      for (int x = 0; x < 8; x++) {
        const int j = x * per + k;
        if (j < nw) { for (int q = 0; q < 4; q++) order[(size_t)b * 4 + q] = items[(size_t)j * 4 + q]; b++; }
      }
  }
  HCHK(P.work.ensure(sizeof(int) * 4 * (size_t)(nw > 0 ? nw : 1)), "hipMalloc");
  if (nw > 0) HCHK(hipMemcpy(P.work.p, order.data(), sizeof(int) * 4 * (size_t)nw, hipMemcpyHostToDevice), "H2D work list");
  P.nwork = nw;
  P.rows = rows;
  P.ngrid = ngrid; P.ig_first = ngrid > 0 ? h_igrid[0] : 0; P.ig_last = ngrid > 0 ? h_igrid[ngrid - 1] : 0;
  for (int k = 0; k < 10; k++) P.ig_sample[k] = h_igrid[(long)k * (ngrid - 1) / 9];
  P.version = L.version;
  return 0;
}

// returns 0 and sets done when the level took the dense path; done = false: the caller walks the tree
int tile_level_sweep(AmrRes &R, const ramses_amd_hydro_params *p, int ilevel, int ngrid, const int *h_igrid, double dx, double dt, int nvector,
                     int interpol_var, int interpol_type, bool &done) {
  done = false;
  if (!R.map.on || ilevel < 3 || ilevel > R.map.nlev || R.map.lev[ilevel].layout != 1) return 0;
  amrlayout::LevelMap &L = R.map.lev[ilevel];
  const bool covered = (long)L.n == (long)L.no * L.no * L.no && ngrid == L.n;
  if (!env_on(covered ? "RAMSES_AMD_COVERED_DENSE" : "RAMSES_AMD_TILE_DENSE") || !env_on("RAMSES_AMD_TILE_SWEEP")) return 0;
  {
    // a small level is quicker through the tree: the dense sweep is a pipeline of >= 19 plane iterations per workgroup whatever
    // the level holds (measured on MI355X: equal at 32768 octs = a 64^3 level, 1.6x quicker at 262144; profiles/r05_tile_crossover.txt; RAMSES_AMD_TILE_MIN_OCTS overrides, 0: always dense)
    long min_octs = 32768;
    if (const char *e = getenv("RAMSES_AMD_TILE_MIN_OCTS")) { const long v = atol(e); if (v >= 0) min_octs = v; }
    if (ngrid < min_octs) return 0;
  }
  // (round 6: runs with one or two passive scalars and the Newton solver too -- kernels of 8 rows, work items of 4)
  const int nvar = R.nvar;
  if (nvar < 5 || nvar > 7 || p->nvar != nvar || p->ndim != 3 || p->difmag > 0.0 || R.pfix) return 0;
  if (p->scheme != RAMSES_AMD_SCHEME_MUSCL && !(p->scheme == RAMSES_AMD_SCHEME_PLMDE && nvar == 5)) return 0;
  const int st = p->slope_type;
  if (!(st == 0 || st == 1 || st == 2 || st == 3 || st == 7 || st == 8)) return 0;
  const int rows = strictmode::tile_sweep_rows(p->riemann, nvar, st, p->scheme);
  if (interpol_var < 0 || interpol_var > 2 || interpol_type < 0 || interpol_type > 4) return 0;
  if ((unsigned long)R.ncell * 8ul >= (1ul << 31)) {      // lane offsets into a cell vector are 31-bit byte offsets
    static bool told = false;
    if (!told) { told = true; fprintf(stderr, "ramses_amd: the device's cell vectors hold %ld cells (>= 2^28): the levels keep the tree-walking sweep\n", R.ncell); }
    return 0;
  }
  hipStream_t s = nullptr;
  if ((size_t)ilevel >= R.plan.size()) R.plan.resize((size_t)ilevel + 1);
  LevelPlan &P = R.plan[ilevel];
  // (the plan of a level that kept its layout survives a regrid of the finer levels: its octs, ghosts and work items are the same.
  //  The list of a level must not change between two regrids without its first or last entry or its length changing: it is
  //  active(ilevel)%igrid, which only build_comm / refine_fine / load_balance rewrite.)
  bool same_list = P.version == L.version && P.rows == rows && P.ngrid == ngrid && P.ig_first == h_igrid[0] && P.ig_last == h_igrid[ngrid - 1];
  for (int k = 0; same_list && k < 10; k++) same_list = P.ig_sample[k] == h_igrid[(long)k * (ngrid - 1) / 9];
  if (!same_list)
    if (int rc = build_plan(R, ilevel, ngrid, h_igrid, rows, P)) return rc;
  if (P.nwork == 0) { done = true; return 0; }
  if (P.nghost > 0) {
    const dim3 gg((P.nghost + 127) / 128), gb(128);
    double *gf = R.grav ? R.f.as<double>() : nullptr;
    if (nvar == 5) hipLaunchKernelGGL(plan_ghost_fill_kernel<5>, gg, gb, 0, s, R.uold.as<double>(), gf, P.gslot.as<int>(), P.gsten.as<int>(), P.nghost, R.ncell, R.ncoarse, R.ngridmax, interpol_var, interpol_type, p->smallr);
    else if (nvar == 6) hipLaunchKernelGGL(plan_ghost_fill_kernel<6>, gg, gb, 0, s, R.uold.as<double>(), gf, P.gslot.as<int>(), P.gsten.as<int>(), P.nghost, R.ncell, R.ncoarse, R.ngridmax, interpol_var, interpol_type, p->smallr);
    else hipLaunchKernelGGL(plan_ghost_fill_kernel<7>, gg, gb, 0, s, R.uold.as<double>(), gf, P.gslot.as<int>(), P.gsten.as<int>(), P.nghost, R.ncell, R.ncoarse, R.ngridmax, interpol_var, interpol_type, p->smallr);
    HCHK(hipGetLastError(), "ghost octs");
  }
  SweepArgs A;
  A.uold = R.uold.as<double>(); A.unew = R.unew.as<double>(); A.grav = R.grav ? R.f.as<double>() : nullptr;      // the cell vectors themselves
  A.stat = R.stat.as<unsigned char>(); A.dir = L.dir.as<int>(); A.work = P.work.as<int>(); A.nwork = P.nwork;
  A.ntx = L.ntx; A.nty = L.nty; A.ntz = L.ntz; A.ngd = R.ngridmax; A.ncoarse = R.ncoarse;
  // (the finest level of the tree: nothing has touched unew since set_unew copied uold into it -- the contract of this routine,
  //  hydro/godunov_fine.f90:5-35 after amr_step's set_unew -- so the kernel re-reads uold from L2 instead of streaming unew)
  A.base_uold = (ilevel >= R.map.nlev || R.map.lev[ilevel + 1].n == 0) ? 1 : 0;
  const int n = 2 * L.no;
  A.nx = A.ny = A.nz = n; A.ng = 0;
  A.pitch_y = n; A.pitch_z = (long)n * n; A.pitch_var = R.ncell;
  A.zchunk = n < 128 ? n : 128;
  A.region = SWEEP_ALL;
  A.dt = dt; A.dx = dx; A.rdx = 1.0 / dx;
  int ex = 0;
  const double m = std::frexp(dx, &ex);
  A.pow2 = (m == 0.5) ? 1 : 0;
  A.P = make_const_amr(p);
  // the surface pass first (it reads uold and the ghost cells only): the fluxes owed to the coarser level, one record per
  // (event, fine face).  Arithmetic: strict (bit-identical) unless the caller's parameters ask for the fast build (fast_math: the
  // patched program's default, certified <= 1e-12 against the reference program on an AMR run with sub-cycling and regrids,
  // tests/test_fast_certificate_gpu.py; RAMSES_AMD_STRICT=1 selects the bit-identical build)
  const bool fast = p->fast_math != 0;
  if (P.nevent > 0) {
    SurfArgs S;
    S.uold = A.uold; S.grav = A.grav; S.stat = A.stat; S.dir = A.dir; S.tileid = L.tileid.as<int>();
    S.events = P.events.as<int>() + 1; S.ig = R.cur_ig; S.rec = P.corr.as<double>(); S.nevent = P.nevent;
    S.base = L.base; S.ncoarse = R.ncoarse; S.ngd = R.ngridmax; S.ncell = R.ncell;
    S.no = L.no; S.ntx = L.ntx; S.nty = L.nty; S.ntz = L.ntz;
    S.dt = A.dt; S.dx = A.dx; S.rdx = A.rdx; S.pow2 = A.pow2; S.P = A.P;
    { const char *e = getenv("RAMSES_AMD_EVENT_LANES"); S.qminor = e ? (e[0] == 'q' ? 1 : 0) : EVENT_QMINOR_DEFAULT; }
    // (the pass on a stream of its own beside the marching kernel was measured in round 6 -- 2.62 -> 2.54 ms strict, 2.09 -> 2.08 fast
    //  on the shell level: a CU the marching kernel fills has no registers left for it, the two take turns; dropped)
    hipError_t es = fast ? fastmode::launch_surface_flux(S, st, p->riemann, nvar, p->scheme, R.grav, s) : strictmode::launch_surface_flux(S, st, p->riemann, nvar, p->scheme, R.grav, s);
    if (es == hipErrorInvalidValue) { (void)hipGetLastError(); return 0; }     // a variant the tile kernels do not cover
    HCHK(es, "surface pass of a level in tiles");
  }
  hipError_t e = fast ? fastmode::launch_godunov_sweep(A, st, p->riemann, rows + 4, p->scheme, nvar, R.grav, s)
                      : strictmode::launch_godunov_sweep(A, st, p->riemann, rows + 4, p->scheme, nvar, R.grav, s);
  if (e == hipErrorInvalidValue) { (void)hipGetLastError(); return 0; }     // a variant the tile kernels do not cover
  HCHK(e, "dense sweep of a level in tiles");
  // what the level owes to the leaf cells of the coarser one, replayed in the reference's order
  {
    PlanArgs Q;
    Q.son = R.son.as<int>(); Q.nbor = R.nbor.as<int>(); Q.father = R.father.as<int>(); Q.iperm = R.map.iperm.as<int>();
    Q.stat = R.stat.as<unsigned char>(); Q.octpos = R.octpos.as<int>(); Q.ig = R.cur_ig; Q.n = ngrid;
    Q.ncell = R.ncell; Q.ncoarse = R.ncoarse; Q.ngd = R.ngridmax;
    Q.dir = L.dir.as<int>(); Q.tileid = L.tileid.as<int>(); Q.base = L.base; Q.no = L.no; Q.ntx = L.ntx; Q.nty = L.nty; Q.ntz = L.ntz;
    if (P.nevent > 0) {
      hipLaunchKernelGGL(tile_coarse_update_kernel, dim3((P.nevent + 255) / 256), dim3(256), 0, s, Q, R.unew.as<double>(), P.corr.as<double>(), P.corr_tgt.as<int>(),
                         P.evt_of.as<int>(), P.events.as<int>() + 1, P.nevent, nvector, nvar);
      HCHK(hipGetLastError(), "coarse corrections");
    }
  }
  if (covered) R.covered_sweeps++;
  R.tile_sweeps++;
  done = true;
  return 0;
}
}  // namespace

extern "C" int64_t ramses_amd_amrres_covered_sweeps(void) { return g_ar.covered_sweeps; }
// sweeps of AMR levels so far: through the dense kernel on tiles / through the tree-walking kernel
extern "C" int64_t ramses_amd_amrres_tile_sweeps(void) { return g_ar.tile_sweeps; }
extern "C" int64_t ramses_amd_amrres_tree_sweeps(void) { return g_ar.tree_sweeps; }
// regrids after which the levels that had kept their layout were laid out again, their state moved on the device
extern "C" int64_t ramses_amd_amrres_relayouts(void) { return g_ar.relayouts; }
// the coarsest level the last ramses_amd_amrres_tree / _load numbered again (its octs and those of every finer level have new device
// indices: the caller reloads them -- ramses_amd_amrres_load_level, _load_f -- before anything reads them; the levels below keep
// indices and data); nlev + 1: none.  0: the host's numbering is in force (every call keeps every index).
extern "C" int ramses_amd_amrres_first_changed(void) { return g_ar.valid && g_ar.map.on ? g_ar.map.first_changed : 0; }
// levels the device stores in tiles (0: the host's numbering is in force)
extern "C" int ramses_amd_amrres_tiled_levels(void) { return g_ar.valid && g_ar.map.on ? (int)g_ar.map.tiles_levels : 0; }

// godunov_fine(ilevel) on the resident arrays
int ramses_amd_amrres_godunov(const ramses_amd_hydro_params *p, int ilevel, int ngrid, const int *igrid, double dx, double dt,
                              int nvector, int interpol_var, int interpol_type) {
  if (!p) return failf(RAMSES_AMD_EINVAL, "NULL argument");
  LvlArgs A;
  if (int rc = set_level(g_ar, ngrid, igrid, A)) return rc;
  if (ngrid == 0) return 0;
  AmrRes &R = g_ar;
  {
    bool done = false;
    if (int rc = tile_level_sweep(R, p, ilevel, ngrid, igrid, dx, dt, nvector, interpol_var, interpol_type, done)) return rc;
    if (done) return 0;          // (asynchronous: a bad oct list is reported by the next routine that reads something back)
  }
  R.tree_sweeps++;
  const int64_t nw = ramses_amd_godunov_fine_amr_workspace(ngrid, R.ngridmax);
  if (nw < 0) return (int)nw;
  HCHK(R.work.ensure((size_t)nw), "hipMalloc work");
  if (!R.err_pending) HCHK(hipMemsetAsync(R.err.p, 0, sizeof(int), nullptr), "memset");
  if (int rc = ramses_amd_godunov_fine_amr_device(p, ilevel, ngrid, R.cur_ig, R.son.as<int>(), R.nbor.as<int>(), R.father.as<int>(),
                                                  R.ngridmax, R.ncoarse, R.uold.as<double>(), R.unew.as<double>(), R.grav ? R.f.as<double>() : nullptr, R.pfix ? R.divu.as<double>() : nullptr, R.pfix ? R.enew.as<double>() : nullptr,
                                                  dx, dt, nvector, interpol_var, interpol_type, R.work.p, R.err.as<int>(), nullptr)) return rc;
  // (asynchronous, like the sweep of a level in tiles: the counter of missing father cells is read by the next routine that
  //  synchronises anyway -- check_lists)
  R.err_pending = std::max(R.err_pending, ilevel);
  return 0;
}


// ---- self-gravity: the acceleration f(1:ncell,1:3) is computed by force_fine into the host array (and stays authoritative
// there); the device keeps a copy for synchro_hydro_fine, the gravity terms of courant_fine / godunov_fine / set_uold

// f of one level's cells from the host array (after force_fine(ilevel), after a regrid)
int ramses_amd_amrres_load_f(int ngrid, const int *igrid, const double *f) {
  AmrRes &R = g_ar;
  LvlArgs A;
  if (!f) return failf(RAMSES_AMD_EINVAL, "NULL argument");
  if (int rc = set_level(R, ngrid, igrid, A)) return rc;
  if (!R.grav) {
    HCHK(R.f.ensure(sizeof(double) * 3 * (size_t)R.ncell), "hipMalloc f");
    HCHK(hipMemsetAsync(R.f.p, 0, sizeof(double) * 3 * (size_t)R.ncell, nullptr), "memset f");
    R.grav = true;
  }
  if (ngrid == 0) return 0;
  const long tot = (long)ngrid * 8;
  R.hpack.resize((size_t)tot * 3);
  for (int k = 0; k < 3; k++)
    for (int ind = 0; ind < 8; ind++) {
      const double *src = f + (size_t)k * R.ncell_h + R.ncoarse + (size_t)ind * R.ngh - 1;
      double *dst = R.hpack.data() + (size_t)k * tot + (size_t)ind * ngrid;
      for (int i = 0; i < ngrid; i++) dst[i] = src[igrid[i]];
    }
  HCHK(R.pack.ensure(sizeof(double) * (size_t)tot * 3), "hipMalloc");
  HCHK(hipMemcpy(R.pack.p, R.hpack.data(), sizeof(double) * (size_t)tot * 3, hipMemcpyHostToDevice), "H2D f");
  R.f_up_bytes += (int64_t)sizeof(double) * tot * 3;
  hipLaunchKernelGGL(lvl_pack_comp_kernel<false>, dim3(grid_for(tot)), dim3(256), 0, nullptr, R.f.as<double>(), R.pack.as<double>(), R.cur_ig, ngrid, 3,
                     R.ncell, R.ncoarse, R.ngridmax);
  HCHK(hipGetLastError(), "f unpack launch");
  HCHK(hipStreamSynchronize(nullptr), "sync");
  return 0;
}
int ramses_amd_amrres_has_gravity(void) { return g_ar.valid && g_ar.grav ? 1 : 0; }

// f of a level's own cells straight from the device buffer force_fine's kernel filled (d_fpack[3][8][ngrid], the octs in the order
// of igrid: csrc/pois_amr.hip) -- the acceleration of a resident run with several ranks never visits the host (round 6); the
// virtual octs follow with the exchange of direction 7
int ramses_amd_amrres_take_f_device(int ngrid, const int *igrid, const double *d_fpack) {
  AmrRes &R = g_ar;
  LvlArgs A;
  if (!d_fpack) return failf(RAMSES_AMD_EINVAL, "NULL argument");
  if (int rc = set_level(R, ngrid, igrid, A)) return rc;
  if (!R.grav) {
    HCHK(R.f.ensure(sizeof(double) * 3 * (size_t)R.ncell), "hipMalloc f");
    HCHK(hipMemsetAsync(R.f.p, 0, sizeof(double) * 3 * (size_t)R.ncell, nullptr), "memset f");
    R.grav = true;
  }
  if (ngrid == 0) return 0;
  hipLaunchKernelGGL(lvl_pack_comp_kernel<false>, dim3(grid_for((long)ngrid * 8)), dim3(256), 0, nullptr, R.f.as<double>(), const_cast<double *>(d_fpack), R.cur_ig,
                     ngrid, 3, R.ncell, R.ncoarse, R.ngridmax);
  HCHK(hipGetLastError(), "f unpack launch");
  return 0;
}
// f of the listed octs back into the host array (backup_poisson, load_balance: the only host readers of a resident run)
int ramses_amd_amrres_sync_f(int ngrid, const int *igrid, double *f) {
  AmrRes &R = g_ar;
  LvlArgs A;
  if (!f) return failf(RAMSES_AMD_EINVAL, "NULL argument");
  if (!R.grav) return failf(RAMSES_AMD_EINVAL, "sync_f: no acceleration on the device");
  if (int rc = set_level(R, ngrid, igrid, A)) return rc;
  if (ngrid == 0) return 0;
  const long tot = (long)ngrid * 8;
  HCHK(R.pack.ensure(sizeof(double) * (size_t)tot * 3), "hipMalloc");
  hipLaunchKernelGGL(lvl_pack_comp_kernel<true>, dim3(grid_for(tot)), dim3(256), 0, nullptr, R.f.as<double>(), R.pack.as<double>(), R.cur_ig, ngrid, 3,
                     R.ncell, R.ncoarse, R.ngridmax);
  HCHK(hipGetLastError(), "f pack launch");
  R.hpack.resize((size_t)tot * 3);
  HCHK(hipMemcpy(R.hpack.data(), R.pack.p, sizeof(double) * (size_t)tot * 3, hipMemcpyDeviceToHost), "D2H f");
  R.f_down_bytes += (int64_t)sizeof(double) * tot * 3;
  for (int k = 0; k < 3; k++)
    for (int ind = 0; ind < 8; ind++) {
      double *dst = f + (size_t)k * R.ncell_h + R.ncoarse + (size_t)ind * R.ngh - 1;
      const double *src = R.hpack.data() + (size_t)k * tot + (size_t)ind * ngrid;
      for (int i = 0; i < ngrid; i++) dst[igrid[i]] = src[i];
    }
  return 0;
}
// RAMSES_AMD_F_CHECK=1 (patch/force_fine.f90): the largest |device f - host f| over the listed octs, and how many cells differ
int ramses_amd_amrres_compare_f(int ngrid, const int *igrid, const double *f, double *maxdiff, int64_t *ndiff) {
  AmrRes &R = g_ar;
  LvlArgs A;
  if (!f || !maxdiff || !ndiff) return failf(RAMSES_AMD_EINVAL, "NULL argument");
  *maxdiff = 0.0; *ndiff = 0;
  if (!R.grav) return failf(RAMSES_AMD_EINVAL, "compare_f: no acceleration on the device");
  if (int rc = set_level(R, ngrid, igrid, A)) return rc;
  if (ngrid == 0) return 0;
  const long tot = (long)ngrid * 8;
  HCHK(R.pack.ensure(sizeof(double) * (size_t)tot * 3), "hipMalloc");
  hipLaunchKernelGGL(lvl_pack_comp_kernel<true>, dim3(grid_for(tot)), dim3(256), 0, nullptr, R.f.as<double>(), R.pack.as<double>(), R.cur_ig, ngrid, 3,
                     R.ncell, R.ncoarse, R.ngridmax);
  HCHK(hipGetLastError(), "f pack launch");
  std::vector<double> h((size_t)tot * 3);
  HCHK(hipMemcpy(h.data(), R.pack.p, sizeof(double) * (size_t)tot * 3, hipMemcpyDeviceToHost), "D2H f");
  for (int k = 0; k < 3; k++)
    for (int ind = 0; ind < 8; ind++) {
      const double *ref = f + (size_t)k * R.ncell_h + R.ncoarse + (size_t)ind * R.ngh - 1;
      const double *dev = h.data() + (size_t)k * tot + (size_t)ind * ngrid;
      for (int i = 0; i < ngrid; i++) {
        const double d = std::fabs(dev[i] - ref[igrid[i]]);
        if (d > 0.0 || dev[i] != ref[igrid[i]]) { (*ndiff)++; if (d > *maxdiff) *maxdiff = d; }
      }
    }
  return 0;
}
// bytes of f that crossed PCIe since the start: out[0] host -> device (ramses_amd_amrres_load_f), out[1] device -> host
int ramses_amd_amrres_f_traffic(int64_t *out2) {
  if (!out2) return failf(RAMSES_AMD_EINVAL, "NULL argument");
  out2[0] = g_ar.f_up_bytes; out2[1] = g_ar.f_down_bytes;
  return 0;
}

// the oct centres xg(1:ngridmax,1:3) (after refine_fine, with the tree): what rho_fine's deposit needs beyond the tree
int ramses_amd_amrres_xg(const double *xg) {
  AmrRes &R = g_ar;
  if (!R.valid) return failf(RAMSES_AMD_EINVAL, "no resident AMR state");
  if (!xg) return failf(RAMSES_AMD_EINVAL, "NULL argument");
  HCHK(R.xg.ensure(sizeof(double) * 3 * (size_t)R.ngridmax), "hipMalloc xg");
  if (!R.map.on) {
    HCHK(hipMemcpy(R.xg.p, xg, sizeof(double) * 3 * (size_t)R.ngridmax, hipMemcpyHostToDevice), "H2D xg");
  } else {
    HCHK(R.work.ensure(sizeof(double) * 3 * (size_t)R.ngh), "hipMalloc staging");
    HCHK(hipMemcpyAsync(R.work.p, xg, sizeof(double) * 3 * (size_t)R.ngh, hipMemcpyHostToDevice, nullptr), "H2D xg");
    for (int l = 1; l <= R.map.nlev; l++) {
      amrlayout::LevelMap &L = R.map.lev[l];
      hipLaunchKernelGGL(amrlayout::move_oct_kernel, dim3(amrlayout::grid1((long)L.n * 3)), dim3(256), 0, nullptr, L.hoct.as<int>(), L.doct.as<int>(), L.n, 3, R.ngh,
                         R.ngridmax, R.work.as<double>(), R.xg.as<double>());
    }
    HCHK(hipGetLastError(), "xg into the device's numbering");
    HCHK(hipStreamSynchronize(nullptr), "sync");
  }
  R.xg_valid = true;
  return 0;
}

// rho_fine(ilevel,icount)'s hydro deposit (pm/rho_fine.f90:45-60: multipole_fine(l) and cic_from_multipole(l) for
// l = nlevelmax .. ilevel) on the resident density, single rank, periodic nx=ny=nz=1 box, no particles.
//   first[0..nlev], igrid_all: active(l)%igrid for l = ilevel .. nlevelmax one after the other (first[l-ilevel] .. first[l-ilevel+1])
//   rho (host, ncell): the cells of the visited levels receive the deposit;  multipole4: the four sums of cic_from_multipole at
//   levelmin (written when ilevel == levelmin, where the reference has just reset them; untouched otherwise)
int ramses_amd_amrres_rho_fine(const ramses_amd_hydro_params *p, int ilevel, int nlevelmax, int levelmin, int nvector,
                               const int *first, const int *igrid_all, double boxlen_over_nx, double *rho, double *multipole4) {
  AmrRes &R = g_ar;
  if (!R.valid) return failf(RAMSES_AMD_EINVAL, "no resident AMR state (ramses_amd_amrres_load)");
  if (!p || !first || !igrid_all || !rho || !multipole4) return failf(RAMSES_AMD_EINVAL, "NULL argument");
  if (!R.xg_valid) return failf(RAMSES_AMD_EINVAL, "rho_fine: no oct centres on the device (ramses_amd_amrres_xg)");
  if (ilevel < 2 || nlevelmax < ilevel || nvector < 1) return failf(RAMSES_AMD_EINVAL, "rho_fine: bad level range / nvector");
  if (R.ncoarse != 1) return failf(RAMSES_AMD_EUNSUPPORTED, "rho_fine on the device covers a periodic box of one coarse cell");
  const int nlev = nlevelmax - ilevel + 1;
  const int ntot = first[nlev];
  if (first[0] != 0 || ntot < 0) return failf(RAMSES_AMD_EINVAL, "rho_fine: bad list offsets");
  hipStream_t s = nullptr;
  const size_t cb = sizeof(double) * (size_t)R.ncell;
  HCHK(R.mp.ensure(4 * cb), "hipMalloc multipoles"); HCHK(R.rho.ensure(cb), "hipMalloc rho");
  if (R.posof.cap < sizeof(int) * (size_t)R.ngridmax) {
    HCHK(R.posof.ensure(sizeof(int) * (size_t)R.ngridmax), "hipMalloc posof");
    HCHK(hipMemsetAsync(R.posof.p, 0xff, sizeof(int) * (size_t)R.ngridmax, s), "memset posof");
  }
  if (int rc = upload_list(R, R.lists, igrid_all, ntot)) return rc;
  for (int lev = nlevelmax; lev >= ilevel; lev--) {
    const int lo = first[lev - ilevel], n = first[lev - ilevel + 1] - lo;
    if (n < 0 || lo + n > ntot || n > R.ngridmax) return failf(RAMSES_AMD_EINVAL, "rho_fine: bad list of level %d", lev);
    if (n == 0) continue;
    const int *d_ig = R.lists.as<int>() + lo;
    HCHK(launch_amr_rho_level(R.uold.as<double>(), R.mp.as<double>(), R.rho.as<double>(), R.xg.as<double>(), R.son.as<int>(), R.nbor.as<int>(),
                              R.father.as<int>(), d_ig, R.posof.as<int>(), n, nvector, R.ncoarse, R.ngridmax, lev, boxlen_over_nx, p->smallr, s),
         "rho_fine level launch");
    if (lev == levelmin) {
      HCHK(R.mpscratch.ensure(multipole_scratch_bytes((long)n * 8)), "hipMalloc multipole scratch");
      HCHK(launch_multipole_vec(R.mp.as<double>(), d_ig, n, nvector, R.ncell, R.ncoarse, R.ngridmax, R.red.as<double>(), R.mpscratch.p, s),
           "multipole launch");
      HCHK(hipMemcpyAsync(multipole4, R.red.p, sizeof(double) * 4, hipMemcpyDeviceToHost, s), "D2H multipole");
    }
    // the deposit of the level back into the host vector (multigrid_fine / phi_fine_cg / force_fine's diagnostics read it there)
    const long tot = (long)n * 8;
    HCHK(R.pack.ensure(sizeof(double) * (size_t)tot), "hipMalloc");
    hipLaunchKernelGGL(lvl_pack_comp_kernel<true>, dim3(grid_for(tot)), dim3(256), 0, s, R.rho.as<double>(), R.pack.as<double>(), d_ig, n, 1,
                       R.ncell, R.ncoarse, R.ngridmax);
    HCHK(hipGetLastError(), "rho pack launch");
    R.hpack.resize((size_t)tot);
    HCHK(hipMemcpy(R.hpack.data(), R.pack.p, sizeof(double) * (size_t)tot, hipMemcpyDeviceToHost), "D2H rho");
    const int *ig = igrid_all + lo;
    for (int ind = 0; ind < 8; ind++) {
      double *dst = rho + R.ncoarse + (size_t)ind * R.ngh - 1;
      const double *src = R.hpack.data() + (size_t)ind * n;
      for (int i = 0; i < n; i++) dst[ig[i]] = src[i];
    }
  }
  HCHK(hipStreamSynchronize(s), "sync");
  return 0;
}

// rho_fine's hydro deposit with several ranks, one level at a time from nlevelmax down (pm/rho_fine.f90:45-60); the caller does
// the reference's exchanges between the steps (ramses_amd_amrres_halo_* with dir 6, 4, 5):
//   _multipole(l)   multipole_fine(l) on the rank's own octs (igrid_all: n_own own octs followed by the reception octs, n_all)
//   [dir 6]         make_virtual_fine_dp(unew(1,1:4),l): a split cell's son oct may belong to another rank (:814-817)
//   _deposit(l)     cic_from_multipole(l): rho of own + reception cells from the own octs' pseudo-particles
//   [dir 4, dir 5]  make_virtual_reverse_dp(rho,l), make_virtual_fine_dp(rho,l) (:58-59)
//   _finish(l)      rho of the level's cells (own + reception) into the host vector; at levelmin the rank's four sequential
//                   multipole sums (the caller's MPI_ALLREDUCE follows, :176-183)
int ramses_amd_amrres_rho_mpi_multipole(const ramses_amd_hydro_params *p, int ilevel, int n_own, int n_all, const int *igrid_all,
                                        double boxlen_over_nx) {
  AmrRes &R = g_ar;
  if (!R.valid) return failf(RAMSES_AMD_EINVAL, "no resident AMR state (ramses_amd_amrres_load)");
  if (!p || (n_all > 0 && !igrid_all)) return failf(RAMSES_AMD_EINVAL, "NULL argument");
  if (!R.xg_valid) return failf(RAMSES_AMD_EINVAL, "rho_fine: no oct centres on the device (ramses_amd_amrres_xg)");
  if (ilevel < 2 || ilevel > 30 || n_own < 0 || n_all < n_own || n_all > R.ngridmax) return failf(RAMSES_AMD_EINVAL, "rho_fine: bad level / list");
  if (R.ncoarse != 1) return failf(RAMSES_AMD_EUNSUPPORTED, "rho_fine on the device covers a periodic box of one coarse cell");
  hipStream_t s = nullptr;
  const size_t cb = sizeof(double) * (size_t)R.ncell;
  if (R.mp.cap < 4 * cb) { HCHK(R.mp.ensure(4 * cb), "hipMalloc multipoles"); HCHK(hipMemsetAsync(R.mp.p, 0, 4 * cb, s), "memset"); }
  if (R.rho.cap < cb) { HCHK(R.rho.ensure(cb), "hipMalloc rho"); HCHK(hipMemsetAsync(R.rho.p, 0, cb, s), "memset"); }
  if (int rc = upload_list(R, R.lists, igrid_all, n_all)) return rc;
  R.rl_level = ilevel; R.rl_nown = n_own; R.rl_nall = n_all;
  HCHK(launch_amr_multipole_level(R.uold.as<double>(), R.mp.as<double>(), R.xg.as<double>(), R.son.as<int>(), R.lists.as<int>(), n_own, R.ncoarse,
                                  R.ngridmax, ilevel, boxlen_over_nx, p->smallr, s), "multipole_fine launch");
  return 0;
}
int ramses_amd_amrres_rho_mpi_deposit(int ilevel, int nvector, double boxlen_over_nx) {
  AmrRes &R = g_ar;
  if (!R.valid || R.rl_level != ilevel) return failf(RAMSES_AMD_EINVAL, "rho_fine: level %d was not opened by ramses_amd_amrres_rho_mpi_multipole", ilevel);
  if (nvector < 1) return failf(RAMSES_AMD_EINVAL, "rho_fine: bad nvector");
  unsigned hcap = 1024;
  while (hcap < 2u * (unsigned)R.rl_nown) hcap <<= 1;
  HCHK(R.hkeys.ensure(sizeof(unsigned long long) * (size_t)hcap), "hipMalloc"); HCHK(R.hvals.ensure(sizeof(int) * (size_t)hcap), "hipMalloc");
  HCHK(launch_amr_deposit_level(R.mp.as<double>(), R.rho.as<double>(), R.xg.as<double>(), R.lists.as<int>(), R.rl_nown, R.rl_nall, nvector, R.ncoarse,
                                R.ngridmax, ilevel, boxlen_over_nx, R.hkeys.as<unsigned long long>(), R.hvals.as<int>(), hcap, nullptr), "cic_from_multipole launch");
  return 0;
}
int ramses_amd_amrres_rho_mpi_finish(int ilevel, int levelmin, int nvector, const int *igrid_all, double *rho, double *multipole4) {
  AmrRes &R = g_ar;
  if (!R.valid || R.rl_level != ilevel) return failf(RAMSES_AMD_EINVAL, "rho_fine: level %d was not opened by ramses_amd_amrres_rho_mpi_multipole", ilevel);
  if (!rho || !multipole4 || (R.rl_nall > 0 && !igrid_all)) return failf(RAMSES_AMD_EINVAL, "NULL argument");
  hipStream_t s = nullptr;
  const int n_own = R.rl_nown, n = R.rl_nall;
  R.rl_level = 0;
  if (ilevel == levelmin) {
    for (int d = 0; d < 4; d++) multipole4[d] = 0.0;
    if (n_own > 0) {
      HCHK(R.mpscratch.ensure(multipole_scratch_bytes((long)n_own * 8)), "hipMalloc multipole scratch");
      HCHK(launch_multipole_vec(R.mp.as<double>(), R.lists.as<int>(), n_own, nvector, R.ncell, R.ncoarse, R.ngridmax, R.red.as<double>(), R.mpscratch.p, s),
           "multipole launch");
      HCHK(hipMemcpyAsync(multipole4, R.red.p, sizeof(double) * 4, hipMemcpyDeviceToHost, s), "D2H multipole");
    }
  }
  // (ramses_amd_amrres_rho_keep(1): the solver and force_fine of this level read the deposit on the device -- the distributed
  //  dense multigrid of a uniform run, round 6 -- and the host vector is fetched by ramses_amd_amrres_sync_rho when somebody asks)
  if (n > 0 && !R.rho_keep) {
    const long tot = (long)n * 8;
    HCHK(R.pack.ensure(sizeof(double) * (size_t)tot), "hipMalloc");
    hipLaunchKernelGGL(lvl_pack_comp_kernel<true>, dim3(grid_for(tot)), dim3(256), 0, s, R.rho.as<double>(), R.pack.as<double>(), R.lists.as<int>(), n, 1,
                       R.ncell, R.ncoarse, R.ngridmax);
    HCHK(hipGetLastError(), "rho pack launch");
    R.hpack.resize((size_t)tot);
    HCHK(hipMemcpy(R.hpack.data(), R.pack.p, sizeof(double) * (size_t)tot, hipMemcpyDeviceToHost), "D2H rho");
    R.rho_down_bytes += (int64_t)sizeof(double) * tot;
    for (int ind = 0; ind < 8; ind++) {
      double *dst = rho + R.ncoarse + (size_t)ind * R.ngh - 1;
      const double *src = R.hpack.data() + (size_t)ind * n;
      for (int i = 0; i < n; i++) dst[igrid_all[i]] = src[i];
    }
  }
  HCHK(hipStreamSynchronize(s), "sync");
  return 0;
}
int ramses_amd_amrres_rho_keep(int on) { g_ar.rho_keep = on != 0; return 0; }
// rho of the listed octs back into the host vector (backup_poisson; a solver that reads the host vector after all)
int ramses_amd_amrres_sync_rho(int ngrid, const int *igrid, double *rho) {
  AmrRes &R = g_ar;
  LvlArgs A;
  if (!rho) return failf(RAMSES_AMD_EINVAL, "NULL argument");
  if (!R.rho.p) return failf(RAMSES_AMD_EINVAL, "sync_rho: rho_fine has not run on the device");
  if (int rc = set_level(R, ngrid, igrid, A)) return rc;
  if (ngrid == 0) return 0;
  const long tot = (long)ngrid * 8;
  HCHK(R.pack.ensure(sizeof(double) * (size_t)tot), "hipMalloc");
  hipLaunchKernelGGL(lvl_pack_comp_kernel<true>, dim3(grid_for(tot)), dim3(256), 0, nullptr, R.rho.as<double>(), R.pack.as<double>(), R.cur_ig, ngrid, 1,
                     R.ncell, R.ncoarse, R.ngridmax);
  HCHK(hipGetLastError(), "rho pack launch");
  R.hpack.resize((size_t)tot);
  HCHK(hipMemcpy(R.hpack.data(), R.pack.p, sizeof(double) * (size_t)tot, hipMemcpyDeviceToHost), "D2H rho");
  R.rho_down_bytes += (int64_t)sizeof(double) * tot;
  for (int ind = 0; ind < 8; ind++) {
    double *dst = rho + R.ncoarse + (size_t)ind * R.ngh - 1;
    const double *src = R.hpack.data() + (size_t)ind * ngrid;
    for (int i = 0; i < ngrid; i++) dst[igrid[i]] = src[i];
  }
  return 0;
}
// the deposit of the rank's own octs into a dense brick on the device: brick[order[ind * ngrid + g]] = rho(cell ind of oct igrid[g])
// (order: the list of ramses_amd_mgdist_set_order); and max |rho| over those cells (force_fine's diagnostic, :177-181)
namespace {
__global__ __launch_bounds__(256) void rho_to_brick_kernel(const double *__restrict__ rho, const int *__restrict__ ig, const int *__restrict__ order, long ngrid,
                                                           long ncoarse, long ngridmax, double *__restrict__ brick) {
  const long k = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (k < 8 * ngrid) brick[order[k]] = rho[ncoarse + (k / ngrid) * ngridmax + ig[k % ngrid] - 1];
}
__global__ __launch_bounds__(256) void rho_absmax_kernel(const double *__restrict__ rho, const int *__restrict__ ig, long ngrid, long ncoarse, long ngridmax,
                                                         unsigned long long *__restrict__ out) {
  double m = 0.0;
  for (long k = (long)blockIdx.x * blockDim.x + threadIdx.x; k < 8 * ngrid; k += (long)gridDim.x * blockDim.x)
    m = fmax(m, fabs(rho[ncoarse + (k / ngrid) * ngridmax + ig[k % ngrid] - 1]));
  // (non-negative doubles order like their bit patterns)
  atomicMax(out, (unsigned long long)__double_as_longlong(m));
}
}  // namespace
int ramses_amd_amrres_rho_to_brick(int ngrid, const int *igrid, const int *d_order, double *d_brick) {
  AmrRes &R = g_ar;
  LvlArgs A;
  if (!d_order || !d_brick) return failf(RAMSES_AMD_EINVAL, "NULL argument");
  if (!R.rho.p) return failf(RAMSES_AMD_EINVAL, "rho_to_brick: rho_fine has not run on the device");
  if (int rc = set_level(R, ngrid, igrid, A)) return rc;
  if (ngrid == 0) return 0;
  hipLaunchKernelGGL(rho_to_brick_kernel, dim3((unsigned)(((long)ngrid * 8 + 255) / 256)), dim3(256), 0, nullptr, R.rho.as<double>(), R.cur_ig, d_order, (long)ngrid,
                     R.ncoarse, R.ngridmax, d_brick);
  HCHK(hipGetLastError(), "rho -> brick launch");
  return 0;
}
int ramses_amd_amrres_rho_absmax(int ngrid, const int *igrid, double *out) {
  AmrRes &R = g_ar;
  LvlArgs A;
  if (!out) return failf(RAMSES_AMD_EINVAL, "NULL argument");
  *out = 0.0;
  if (!R.rho.p) return failf(RAMSES_AMD_EINVAL, "rho_absmax: rho_fine has not run on the device");
  if (int rc = set_level(R, ngrid, igrid, A)) return rc;
  if (ngrid == 0) return 0;
  HCHK(R.okbuf.ensure(sizeof(unsigned long long) * 2), "hipMalloc");
  unsigned long long *d = reinterpret_cast<unsigned long long *>(R.okbuf.p);
  HCHK(hipMemsetAsync(d, 0, sizeof(unsigned long long), nullptr), "memset");
  hipLaunchKernelGGL(rho_absmax_kernel, dim3(grid_for((long)ngrid * 8)), dim3(256), 0, nullptr, R.rho.as<double>(), R.cur_ig, (long)ngrid, R.ncoarse, R.ngridmax, d);
  HCHK(hipGetLastError(), "max |rho| launch");
  unsigned long long bits = 0;
  HCHK(hipMemcpy(&bits, d, sizeof(bits), hipMemcpyDeviceToHost), "D2H");
  memcpy(out, &bits, sizeof(double));
  return 0;
}
// bytes of rho_fine's deposit that went back to the host vector since the start
int64_t ramses_amd_amrres_rho_traffic(void) { return g_ar.rho_down_bytes; }

// the density uold(:,1) of one level's cells back into the host array (rho_fine's multipole_fine reads nothing else)
int ramses_amd_amrres_sync_density(int ngrid, const int *igrid, double *uold) {
  AmrRes &R = g_ar;
  LvlArgs A;
  if (int rc = set_level(R, ngrid, igrid, A)) return rc;
  if (uold != R.h_uold) return failf(RAMSES_AMD_EINVAL, "sync_density: not the array the state was loaded from");
  if (ngrid == 0) return 0;
  const long tot = (long)ngrid * 8;
  HCHK(R.pack.ensure(sizeof(double) * (size_t)tot), "hipMalloc");
  hipLaunchKernelGGL(lvl_pack_comp_kernel<true>, dim3(grid_for(tot)), dim3(256), 0, nullptr, R.uold.as<double>(), R.pack.as<double>(), R.cur_ig, ngrid, 1,
                     R.ncell, R.ncoarse, R.ngridmax);
  HCHK(hipGetLastError(), "density pack launch");
  R.hpack.resize((size_t)tot);
  HCHK(hipMemcpy(R.hpack.data(), R.pack.p, sizeof(double) * (size_t)tot, hipMemcpyDeviceToHost), "D2H density");
  for (int ind = 0; ind < 8; ind++) {
    double *dst = uold + R.ncoarse + (size_t)ind * R.ngh - 1;
    const double *src = R.hpack.data() + (size_t)ind * ngrid;
    for (int i = 0; i < ngrid; i++) dst[igrid[i]] = src[i];
  }
  return 0;
}

int ramses_amd_amrres_synchro(const ramses_amd_hydro_params *p, int ngrid, const int *igrid, double dteff) {
  if (!p) return failf(RAMSES_AMD_EINVAL, "NULL argument");
  LvlArgs A;
  if (int rc = set_level(g_ar, ngrid, igrid, A)) return rc;
  if (!g_ar.grav) return failf(RAMSES_AMD_EINVAL, "synchro_hydro_fine: no acceleration on the device (ramses_amd_amrres_load_f)");
  if (ngrid == 0) return 0;
  if (int rc = sorted_list(g_ar, A)) return rc;
  hipLaunchKernelGGL(lvl_synchro_kernel, dim3(grid_for((long)ngrid * 8)), dim3(256), 0, nullptr, A, g_ar.f.as<double>(), dteff, p->smallr);
  HCHK(hipGetLastError(), "synchro launch");
  return 0;
}

// set_uold with poisson: add_gravity_source_terms on unew, then the scalar fix and uold = unew
int ramses_amd_amrres_set_uold_grav(const ramses_amd_hydro_params *p, int ngrid, const int *igrid, double dt) {
  if (!p) return failf(RAMSES_AMD_EINVAL, "NULL argument");
  LvlArgs A;
  if (int rc = set_level(g_ar, ngrid, igrid, A)) return rc;
  if (!g_ar.grav) return failf(RAMSES_AMD_EINVAL, "set_uold: no acceleration on the device (ramses_amd_amrres_load_f)");
  if (ngrid == 0) return 0;
  if (int rc = sorted_list(g_ar, A)) return rc;
  hipLaunchKernelGGL(lvl_gravity_source_kernel, dim3(grid_for((long)ngrid * 8)), dim3(256), 0, nullptr, A, g_ar.f.as<double>(), dt, p->smallr);
  hipLaunchKernelGGL(lvl_set_uold_kernel, dim3(grid_for((long)ngrid * 8)), dim3(256), 0, nullptr, A, p->smallr);
  HCHK(hipGetLastError(), "set_uold launch");
  return 0;
}


// ---- pressure_fix: divu / enew live on the device only (they are scratch of one hydro step)
int ramses_amd_amrres_enable_pfix(void) {
  AmrRes &R = g_ar;
  if (!R.valid) return failf(RAMSES_AMD_EINVAL, "no resident AMR state");
  if (R.pfix) return 0;
  HCHK(R.divu.ensure(sizeof(double) * (size_t)R.ncell), "hipMalloc divu"); HCHK(R.enew.ensure(sizeof(double) * (size_t)R.ncell), "hipMalloc enew");
  HCHK(hipMemsetAsync(R.divu.p, 0, sizeof(double) * (size_t)R.ncell, nullptr), "memset"); HCHK(hipMemsetAsync(R.enew.p, 0, sizeof(double) * (size_t)R.ncell, nullptr), "memset");
  R.pfix = true;
  return 0;
}
// set_unew with pressure_fix: unew = uold, divu = 0, enew = internal energy
int ramses_amd_amrres_set_unew_pfix(const ramses_amd_hydro_params *p, int ngrid, const int *igrid) {
  if (!p) return failf(RAMSES_AMD_EINVAL, "NULL argument");
  LvlArgs A;
  if (int rc = set_level(g_ar, ngrid, igrid, A)) return rc;
  if (!g_ar.pfix) return failf(RAMSES_AMD_EINVAL, "set_unew: pressure_fix not enabled (ramses_amd_amrres_enable_pfix)");
  if (ngrid == 0) return 0;
  if (int rc = sorted_list(g_ar, A)) return rc;
  const dim3 g(grid_for((long)ngrid * 8)), b(256);
  hipLaunchKernelGGL(lvl_copy_kernel, g, b, 0, nullptr, A, A.unew, A.uold);
  hipLaunchKernelGGL(lvl_pfix_init_kernel, g, b, 0, nullptr, A, g_ar.divu.as<double>(), g_ar.enew.as<double>(), p->smallr);
  HCHK(hipGetLastError(), "set_unew launch");
  return 0;
}
// set_uold with pressure_fix: (add_gravity_source_terms,) add_pdv_source_terms, the scalar fix and uold = unew, the energy switch
int ramses_amd_amrres_set_uold_pfix(const ramses_amd_hydro_params *p, int ngrid, const int *igrid, double dt, double dx_loc, double beta_fix,
                                    double hexp) {
  if (!p) return failf(RAMSES_AMD_EINVAL, "NULL argument");
  LvlArgs A;
  if (int rc = set_level(g_ar, ngrid, igrid, A)) return rc;
  AmrRes &R = g_ar;
  if (!R.pfix) return failf(RAMSES_AMD_EINVAL, "set_uold: pressure_fix not enabled (ramses_amd_amrres_enable_pfix)");
  if (ngrid == 0) return 0;
  if (int rc = sorted_list(g_ar, A)) return rc;
  const dim3 g(grid_for((long)ngrid * 8)), b(256);
  if (R.grav) hipLaunchKernelGGL(lvl_gravity_source_kernel, g, b, 0, nullptr, A, R.f.as<double>(), dt, p->smallr);
  hipLaunchKernelGGL(lvl_pdv_kernel, g, b, 0, nullptr, A, R.enew.as<double>(), dx_loc, dt, p->gamma, p->smallr);
  hipLaunchKernelGGL(lvl_set_uold_kernel, g, b, 0, nullptr, A, p->smallr);
  hipLaunchKernelGGL(lvl_pfix_switch_kernel, g, b, 0, nullptr, A, R.divu.as<double>(), R.enew.as<double>(), dx_loc, dt, beta_fix, hexp, p->smallr);
  HCHK(hipGetLastError(), "set_uold launch");
  return 0;
}


// ---- several MPI ranks: the virtual-boundary exchanges of amr_step on the resident cell vectors ----------------------------
// (amr/virtual_boundaries.f90:373-528 make_virtual_fine_dp, :693-983 make_virtual_reverse_dp; callers amr/amr_step.f90:61,
// 287,397,417-418,505).  The device holds the reference's own layout, so the reference's own communicators -- oct lists per
// peer -- address it directly.  A message is what the reference sends, all variables at once: for the octs i = 1..n of the
// list, buf[(v*8 + ind)*n + i] = vec(ncoarse + ind*ngridmax + igrid(i), v).

// which column of a host array is xx?  1..ncol, or 0
int ramses_amd_which_column(const double *xx, const double *base, int64_t ncell, int ncol) {
  if (!xx || !base || ncell < 1 || ncol < 1) return 0;
  const ptrdiff_t off = xx - base;
  if (off < 0 || off % ncell != 0 || off / ncell >= ncol) return 0;
  return (int)(off / ncell) + 1;
}

int ramses_amd_amrres_comm_epoch(int ilevel) {
  AmrRes &R = g_ar;
  if (ilevel < 0 || (size_t)ilevel >= R.comm.size()) return -1;
  return R.comm[ilevel].epoch;
}

// the communicators of a level after build_comm: em_n / rc_n [ncpu] octs per peer, em_ig / rc_ig the concatenated lists
int ramses_amd_amrres_comm_set(int ilevel, int epoch, int ncpu, const int *em_n, const int *em_ig, const int *rc_n, const int *rc_ig) {
  AmrRes &R = g_ar;
  if (ilevel < 1 || ilevel > 64 || ncpu < 1 || !em_n || !rc_n || epoch < 0) return failf(RAMSES_AMD_EINVAL, "comm_set: bad argument");
  if ((size_t)ilevel >= R.comm.size()) R.comm.resize((size_t)ilevel + 1);
  CommLevel &L = R.comm[ilevel];
  L.epoch = -1;
  L.ncpu = ncpu;
  L.em_first.assign((size_t)ncpu + 1, 0); L.rc_first.assign((size_t)ncpu + 1, 0);
  for (int c = 0; c < ncpu; c++) {
    if (em_n[c] < 0 || rc_n[c] < 0) return failf(RAMSES_AMD_EINVAL, "comm_set: negative list length");
    L.em_first[c + 1] = L.em_first[c] + em_n[c];
    L.rc_first[c + 1] = L.rc_first[c] + rc_n[c];
  }
  const int nem = L.em_first[ncpu], nrc = L.rc_first[ncpu];
  if ((nem > 0 && !em_ig) || (nrc > 0 && !rc_ig)) return failf(RAMSES_AMD_EINVAL, "comm_set: NULL list");
  HCHK(L.em_ig.ensure(sizeof(int) * (size_t)(nem > 0 ? nem : 1)), "hipMalloc"); HCHK(L.rc_ig.ensure(sizeof(int) * (size_t)(nrc > 0 ? nrc : 1)), "hipMalloc");
  HCHK(L.em_raw.ensure(sizeof(int) * (size_t)(nem > 0 ? nem : 1)), "hipMalloc"); HCHK(L.rc_raw.ensure(sizeof(int) * (size_t)(nrc > 0 ? nrc : 1)), "hipMalloc");
  if (nem > 0) HCHK(hipMemcpy(L.em_raw.p, em_ig, sizeof(int) * (size_t)nem, hipMemcpyHostToDevice), "H2D emission list");
  if (nrc > 0) HCHK(hipMemcpy(L.rc_raw.p, rc_ig, sizeof(int) * (size_t)nrc, hipMemcpyHostToDevice), "H2D reception list");
  L.serial = -1;             // translated into device indices by the exchange that uses them (comm_of)
  L.epoch = epoch;
  return 0;
}

namespace {
int comm_of(AmrRes &R, int ilevel, CommLevel *&L) {
  if (!R.valid) return failf(RAMSES_AMD_EINVAL, "no resident AMR state (ramses_amd_amrres_load)");
  if (ilevel < 1 || (size_t)ilevel >= R.comm.size() || R.comm[ilevel].epoch < 0)
    return failf(RAMSES_AMD_EINVAL, "level %d: no communicators on the device (ramses_amd_amrres_comm_set)", ilevel);
  L = &R.comm[ilevel];
  if (L->serial != R.map.serial) {
    // the lists in the numbers of the layout in force (a regrid of OTHER levels renumbers nothing here, but costs nothing either)
    const int nem = L->em_first[L->ncpu], nrc = L->rc_first[L->ncpu];
    if (R.map.on) {
      if (nem > 0) hipLaunchKernelGGL(amrlayout::xlate_list_copy_kernel, dim3(amrlayout::grid1(nem)), dim3(256), 0, nullptr, L->em_raw.as<int>(), L->em_ig.as<int>(), nem,
                                      R.map.perm.as<int>(), R.ngh, R.bad.as<int>());
      if (nrc > 0) hipLaunchKernelGGL(amrlayout::xlate_list_copy_kernel, dim3(amrlayout::grid1(nrc)), dim3(256), 0, nullptr, L->rc_raw.as<int>(), L->rc_ig.as<int>(), nrc,
                                      R.map.perm.as<int>(), R.ngh, R.bad.as<int>());
      HCHK(hipGetLastError(), "communicator translation");
    } else {
      if (nem > 0) HCHK(hipMemcpyAsync(L->em_ig.p, L->em_raw.p, sizeof(int) * (size_t)nem, hipMemcpyDeviceToDevice, nullptr), "copy");
      if (nrc > 0) HCHK(hipMemcpyAsync(L->rc_ig.p, L->rc_raw.p, sizeof(int) * (size_t)nrc, hipMemcpyDeviceToDevice, nullptr), "copy");
    }
    L->serial = R.map.serial;
  }
  return 0;
}
// dir 0: make_virtual_fine_dp on uold(:,1:nvar); 1: make_virtual_reverse_dp on unew(:,1:nvar); 2 / 3: the same on enew / divu;
// rho_fine with several ranks: 4 make_virtual_reverse_dp(rho), 5 make_virtual_fine_dp(rho), 6 make_virtual_fine_dp on the multipoles
// force_fine with several ranks: 7 make_virtual_fine_dp on f(:,1:3) (poisson/force_fine.f90:137-139), one exchange for the three
struct HaloSpec { double *vec; int ncomp; bool reverse; };
int halo_spec(AmrRes &R, int dir, HaloSpec &S) {
  switch (dir) {
    case 0: S = {R.uold.as<double>(), R.nvar, false}; return 0;
    case 1: S = {R.unew.as<double>(), R.nvar, true}; return 0;
    case 2: case 3:
      if (!R.pfix) return failf(RAMSES_AMD_EINVAL, "halo on enew/divu: pressure_fix is not enabled");
      S = {dir == 2 ? R.enew.as<double>() : R.divu.as<double>(), 1, true}; return 0;
    case 7:
      if (!R.grav) return failf(RAMSES_AMD_EINVAL, "halo on f: no acceleration on the device");
      S = {R.f.as<double>(), 3, false}; return 0;
    case 4: case 5: case 6:
      if (!R.rho.p || !R.mp.p) return failf(RAMSES_AMD_EINVAL, "halo on rho / the multipoles: rho_fine has not run on the device");
      if (dir == 6) S = {R.mp.as<double>(), 4, false};
      else S = {R.rho.as<double>(), 1, dir == 4};
      return 0;
  }
  return failf(RAMSES_AMD_EINVAL, "halo: bad direction %d", dir);
}
// gather every peer's message into sendbuf; fills the [ncpu+1] offset tables (in doubles)
int halo_pack(AmrRes &R, CommLevel &L, const HaloSpec &S) {
  const std::vector<int> &sf = S.reverse ? L.rc_first : L.em_first, &rf = S.reverse ? L.em_first : L.rc_first;
  const int *sig = S.reverse ? L.rc_ig.as<int>() : L.em_ig.as<int>();
  const size_t per = (size_t)8 * S.ncomp;
  R.f_send_off.assign((size_t)L.ncpu + 1, 0); R.f_recv_off.assign((size_t)L.ncpu + 1, 0);
  for (int c = 0; c <= L.ncpu; c++) { R.f_send_off[c] = (int64_t)(per * sf[c]); R.f_recv_off[c] = (int64_t)(per * rf[c]); }
  HCHK(R.sendbuf.ensure(sizeof(double) * per * (size_t)(sf[L.ncpu] > 0 ? sf[L.ncpu] : 1)), "hipMalloc sendbuf");
  HCHK(R.recvbuf.ensure(sizeof(double) * per * (size_t)(rf[L.ncpu] > 0 ? rf[L.ncpu] : 1)), "hipMalloc recvbuf");
  for (int c = 0; c < L.ncpu; c++) {
    const int n = sf[c + 1] - sf[c];
    if (n <= 0) continue;
    hipLaunchKernelGGL(lvl_pack_comp_kernel<true>, dim3(grid_for((long)n * 8)), dim3(256), 0, nullptr, S.vec, R.sendbuf.as<double>() + per * sf[c],
                       sig + sf[c], n, S.ncomp, R.ncell, R.ncoarse, R.ngridmax);
  }
  HCHK(hipGetLastError(), "halo pack launch");
  return 0;
}
// scatter (forward) or accumulate in icpu order (reverse) what arrived in recvbuf
int halo_unpack(AmrRes &R, CommLevel &L, const HaloSpec &S) {
  const std::vector<int> &rf = S.reverse ? L.em_first : L.rc_first;
  const int *rig = S.reverse ? L.em_ig.as<int>() : L.rc_ig.as<int>();
  const size_t per = (size_t)8 * S.ncomp;
  for (int c = 0; c < L.ncpu; c++) {
    const int n = rf[c + 1] - rf[c];
    if (n <= 0) continue;
    if (S.reverse) hipLaunchKernelGGL(lvl_acc_comp_kernel, dim3(grid_for((long)n * 8)), dim3(256), 0, nullptr, S.vec, R.recvbuf.as<double>() + per * rf[c],
                                      rig + rf[c], n, S.ncomp, R.ncell, R.ncoarse, R.ngridmax);
    else hipLaunchKernelGGL(lvl_pack_comp_kernel<false>, dim3(grid_for((long)n * 8)), dim3(256), 0, nullptr, S.vec, R.recvbuf.as<double>() + per * rf[c],
                            rig + rf[c], n, S.ncomp, R.ncell, R.ncoarse, R.ngridmax);
  }
  HCHK(hipGetLastError(), "halo unpack launch");
  return 0;
}
}  // namespace

// make_boundary_hydro(ilevel) of a run with physical boundaries (hydro/hydro_boundary.f90:5-269): nregion regions in the
// reference's order (a later region may read what an earlier one wrote: corners), btype = boundary_type(1:nregion)
// (1-6 reflexive, 11-16 free), ngrid[r] octs of region r on this level, igrid = the regions' lists one after the other,
// nvector = the reference build's NVECTOR (the chunking of its loop decides what a region two octs deep reads); imposed =
// for the imposed regions (21-26) in order, the states boundana returned, [nvar][8][ngrid[r]] each (host; NULL: none).
int ramses_amd_amrres_boundary_hydro(int nregion, const int *btype, const int *ngrid, const int *igrid, int no_inflow, double smallr,
                                     int nvector, const double *imposed) {
  AmrRes &R = g_ar;
  if (!R.valid) return failf(RAMSES_AMD_EINVAL, "no resident AMR state (ramses_amd_amrres_load)");
  if (nregion < 0 || (nregion > 0 && (!btype || !ngrid))) return failf(RAMSES_AMD_EINVAL, "bad boundary description");
  if (R.nvar > 8) return failf(RAMSES_AMD_EUNSUPPORTED, "make_boundary_hydro on the device: nvar <= 8 (got %d)", R.nvar);
  if (nvector < 1) return failf(RAMSES_AMD_EINVAL, "nvector must be >= 1");
  long ntot = 0;
  int nmax = 0;
  for (int r = 0; r < nregion; r++) {
    const int k = btype[r] / 10, d = btype[r] % 10;
    if (k < 0 || k > 2 || d < 1 || d > 6) return failf(RAMSES_AMD_EINVAL, "make_boundary_hydro: bad boundary_type %d", btype[r]);
    if (k == 2 && !imposed) return failf(RAMSES_AMD_EINVAL, "make_boundary_hydro: an imposed boundary needs its states (boundana's output)");
    if (ngrid[r] < 0) return failf(RAMSES_AMD_EINVAL, "bad boundary oct count");
    ntot += ngrid[r];
    if (ngrid[r] > nmax) nmax = ngrid[r];
  }
  if (ntot == 0) return 0;
  if (!igrid) return failf(RAMSES_AMD_EINVAL, "NULL oct list");
  HCHK(R.bnd_list.ensure(sizeof(int) * (size_t)ntot), "hipMalloc boundary octs");
  HCHK(R.bnd_tmp.ensure(sizeof(double) * (size_t)nmax * 8 * (size_t)R.nvar), "hipMalloc boundary states");
  if (!R.bnd_pos.p || R.bnd_pos.cap < sizeof(int) * (size_t)R.ngridmax) R.bnd_pos_clean = false;
  HCHK(R.bnd_pos.ensure(sizeof(int) * (size_t)R.ngridmax), "hipMalloc boundary positions");
  if (!R.bnd_pos_clean) HCHK(hipMemsetAsync(R.bnd_pos.p, 0, sizeof(int) * (size_t)R.ngridmax, nullptr), "memset");
  // (the marks of a region are set and cleared around its kernels: an error return in between leaves marks behind, so the
  //  table counts as clean again only after the last region's clear has been queued)
  R.bnd_pos_clean = false;
  if (R.nvar < 5) return failf(RAMSES_AMD_EUNSUPPORTED, "make_boundary_hydro on the device: the 3-D hydro layout (rho, rho u, rho v, rho w, E first)");
  if (int rc = upload_list(R, R.bnd_list, igrid, (int)ntot)) return rc;
  BndArgs A;
  A.uold = R.uold.as<double>(); A.tmp = R.bnd_tmp.as<double>();
  A.son = R.son.as<int>(); A.nbor = R.nbor.as<int>(); A.pos = R.bnd_pos.as<int>();
  A.nvar = R.nvar; A.no_inflow = no_inflow; A.nvector = nvector; A.ncell = R.ncell; A.ncoarse = R.ncoarse; A.ngridmax = R.ngridmax; A.smallr = smallr;
  long off = 0, imp_off = 0;
  for (int r = 0; r < nregion; r++) {
    const int n = ngrid[r];
    if (n > 0) {
      A.list = R.bnd_list.as<int>() + off; A.n = n; A.type = btype[r];
      if (btype[r] / 10 == 2) {
        // imposed boundary (:215-241): the caller evaluated boundana for every cell of the region; [nvar][8][n] like tmp
        const size_t cnt = (size_t)n * 8 * (size_t)R.nvar;
        HCHK(hipMemcpyAsync(A.tmp, imposed + imp_off, sizeof(double) * cnt, hipMemcpyHostToDevice, nullptr), "H2D imposed states");
        imp_off += (long)cnt;
        hipLaunchKernelGGL(bnd_store_kernel, dim3(grid_for((long)n * 8)), dim3(256), 0, nullptr, A);
      } else {
        hipLaunchKernelGGL(bnd_mark_kernel, dim3(grid_for(n)), dim3(256), 0, nullptr, A, 0);
        hipLaunchKernelGGL(bnd_compute_kernel, dim3(grid_for((long)n * 8)), dim3(256), 0, nullptr, A);
        hipLaunchKernelGGL(bnd_store_kernel, dim3(grid_for((long)n * 8)), dim3(256), 0, nullptr, A);
        hipLaunchKernelGGL(bnd_mark_kernel, dim3(grid_for(n)), dim3(256), 0, nullptr, A, 1);
      }
    }
    off += n;
  }
  HCHK(hipGetLastError(), "make_boundary_hydro launch");
  // the list buffer is reused by the next call: the copy above must not overtake these launches, and the caller's list may go
  HCHK(hipStreamSynchronize(nullptr), "sync");
  R.bnd_pos_clean = true;
  return 0;
}

// set_unew's second loop: unew (and divu, enew with pressure_fix) of the virtual octs = 0
int ramses_amd_amrres_zero_unew_virtual(int ilevel) {
  AmrRes &R = g_ar;
  CommLevel *L;
  if (int rc = comm_of(R, ilevel, L)) return rc;
  const int n = L->rc_first[L->ncpu];
  if (n <= 0) return 0;
  const dim3 g(grid_for((long)n * 8)), b(256);
  hipLaunchKernelGGL(lvl_zero_comp_kernel, g, b, 0, nullptr, R.unew.as<double>(), L->rc_ig.as<int>(), n, R.nvar, R.ncell, R.ncoarse, R.ngridmax);
  if (R.pfix) {
    hipLaunchKernelGGL(lvl_zero_comp_kernel, g, b, 0, nullptr, R.divu.as<double>(), L->rc_ig.as<int>(), n, 1, R.ncell, R.ncoarse, R.ngridmax);
    hipLaunchKernelGGL(lvl_zero_comp_kernel, g, b, 0, nullptr, R.enew.as<double>(), L->rc_ig.as<int>(), n, 1, R.ncell, R.ncoarse, R.ngridmax);
  }
  HCHK(hipGetLastError(), "set_unew (virtual octs) launch");
  return 0;
}

// One exchange over RCCL: pack, one grouped send/recv (a message per peer), unpack / accumulate; asynchronous
extern "C" int ramses_amd_rccl_exchange(int npeer, const int *peer, const double *d_send, const int64_t *send_off, const int64_t *send_cnt,
                                        double *d_recv, const int64_t *recv_off, const int64_t *recv_cnt, void *stream);
int ramses_amd_amrres_halo_rccl(int ilevel, int dir, int myid) {
  AmrRes &R = g_ar;
  CommLevel *L;
  HaloSpec S;
  if (int rc = comm_of(R, ilevel, L)) return rc;
  if (int rc = halo_spec(R, dir, S)) return rc;
  if (int rc = halo_pack(R, *L, S)) return rc;
  std::vector<int> peer;
  std::vector<int64_t> so, sc, ro, rcn;
  for (int c = 0; c < L->ncpu; c++) {
    const int64_t ns = R.f_send_off[c + 1] - R.f_send_off[c], nr = R.f_recv_off[c + 1] - R.f_recv_off[c];
    if (ns == 0 && nr == 0) continue;
    // c == myid - 1 (a communicator of the rank with itself) is legal: the grouped exchange matches the send to self with
    // the receive from self; build_comm never produces one, the single-rank execution test of the transport does
    (void)myid;
    peer.push_back(c); so.push_back(R.f_send_off[c]); sc.push_back(ns); ro.push_back(R.f_recv_off[c]); rcn.push_back(nr);
  }
  if (int rc = ramses_amd_rccl_exchange((int)peer.size(), peer.data(), R.sendbuf.as<double>(), so.data(), sc.data(), R.recvbuf.as<double>(),
                                        ro.data(), rcn.data(), nullptr)) return rc;
  return halo_unpack(R, *L, S);
}

// The same with the caller's own MPI as transport (several ranks on one GPU, or no RCCL): stage_out packs on the device and
// hands pinned host buffers over -- the message for peer icpu at h_send + send_off[icpu-1], send_off[icpu]-send_off[icpu-1]
// doubles, likewise h_recv / recv_off -- stage_in applies what arrived.  Addresses as integers for c_f_pointer.
int ramses_amd_amrres_halo_stage_out(int ilevel, int dir, int ncpu, int64_t *h_send_addr, int64_t *h_recv_addr, int64_t *send_off, int64_t *recv_off) {
  AmrRes &R = g_ar;
  CommLevel *L;
  HaloSpec S;
  if (!h_send_addr || !h_recv_addr || !send_off || !recv_off) return failf(RAMSES_AMD_EINVAL, "NULL argument");
  if (int rc = comm_of(R, ilevel, L)) return rc;
  if (ncpu != L->ncpu) return failf(RAMSES_AMD_EINVAL, "ncpu mismatch");
  if (int rc = halo_spec(R, dir, S)) return rc;
  if (int rc = halo_pack(R, *L, S)) return rc;
  const size_t ns = (size_t)R.f_send_off[ncpu], nr = (size_t)R.f_recv_off[ncpu];
  HCHK(R.h_send.ensure(sizeof(double) * (ns > 0 ? ns : 1)), "hipHostMalloc"); HCHK(R.h_recv.ensure(sizeof(double) * (nr > 0 ? nr : 1)), "hipHostMalloc");
  if (ns > 0) HCHK(hipMemcpyAsync(R.h_send.p, R.sendbuf.p, sizeof(double) * ns, hipMemcpyDeviceToHost, nullptr), "D2H halo");
  HCHK(hipStreamSynchronize(nullptr), "sync");
  *h_send_addr = (int64_t)(intptr_t)R.h_send.p; *h_recv_addr = (int64_t)(intptr_t)R.h_recv.p;
  for (int c = 0; c <= ncpu; c++) { send_off[c] = R.f_send_off[c]; recv_off[c] = R.f_recv_off[c]; }
  R.halo_level = ilevel; R.halo_dir = dir;
  return 0;
}
int ramses_amd_amrres_halo_stage_in(int ilevel, int dir) {
  AmrRes &R = g_ar;
  CommLevel *L;
  HaloSpec S;
  if (int rc = comm_of(R, ilevel, L)) return rc;
  if (int rc = halo_spec(R, dir, S)) return rc;
  if (R.halo_level != ilevel || R.halo_dir != dir || R.f_recv_off.size() != (size_t)L->ncpu + 1)
    return failf(RAMSES_AMD_EINVAL, "halo_stage_in(level %d, dir %d) does not close the exchange halo_stage_out opened (level %d, dir %d)",
                 ilevel, dir, R.halo_level, R.halo_dir);
  R.halo_level = 0; R.halo_dir = -1;
  const size_t nr = (size_t)R.f_recv_off[L->ncpu];
  if (nr > 0) HCHK(hipMemcpyAsync(R.recvbuf.p, R.h_recv.p, sizeof(double) * nr, hipMemcpyHostToDevice, nullptr), "H2D halo");
  return halo_unpack(R, *L, S);
}

}  // extern "C"

#include "warm.hpp"
RAMSES_AMD_TU_WARM(capi_amr)
