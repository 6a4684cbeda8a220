// cg_amr.hip -- conjugate-gradient Poisson solver on one AMR level (gfx950).
//
// Restates on the device the iteration loop of phi_fine_cg (poisson/phi_fine_cg.f90:88-187)
// and cmp_Ap_cg (:344-447): unpreconditioned CG on A = I - (1/6) sum of the six neighbours,
// zero outside the level.  The vectors stay in the reference's own cell layout (x = phi,
// r = f(:,1), p = f(:,2), z = f(:,3)), so neighbour cells in octs that are not in the level's
// list (other ranks' octs, physical-boundary octs) are read where the reference reads them.
// One thread per oct: 6 neighbour octs from a table built once per solve, 8 cells each.
// Three launches per iteration: `update_p_kernel` (p = r + beta*p), `ap_kernel` (z = A p and the
// products p*z), `update_xr_kernel` (recurrences on x and r, products r*r).  In the last two the
// block that finishes last adds the per-block partial sums in a fixed order (deterministic), and
// `update_xr_kernel`'s stores r2 into pinned host memory, which is all the host loop reads.
// (Forming p of the neighbour cells inside `ap_kernel` instead -- two launches -- was measured
// slower: 45 us against 9 + 19 us at 2 M cells; the kernels are bound by load issue, not by HBM.)
//
// Every cell update is the reference's expression, operation for operation (this unit is
// compiled with -ffp-contract=off).  The three dot products of an iteration are either
//   * ordered (default): products written in the reference's summation order (ind outermost, then the
//     oct list) and summed SEQUENTIALLY IN PARALLEL by the scan of parity functions of parity_scan.hpp
//     (signed terms; exact integer arithmetic inside a binade, IEEE adds where the sum crosses one) --
//     bit-identical to the reference; as a check of the scan a single lane can add them one after
//     the other instead (a dependent chain of N adds: ~650 ms per sum at 16.8 M cells); or
//   * parallel (RAMSES_AMD_CG_ORDERED=0): a fixed reduction tree (per oct, per block, over the blocks)
//     -- deterministic, equal to the ordered sum to rounding only, and CG amplifies that: phi agrees
//     with the reference to ~1e-9..1e-13 relative depending on the iteration count.
// Per iteration and cell, r, p (x2), z (x2), x are read and p, z, x, r written = 80 B algorithmic
// (+3 B of neighbour table).
#include "cg_amr_args.hpp"
#include "parity_scan.hpp"

namespace ramses_amd {
namespace {

constexpr int TPB = 256;

// neighbour tables of poisson/phi_fine_cg.f90:370-375: oct (0 own, k = k-th neighbour) and octant
// (0-based here) of the left/right neighbour of octant ind in direction idim
__device__ __constant__ int c_iii[3][2][8] = {{{1, 0, 1, 0, 1, 0, 1, 0}, {0, 2, 0, 2, 0, 2, 0, 2}},
                                               {{3, 3, 0, 0, 3, 3, 0, 0}, {0, 0, 4, 4, 0, 0, 4, 4}},
                                               {{5, 5, 5, 5, 0, 0, 0, 0}, {0, 0, 0, 0, 6, 6, 6, 6}}};
__device__ __constant__ int c_jjj[3][2][8] = {{{1, 0, 3, 2, 5, 4, 7, 6}, {1, 0, 3, 2, 5, 4, 7, 6}},
                                               {{2, 3, 0, 1, 6, 7, 4, 5}, {2, 3, 0, 1, 6, 7, 4, 5}},
                                               {{4, 5, 6, 7, 0, 1, 2, 3}, {4, 5, 6, 7, 0, 1, 2, 3}}};

__global__ void setup_kernel(const int *igrid, int ngrid, const int *son, const int *nbor, long ngridmax, int *nb) {
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < 6L * ngrid; t += (long)gridDim.x * blockDim.x) {
    const int k = (int)(t / ngrid), i = (int)(t % ngrid);
    const int cell = nbor[(long)k * ngridmax + igrid[i] - 1];
    const int g = cell > 0 ? son[cell - 1] : 0;
    nb[t] = g > 0 ? g : 0;
  }
}

// block-wide sum with a fixed tree; thread 0 writes partial[blockIdx.x].  With FINISH the block
// that is last to arrive adds the partials of all blocks (fixed order: deterministic whatever the
// arrival order) and stores the total into scal[slot] (slot CG_R2 first moves the old value to
// CG_R2_OLD) and, if host_slot >= 0, into the pinned ring the host loop reads.
template <bool FINISH>
__device__ __forceinline__ void block_partial(double v, const CgLevel &L, int slot, int host_slot) {
  __shared__ double sh[TPB];
  __shared__ bool last;
  sh[threadIdx.x] = v;
  __syncthreads();
  for (int s = TPB / 2; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) sh[threadIdx.x] = sh[threadIdx.x] + sh[threadIdx.x + s];
    __syncthreads();
  }
  if (!FINISH) {
    if (threadIdx.x == 0) L.partial[blockIdx.x] = sh[0];
    return;
  }
  // The partial and the arrival count travel as device-scope atomics (performed at the
  // coherence point, no cache-wide release/acquire fences): the exchange has returned before
  // the count is bumped, so whoever sees the full count can read every partial.
  unsigned long long *part = reinterpret_cast<unsigned long long *>(L.partial);
  unsigned *count = reinterpret_cast<unsigned *>(L.scal + 6);
  if (threadIdx.x == 0) {
    const unsigned long long prev = __hip_atomic_exchange(&part[blockIdx.x], (unsigned long long)__double_as_longlong(sh[0]),
                                                          __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_s_waitcnt(0x0f70);   // vmcnt(0): the returning atomic is done
    asm volatile("" ::"v"(prev));
    last = __hip_atomic_fetch_add(count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1;
  }
  __syncthreads();
  if (!last) return;
  double t = 0.0;
  for (int j = threadIdx.x; j < (int)gridDim.x; j += TPB)
    t = t + __longlong_as_double((long long)__hip_atomic_load(&part[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
  sh[threadIdx.x] = t;
  __syncthreads();
  for (int s = TPB / 2; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) sh[threadIdx.x] = sh[threadIdx.x] + sh[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    if (slot == CG_R2) L.scal[CG_R2_OLD] = L.scal[CG_R2];
    L.scal[slot] = sh[0];
    if (host_slot >= 0) L.host_r2[host_slot] = sh[0];
    __hip_atomic_store(count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// sum of the per-block partials (fixed tree) or of the ordered products (one lane, one add after
// the other), stored into scal[slot]; slot CG_R2 first moves the old value to CG_R2_OLD
__global__ __launch_bounds__(TPB) void final_kernel(const double *partial, int nblocks, const double *prod, long nprod,
                                                     double *scal, int slot, double *host_r2, int host_slot) {
  __shared__ double sh[TPB];
  double total;
  if (prod) {
    // ordered: lanes fetch 64 consecutive products, lane 0's chain adds them in order
    double acc = 0.0;
    if (threadIdx.x < 64) {
      for (long base = 0; base < nprod; base += 64) {
        const long j = base + threadIdx.x;
        const double v = j < nprod ? prod[j] : 0.0;
        const int m = (int)((nprod - base) < 64 ? (nprod - base) : 64);
        if (m == 64) {
#pragma unroll
          for (int k = 0; k < 64; k++) acc = acc + __shfl(v, k, 64);
        } else {
          for (int k = 0; k < m; k++) acc = acc + __shfl(v, k, 64);
        }
      }
    }
    sh[threadIdx.x] = acc;
    __syncthreads();
    total = sh[0];
  } else {
    double v = 0.0;
    for (int j = threadIdx.x; j < nblocks; j += TPB) v = v + partial[j];
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int s = TPB / 2; s > 0; s >>= 1) {
      if ((int)threadIdx.x < s) sh[threadIdx.x] = sh[threadIdx.x] + sh[threadIdx.x + s];
      __syncthreads();
    }
    total = sh[0];
  }
  if (threadIdx.x == 0) {
    if (slot == CG_R2) scal[CG_R2_OLD] = scal[CG_R2];
    scal[slot] = total;
    if (host_slot >= 0) host_r2[host_slot] = total;
  }
}

// rhs_norm (:63-70): fact2*(rho-rho_tot)*(rho-rho_tot)
template <bool FINISH>
__global__ __launch_bounds__(TPB) void rhs_kernel(CgLevel L, const double *rho, double rho_tot, double fact2) {
  double acc = 0.0;
  for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < L.ngrid; i += (long)gridDim.x * TPB) {
    const long g = L.igrid[i] - 1;
#pragma unroll
    for (int ind = 0; ind < 8; ind++) {
      const double d = rho[L.ncoarse + ind * L.ngridmax + g] - rho_tot;
      const double v = fact2 * d * d;
      if (!FINISH) L.prod[(long)ind * L.ngrid + i] = v;
      acc = acc + v;
    }
  }
  block_partial<FINISH>(acc, L, CG_RHS, -1);
}

// r.r (:98-105)
template <bool FINISH>
__global__ __launch_bounds__(TPB) void dot_rr_kernel(CgLevel L, int host_slot) {
  double acc = 0.0;
  for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < L.ngrid; i += (long)gridDim.x * TPB) {
    const long g = L.igrid[i] - 1;
#pragma unroll
    for (int ind = 0; ind < 8; ind++) {
      const double r = L.r[L.ncoarse + ind * L.ngridmax + g];
      const double v = r * r;
      if (!FINISH) L.prod[(long)ind * L.ngrid + i] = v;
      acc = acc + v;
    }
  }
  block_partial<FINISH>(acc, L, CG_R2, host_slot);
}

// recurrence on p (:116-133): p = r + beta*p, beta = 0 in the first iteration, else r2/r2_old
__global__ __launch_bounds__(TPB) void update_p_kernel(CgLevel L, int iter) {
  const double beta = iter == 1 ? 0.0 : L.scal[CG_R2] / L.scal[CG_R2_OLD];
  for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < L.ngrid; i += (long)gridDim.x * TPB) {
    const long g = L.igrid[i] - 1;
#pragma unroll
    for (int ind = 0; ind < 8; ind++) {
      const long c = L.ncoarse + ind * L.ngridmax + g;
      L.p[c] = L.r[c] + beta * L.p[c];
    }
  }
}

// z = A p (cmp_Ap_cg :344-447) and the products p*z (:146-153)
template <bool FINISH>
__global__ __launch_bounds__(TPB) void ap_kernel(CgLevel L) {
  const double oneoversix = 1.0 / 6.0;
  double acc = 0.0;
  for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < L.ngrid; i += (long)gridDim.x * TPB) {
    long gn[7];
    gn[0] = L.igrid[i];
#pragma unroll
    for (int k = 0; k < 6; k++) gn[k + 1] = L.nb[(long)k * L.ngrid + i];
    double own[8];
#pragma unroll
    for (int ind = 0; ind < 8; ind++) own[ind] = L.p[L.ncoarse + ind * L.ngridmax + gn[0] - 1];
#pragma unroll
    for (int ind = 0; ind < 8; ind++) {
      double res = -own[ind];
#pragma unroll
      for (int idim = 0; idim < 3; idim++) {
        double pg, pd;
        {
          const int ig = c_iii[idim][0][ind], id = c_jjj[idim][0][ind];
          pg = ig == 0 ? own[id] : (gn[ig] > 0 ? L.p[L.ncoarse + id * L.ngridmax + gn[ig] - 1] : 0.0);
        }
        {
          const int ig = c_iii[idim][1][ind], id = c_jjj[idim][1][ind];
          pd = ig == 0 ? own[id] : (gn[ig] > 0 ? L.p[L.ncoarse + id * L.ngridmax + gn[ig] - 1] : 0.0);
        }
        res = res + oneoversix * (pg + pd);
      }
      L.z[L.ncoarse + ind * L.ngridmax + gn[0] - 1] = res;
      const double v = own[ind] * res;
      if (!FINISH) L.prod[(long)ind * L.ngrid + i] = v;
      acc = acc + v;
    }
  }
  block_partial<FINISH>(acc, L, CG_PAP, -1);
}

// recurrences on x and r (:160-183) and the products r*r of the next iteration (:98-105)
template <bool FINISH>
__global__ __launch_bounds__(TPB) void update_xr_kernel(CgLevel L, int host_slot) {
  const double alpha = L.scal[CG_R2] / L.scal[CG_PAP];
  double acc = 0.0;
  for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < L.ngrid; i += (long)gridDim.x * TPB) {
    const long g = L.igrid[i] - 1;
#pragma unroll
    for (int ind = 0; ind < 8; ind++) {
      const long c = L.ncoarse + ind * L.ngridmax + g;
      const double p = L.p[c];
      L.x[c] = L.x[c] + alpha * p;
      const double r = L.r[c] - alpha * L.z[c];
      L.r[c] = r;
      const double v = r * r;
      if (!FINISH) L.prod[(long)ind * L.ngrid + i] = v;
      acc = acc + v;
    }
  }
  block_partial<FINISH>(acc, L, CG_R2, host_slot);
}

inline int blocks_for(long work) {
  long b = (work + TPB - 1) / TPB;
  if (b < 1) b = 1;
  if (b > CG_MAX_BLOCKS) b = CG_MAX_BLOCKS;
  return (int)b;
}
// the ordered products as a source of the parity scan, and what happens to the sum (final_kernel's epilogue)
struct ProdSrc {
  const double *prod;
  long n;
  __device__ long count() const { return n; }
  __device__ double operator()(int, long p) const { return prod[p]; }
};
struct ScalFin {
  double *scal;
  int slot;
  double *host_r2;
  int host_slot;
  __device__ void operator()(int, double total) const {
    if (slot == CG_R2) scal[CG_R2_OLD] = scal[CG_R2];
    scal[slot] = total;
    if (host_slot >= 0) host_r2[host_slot] = total;
  }
};
inline void launch_final(const CgLevel &L, int nb, int slot, int host_slot, hipStream_t s) {
  if (L.prod && L.scan) {
    ProdSrc S{L.prod, 8L * L.ngrid};
    ScalFin F{L.scal, slot, L.host_r2, host_slot};
    (void)pscan::launch<ProdSrc, 1, ScalFin>(S, S.n, L.scal + 5, L.scan, s, F);
    return;
  }
  hipLaunchKernelGGL(final_kernel, dim3(1), dim3(TPB), 0, s, (const double *)L.partial, nb, (const double *)L.prod,
                     8L * L.ngrid, L.scal, slot, L.host_r2, host_slot);
}

}  // namespace

size_t cg_scan_bytes(int ngrid) { return pscan::scratch_bytes(8L * (ngrid > 0 ? ngrid : 1), 1); }
size_t ordered_sum_bytes(long n) { return pscan::scratch_bytes(n > 0 ? n : 1, 1); }
hipError_t ordered_sum_launch(const double *x, long n, double *out, void *scratch, hipStream_t s) {
  ProdSrc S{x, n};
  return pscan::launch<ProdSrc, 1>(S, n, out, scratch, s);
}

hipError_t cg_launch_setup(const int *igrid, int ngrid, const int *son, const int *nbor, long ngridmax, int *nb, hipStream_t s) {
  if (ngrid <= 0) return hipSuccess;
  hipLaunchKernelGGL(setup_kernel, dim3(blocks_for(6L * ngrid)), dim3(TPB), 0, s, igrid, ngrid, son, nbor, ngridmax, nb);
  return hipGetLastError();
}
hipError_t cg_launch_rhs_norm(const CgLevel &L, const double *rho, double rho_tot, double fact2, hipStream_t s) {
  const int nb = blocks_for(L.ngrid);
  if (L.prod) {
    hipLaunchKernelGGL(rhs_kernel<false>, dim3(nb), dim3(TPB), 0, s, L, rho, rho_tot, fact2);
    launch_final(L, nb, CG_RHS, -1, s);
  } else {
    hipLaunchKernelGGL(rhs_kernel<true>, dim3(nb), dim3(TPB), 0, s, L, rho, rho_tot, fact2);
  }
  return hipGetLastError();
}
hipError_t cg_launch_dot_rr(const CgLevel &L, int slot, hipStream_t s) {
  const int nb = blocks_for(L.ngrid);
  if (L.prod) {
    hipLaunchKernelGGL(dot_rr_kernel<false>, dim3(nb), dim3(TPB), 0, s, L, -1);
    launch_final(L, nb, CG_R2, slot, s);
  } else {
    hipLaunchKernelGGL(dot_rr_kernel<true>, dim3(nb), dim3(TPB), 0, s, L, slot);
  }
  return hipGetLastError();
}
hipError_t cg_launch_iteration(const CgLevel &L, int iter, int slot, hipStream_t s) {
  const int nb = blocks_for(L.ngrid);
  hipLaunchKernelGGL(update_p_kernel, dim3(nb), dim3(TPB), 0, s, L, iter);
  if (L.prod) {
    hipLaunchKernelGGL(ap_kernel<false>, dim3(nb), dim3(TPB), 0, s, L);
    launch_final(L, nb, CG_PAP, -1, s);
    hipLaunchKernelGGL(update_xr_kernel<false>, dim3(nb), dim3(TPB), 0, s, L, -1);
    launch_final(L, nb, CG_R2, slot, s);
  } else {
    hipLaunchKernelGGL(ap_kernel<true>, dim3(nb), dim3(TPB), 0, s, L);
    hipLaunchKernelGGL(update_xr_kernel<true>, dim3(nb), dim3(TPB), 0, s, L, slot);
  }
  return hipGetLastError();
}

// the three routines of an iteration one by one (several MPI ranks: a reduction over the ranks sits between them)
hipError_t cg_launch_update_p(const CgLevel &L, int iter, hipStream_t s) {
  hipLaunchKernelGGL(update_p_kernel, dim3(blocks_for(L.ngrid)), dim3(TPB), 0, s, L, iter);
  return hipGetLastError();
}
hipError_t cg_launch_ap(const CgLevel &L, hipStream_t s) {
  const int nb = blocks_for(L.ngrid);
  if (L.prod) {
    hipLaunchKernelGGL(ap_kernel<false>, dim3(nb), dim3(TPB), 0, s, L);
    launch_final(L, nb, CG_PAP, -1, s);
  } else {
    hipLaunchKernelGGL(ap_kernel<true>, dim3(nb), dim3(TPB), 0, s, L);
  }
  return hipGetLastError();
}
hipError_t cg_launch_update_xr(const CgLevel &L, hipStream_t s) {
  const int nb = blocks_for(L.ngrid);
  if (L.prod) {
    hipLaunchKernelGGL(update_xr_kernel<false>, dim3(nb), dim3(TPB), 0, s, L, -1);
    launch_final(L, nb, CG_R2, -1, s);
  } else {
    hipLaunchKernelGGL(update_xr_kernel<true>, dim3(nb), dim3(TPB), 0, s, L, -1);
  }
  return hipGetLastError();
}

}  // namespace ramses_amd

#include "warm.hpp"
RAMSES_AMD_TU_WARM(cg_amr)
