// mg_kernels.hip -- fine-level multigrid Poisson solver on a fully refined
// periodic level brick (reference: poisson/multigrid_fine_commons.f90,
// multigrid_fine_fine.f90, multigrid_fine_coarse.f90, force_fine.f90).
//
// All operators are 7-point FP64 stencils: HBM-bound, no MFMA.  Dense bricks
// phi[k][j][i] per multigrid level (n = 2^level) replace the reference's
// per-solve communicator construction (build_parent_comms_mg): on a fully
// refined level parent/child/neighbour indices are arithmetic.
//
// Bit parity: the neighbour sum (x-,y-,z-,x+,y+,z+), the child order of the
// restriction and the weight order of the prolongation are the reference's;
// compiled with -ffp-contract=off.  Red/black = parity of i+j+k, exactly the
// reference's octant sets (1,4,6,7)/(2,3,5,8).
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "mg_args.hpp"

namespace ramses_amd {

__device__ __forceinline__ int wrapi(int i, int n) { return i < 0 ? i + n : (i >= n ? i - n : i); }

// cell index -> (i,j,k) of an n^3 box, n a power of two (every multigrid level is): shifts and
// masks instead of 64-bit integer division, which costs more than the 7-point stencil itself
__device__ __forceinline__ int ilog2(int n) { return 31 - __clz(n); }
__device__ __forceinline__ void decode3(long c, int lg, int &i, int &j, int &k) {
  const int mask = (1 << lg) - 1;
  i = (int)(c & mask);
  j = (int)((c >> lg) & mask);
  k = (int)(c >> (2 * lg));
}

__device__ __forceinline__ double nb_sum6(const double *__restrict__ phi, int i, int j, int k, int n) {
  const long nn = (long)n * n;
  const long row = (long)j * n + (long)k * nn;
  double s = 0.0;
  s = s + phi[row + wrapi(i - 1, n)];
  s = s + phi[(long)wrapi(j - 1, n) * n + (long)k * nn + i];
  s = s + phi[(long)j * n + (long)wrapi(k - 1, n) * nn + i];
  s = s + phi[row + wrapi(i + 1, n)];
  s = s + phi[(long)wrapi(j + 1, n) * n + (long)k * nn + i];
  s = s + phi[(long)j * n + (long)wrapi(k + 1, n) * nn + i];
  return s;
}

// x/6 correctly rounded without the IEEE division sequence: q = RN(x*z),
// r = x - 6q (exact in an FMA), q' = RN(q + r*z) with z = RN(1/6) -- Markstein's
// division by a constant; checked against true division on 4e5 random and
// adversarial operands (and by the bit-parity tests on the GPU).
__device__ __forceinline__ double div6(double x) {
  const double z = 1.0 / 6.0;
  const double q = x * z;
  const double r = __builtin_fma(-6.0, q, x);
  return __builtin_fma(r, z, q);
}

// f2 = fourpi*(rho - rho_tot): make_fine_bc_rhs on an unmasked periodic level
__global__ __launch_bounds__(256) void mg_rhs_kernel(const double *__restrict__ rho, double *__restrict__ f2,
                                                      long N, double fourpi, double rho_tot) {
  for (long c = (long)blockIdx.x * blockDim.x + threadIdx.x; c < N; c += (long)gridDim.x * blockDim.x)
    f2[c] = fourpi * (rho[c] - rho_tot);
}

// one colour of red-black Gauss-Seidel (gauss_seidel_mg_fine/_coarse fast path)
__device__ __forceinline__ void mg_gs_body(double *__restrict__ phi, const double *__restrict__ rhs, int n, double dx2,
                                           int color, long first, long stride) {
  const int nh = n >> 1;
  const long total = (long)nh * n * n;
  for (long t = first; t < total; t += stride) {
    const int lg = ilog2(n);
    const int ih = (int)(t & (nh - 1));
    const int j = (int)((t >> (lg - 1)) & (n - 1));
    const int k = (int)(t >> (2 * lg - 1));
    const int i = 2 * ih + ((j + k + color) & 1);
    const double nb = nb_sum6(phi, i, j, k, n);
    const long c = (long)i + (long)n * (j + (long)n * k);
    phi[c] = div6(nb - dx2 * rhs[c]);
  }
}
__global__ __launch_bounds__(256) void mg_gs_kernel(double *__restrict__ phi, const double *__restrict__ rhs,
                                                     int n, double dx2, int color) {
  mg_gs_body(phi, rhs, n, dx2, color, (long)blockIdx.x * blockDim.x + threadIdx.x, (long)gridDim.x * blockDim.x);
}

// res = -(nb - 6 phi)/dx^2 + rhs ; optional per-block partial sums of res^2
__global__ __launch_bounds__(256) void mg_residual_kernel(const double *__restrict__ phi,
                                                           const double *__restrict__ rhs,
                                                           double *__restrict__ res, int n, double oneoverdx2,
                                                           double *__restrict__ partial) {
  const long N = (long)n * n * n;
  double acc = 0.0;
  for (long c = (long)blockIdx.x * blockDim.x + threadIdx.x; c < N; c += (long)gridDim.x * blockDim.x) {
    int i, j, k;
    decode3(c, ilog2(n), i, j, k);
    const double phi_c = phi[c];
    const double nb = nb_sum6(phi, i, j, k, n);
    const double r = -oneoverdx2 * (nb - 6.0 * phi_c) + rhs[c];
    res[c] = r;
    acc = acc + r * r;
  }
  if (partial) {
    // deterministic in-block tree (fixed order), one partial per block
    __shared__ double sm[256];
    sm[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
      if ((int)threadIdx.x < s) sm[threadIdx.x] = sm[threadIdx.x] + sm[threadIdx.x + s];
      __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = sm[0];
  }
}

// out[0] = scale * sum(partial[0..m)) in index order (deterministic)
__global__ void mg_sum_partials_kernel(const double *__restrict__ partial, int m, double scale,
                                       double *__restrict__ out) {
  __shared__ double sm[256];
  double acc = 0.0;
  for (int t = threadIdx.x; t < m; t += 256) acc = acc + partial[t];
  sm[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) sm[threadIdx.x] = sm[threadIdx.x] + sm[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = scale * sm[0];
}

// coarse rhs = sum over the 8 children (octant order) of res/8 ; coarse
// correction reset to zero in the same pass (multigrid_fine_commons.f90:217-238)
__device__ __forceinline__ void mg_restrict_body(const double *__restrict__ res_f, double *__restrict__ rhs_c,
                                                 double *__restrict__ u1_c, int nf, long first, long stride) {
  const int nc = nf >> 1;
  const long Nc = (long)nc * nc * nc;
  for (long c = first; c < Nc; c += stride) {
    int I, J, K;
    decode3(c, ilog2(nc), I, J, K);
    double acc = 0.0;
#pragma unroll
    for (int ind = 0; ind < 8; ind++) {
      const int ix = ind & 1, iy = (ind >> 1) & 1, iz = (ind >> 2) & 1;
      acc = acc + res_f[(long)(2 * I + ix) + (long)nf * ((2 * J + iy) + (long)nf * (2 * K + iz))] / 8.0;
    }
    rhs_c[c] = acc;
    u1_c[c] = 0.0;
  }
}
__global__ __launch_bounds__(256) void mg_restrict_kernel(const double *__restrict__ res_f,
                                                           double *__restrict__ rhs_c,
                                                           double *__restrict__ u1_c, int nf) {
  mg_restrict_body(res_f, rhs_c, u1_c, nf, (long)blockIdx.x * blockDim.x + threadIdx.x, (long)gridDim.x * blockDim.x);
}

// phi_f += sum_{8 of 27 parents} w*corr_c, weights (a,b,b,c,b,c,c,d)
// One thread per COARSE cell: its 3^3 neighbourhood of corrections is loaded once (27 loads for 8
// children instead of 8 per child) and each child adds its 8 terms in the reference's order
// (interpolate_and_correct_fine, multigrid_fine_fine.f90:596-698: t = 0..7, bit 0/1/2 of t set =
// the parent itself along x/y/z, clear = the neighbour on the child's side).
__device__ __forceinline__ void mg_interp_body(double *__restrict__ phi_f, const double *__restrict__ corr_c, int nf,
                                               long first, long stride) {
  const int nc = nf >> 1;
  const int lgc = ilog2(nc);
  const long Nc = (long)nc * nc * nc;
  const double a = 1.0 / 64.0, b = 3 * a, cc = 9 * a, d = 27 * a;
  const double bbb[8] = {a, b, b, cc, b, cc, cc, d};
  for (long c = first; c < Nc; c += stride) {
    int I, J, K;
    decode3(c, lgc, I, J, K);
    int xi[3] = {wrapi(I - 1, nc), I, wrapi(I + 1, nc)};
    long yo[3] = {(long)wrapi(J - 1, nc) * nc, (long)J * nc, (long)wrapi(J + 1, nc) * nc};
    long zo[3] = {(long)wrapi(K - 1, nc) * nc * nc, (long)K * nc * nc, (long)wrapi(K + 1, nc) * nc * nc};
    double v[3][3][3];
#pragma unroll
    for (int kz = 0; kz < 3; kz++)
#pragma unroll
      for (int ky = 0; ky < 3; ky++)
#pragma unroll
        for (int kx = 0; kx < 3; kx++) v[kz][ky][kx] = corr_c[zo[kz] + yo[ky] + xi[kx]];
#pragma unroll
    for (int iz = 0; iz < 2; iz++)
#pragma unroll
      for (int iy = 0; iy < 2; iy++) {
        const long row = (long)(2 * I) + (long)nf * ((2 * J + iy) + (long)nf * (2 * K + iz));
        double out[2];
#pragma unroll
        for (int ix = 0; ix < 2; ix++) {
          double corr = 0.0;
#pragma unroll
          for (int t = 0; t < 8; t++) {
            const int kx = (t & 1) ? 1 : (ix ? 2 : 0);
            const int ky = (t & 2) ? 1 : (iy ? 2 : 0);
            const int kz = (t & 4) ? 1 : (iz ? 2 : 0);
            corr = corr + bbb[t] * v[kz][ky][kx];
          }
          out[ix] = phi_f[row + ix] + corr;
        }
        phi_f[row] = out[0];
        phi_f[row + 1] = out[1];
      }
  }
}
__global__ __launch_bounds__(256) void mg_interp_kernel(double *__restrict__ phi_f,
                                                         const double *__restrict__ corr_c, int nf) {
  mg_interp_body(phi_f, corr_c, nf, (long)blockIdx.x * blockDim.x + threadIdx.x, (long)gridDim.x * blockDim.x);
}

// residual without the norm (the coarse levels' form): res = -(nb - 6 phi)/dx^2 + rhs
__device__ __forceinline__ void mg_residual_body(const double *__restrict__ phi, const double *__restrict__ rhs,
                                                 double *__restrict__ res, int n, double oneoverdx2, long first, long stride) {
  const long N = (long)n * n * n;
  for (long c = first; c < N; c += stride) {
    int i, j, k;
    decode3(c, ilog2(n), i, j, k);
    const double phi_c = phi[c];
    const double nb = nb_sum6(phi, i, j, k, n);
    res[c] = -oneoverdx2 * (nb - 6.0 * phi_c) + rhs[c];
  }
}

// recursive_multigrid_coarse (multigrid_fine_commons.f90:307-390) from level T.ltop down to level 1 and back up in ONE
// launch of ONE workgroup: at <= 32^3 a colour pass is a 4.6 us launch of a kernel that runs for well under a microsecond
// (profiles/r02_vcycle_levels.txt: ~80 launches = 0.41 ms of a 5.8 ms V-cycle at 512^3).  The same per-cell routines in the
// same order as the per-kernel schedule (every pass is order-independent inside: one colour, or one output per cell), a
// workgroup barrier where that schedule has a kernel boundary: bit-identical.
__global__ __launch_bounds__(1024) void mg_coarse_tail_kernel(MgTailArgs T) {
  const long t = threadIdx.x, nt = blockDim.x;
  for (int l = T.ltop; l >= 2; l--) {
    const int n = 1 << l;
    double *u1 = T.w + T.off[l][0], *u2 = T.w + T.off[l][1], *u3 = T.w + T.off[l][2];
    for (int i = 0; i < 2; i++) {
      mg_gs_body(u1, u2, n, T.dx2[l], 0, t, nt); __syncthreads();
      mg_gs_body(u1, u2, n, T.dx2[l], 1, t, nt); __syncthreads();
    }
    mg_residual_body(u1, u2, u3, n, T.oneoverdx2[l], t, nt); __syncthreads();
    mg_restrict_body(u3, T.w + T.off[l - 1][1], T.w + T.off[l - 1][0], n, t, nt); __syncthreads();
  }
  {
    double *u1 = T.w + T.off[1][0], *u2 = T.w + T.off[1][1];
    for (int i = 0; i < 4; i++) {
      mg_gs_body(u1, u2, 2, T.dx2[1], 0, t, nt); __syncthreads();
      mg_gs_body(u1, u2, 2, T.dx2[1], 1, t, nt); __syncthreads();
    }
  }
  for (int l = 2; l <= T.ltop; l++) {
    const int n = 1 << l;
    double *u1 = T.w + T.off[l][0], *u2 = T.w + T.off[l][1];
    mg_interp_body(u1, T.w + T.off[l - 1][0], n, t, nt); __syncthreads();
    for (int i = 0; i < 2; i++) {
      mg_gs_body(u1, u2, n, T.dx2[l], 0, t, nt); __syncthreads();
      mg_gs_body(u1, u2, n, T.dx2[l], 1, t, nt); __syncthreads();
    }
  }
}

// gradient_phi: f[d] = a(phi(-1)-phi(+1)) - b(phi(-2)-phi(+2))   (force_fine.f90:199-324)
__global__ __launch_bounds__(256) void mg_gradient_kernel(const double *__restrict__ phi, double *__restrict__ f,
                                                           int n, double a, double b) {
  const long N = (long)n * n * n;
  for (long c = (long)blockIdx.x * blockDim.x + threadIdx.x; c < N; c += (long)gridDim.x * blockDim.x) {
    int i, j, k;
    decode3(c, ilog2(n), i, j, k);
    const long nn = (long)n * n;
    {
      const long row = (long)j * n + (long)k * nn;
      f[c] = a * (phi[row + wrapi(i - 1, n)] - phi[row + wrapi(i + 1, n)]) -
             b * (phi[row + wrapi(i - 2, n)] - phi[row + wrapi(i + 2, n)]);
    }
    {
      const long o = (long)k * nn + i;
      f[c + N] = a * (phi[o + (long)wrapi(j - 1, n) * n] - phi[o + (long)wrapi(j + 1, n) * n]) -
                 b * (phi[o + (long)wrapi(j - 2, n) * n] - phi[o + (long)wrapi(j + 2, n) * n]);
    }
    {
      const long o = (long)j * n + i;
      f[c + 2 * N] = a * (phi[o + (long)wrapi(k - 1, n) * nn] - phi[o + (long)wrapi(k + 1, n) * nn]) -
                     b * (phi[o + (long)wrapi(k - 2, n) * nn] - phi[o + (long)wrapi(k + 2, n) * nn]);
    }
  }
}

// ---------------------------------------------------------------------------
// Distributed levels: one rank's n^3 brick of a periodic level with ng ghost
// layers (pitch n+2ng; ghosts filled by the halo exchange).  Same arithmetic
// and operation order as the dense kernels above.
// ---------------------------------------------------------------------------
// a rank's brick of a distributed level: nx x ny x nz cells (each a power of two; bricks of 2 or 4 ranks in a cubic box
// are not cubes) inside ng ghost layers
struct BrickDims {
  int nx, ny, nz;
};
__device__ __forceinline__ long gidx(int i, int j, int k, int ng, int px, int py) {
  return (long)(i + ng) + (long)px * ((j + ng) + (long)py * (k + ng));
}
__device__ __forceinline__ void decode3b(long c, int lgx, int lgy, int &i, int &j, int &k) {
  i = (int)(c & ((1 << lgx) - 1));
  j = (int)((c >> lgx) & ((1 << lgy) - 1));
  k = (int)(c >> (lgx + lgy));
}

// restriction of the residual of the local fine brick into the local coarse brick
__global__ __launch_bounds__(256) void mg_restrict_ghost_kernel(const double *__restrict__ res_f,
                                                                 double *__restrict__ rhs_c, BrickDims F, int ngf,
                                                                 int ngc) {
  const int ncx = F.nx >> 1, ncy = F.ny >> 1, ncz = F.nz >> 1;
  const long Nc = (long)ncx * ncy * ncz;
  const int pfx = F.nx + 2 * ngf, pfy = F.ny + 2 * ngf, pcx = ncx + 2 * ngc, pcy = ncy + 2 * ngc;
  const int lgx = ilog2(ncx), lgy = ilog2(ncy);
  for (long c = (long)blockIdx.x * blockDim.x + threadIdx.x; c < Nc; c += (long)gridDim.x * blockDim.x) {
    int I, J, K;
    decode3b(c, lgx, lgy, I, J, K);
    double acc = 0.0;
#pragma unroll
    for (int ind = 0; ind < 8; ind++) {
      const int ix = ind & 1, iy = (ind >> 1) & 1, iz = (ind >> 2) & 1;
      acc = acc + res_f[gidx(2 * I + ix, 2 * J + iy, 2 * K + iz, ngf, pfx, pfy)] / 8.0;
    }
    rhs_c[gidx(I, J, K, ngc, pcx, pcy)] = acc;
  }
}

// prolongation + correction of the local fine brick.  The coarse correction is
// either the local coarse brick with >= 1 valid ghost layer (cglob = 0) or a
// replicated dense periodic level of cglob^3 cells, of which this rank's part
// starts at (cox, coy, coz).
__global__ __launch_bounds__(256) void mg_interp_ghost_kernel(double *__restrict__ phi_f, BrickDims F, int ngf,
                                                               const double *__restrict__ corr_c, int ngc,
                                                               int cglob, int cox, int coy, int coz) {
  // one thread per coarse cell, as mg_interp_kernel
  const int ncx = F.nx >> 1, ncy = F.ny >> 1, ncz = F.nz >> 1;
  const int lgx = ilog2(ncx), lgy = ilog2(ncy);
  const long Nc = (long)ncx * ncy * ncz;
  const int pfx = F.nx + 2 * ngf, pfy = F.ny + 2 * ngf, pcx = ncx + 2 * ngc, pcy = ncy + 2 * ngc;
  const double a = 1.0 / 64.0, b = 3 * a, cc = 9 * a, d = 27 * a;
  const double bbb[8] = {a, b, b, cc, b, cc, cc, d};
  for (long c = (long)blockIdx.x * blockDim.x + threadIdx.x; c < Nc; c += (long)gridDim.x * blockDim.x) {
    int I, J, K;
    decode3b(c, lgx, lgy, I, J, K);
    double v[3][3][3];
#pragma unroll
    for (int kz = 0; kz < 3; kz++)
#pragma unroll
      for (int ky = 0; ky < 3; ky++)
#pragma unroll
        for (int kx = 0; kx < 3; kx++) {
          const int pi = I - 1 + kx, pj = J - 1 + ky, pk = K - 1 + kz;
          if (cglob) {
            v[kz][ky][kx] = corr_c[(long)wrapi(pi + cox, cglob) +
                                   (long)cglob * (wrapi(pj + coy, cglob) + (long)cglob * wrapi(pk + coz, cglob))];
          } else {
            v[kz][ky][kx] = corr_c[gidx(pi, pj, pk, ngc, pcx, pcy)];
          }
        }
#pragma unroll
    for (int iz = 0; iz < 2; iz++)
#pragma unroll
      for (int iy = 0; iy < 2; iy++) {
        const long row = gidx(2 * I, 2 * J + iy, 2 * K + iz, ngf, pfx, pfy);
        double out[2];
#pragma unroll
        for (int ix = 0; ix < 2; ix++) {
          double corr = 0.0;
#pragma unroll
          for (int t = 0; t < 8; t++) {
            const int kx = (t & 1) ? 1 : (ix ? 2 : 0);
            const int ky = (t & 2) ? 1 : (iy ? 2 : 0);
            const int kz = (t & 4) ? 1 : (iz ? 2 : 0);
            corr = corr + bbb[t] * v[kz][ky][kx];
          }
          out[ix] = phi_f[row + ix] + corr;
        }
        phi_f[row] = out[0];
        phi_f[row + 1] = out[1];
      }
  }
}

// gradient_phi on the local brick (needs 2 valid ghost layers of phi); f is a
// dense [3][nz][ny][nx] array
__global__ __launch_bounds__(256) void mg_gradient_ghost_kernel(const double *__restrict__ phi,
                                                                 double *__restrict__ f, BrickDims B, int ng, double a,
                                                                 double b) {
  const long N = (long)B.nx * B.ny * B.nz;
  const int p = B.nx + 2 * ng, py = B.ny + 2 * ng;
  const long pp = (long)p * py;
  const int lgx = ilog2(B.nx), lgy = ilog2(B.ny);
  for (long c = (long)blockIdx.x * blockDim.x + threadIdx.x; c < N; c += (long)gridDim.x * blockDim.x) {
    int i, j, k;
    decode3b(c, lgx, lgy, i, j, k);
    const long o = gidx(i, j, k, ng, p, py);
    f[c] = a * (phi[o - 1] - phi[o + 1]) - b * (phi[o - 2] - phi[o + 2]);
    f[c + N] = a * (phi[o - p] - phi[o + p]) - b * (phi[o - 2 * p] - phi[o + 2 * p]);
    f[c + 2 * N] = a * (phi[o - pp] - phi[o + pp]) - b * (phi[o - 2 * pp] - phi[o + 2 * pp]);
  }
}

// ---------------------------------------------------------------------------
// launch helpers
// ---------------------------------------------------------------------------
static inline int grid_for(long work, int cap = 4096) {
  long g = (work + 255) / 256;
  if (g < 1) g = 1;
  if (g > cap) g = cap;
  return (int)g;
}

hipError_t mg_launch_rhs(const double *rho, double *f2, long N, double fourpi, double rho_tot, hipStream_t s) {
  hipLaunchKernelGGL(mg_rhs_kernel, dim3(grid_for(N)), dim3(256), 0, s, rho, f2, N, fourpi, rho_tot);
  return hipGetLastError();
}
hipError_t mg_launch_gs(double *phi, const double *rhs, int n, double dx2, int color, hipStream_t s) {
  const long total = (long)(n >> 1) * n * n;
  hipLaunchKernelGGL(mg_gs_kernel, dim3(grid_for(total, 8192)), dim3(256), 0, s, phi, rhs, n, dx2, color);
  return hipGetLastError();
}
int mg_residual_blocks(int n) { return grid_for((long)n * n * n, MG_RESIDUAL_BLOCKS); }
hipError_t mg_launch_residual(const double *phi, const double *rhs, double *res, int n, double dx,
                              double *partial, double *norm_out, hipStream_t s) {
  const int blocks = mg_residual_blocks(n);
  const double oneoverdx2 = 1.0 / (dx * dx);
  hipLaunchKernelGGL(mg_residual_kernel, dim3(blocks), dim3(256), 0, s, phi, rhs, res, n, oneoverdx2,
                     norm_out ? partial : (double *)nullptr);
  if (norm_out)
    hipLaunchKernelGGL(mg_sum_partials_kernel, dim3(1), dim3(256), 0, s, partial, blocks, dx * dx * dx, norm_out);
  return hipGetLastError();
}
hipError_t mg_launch_restrict(const double *res_f, double *rhs_c, double *u1_c, int nf, hipStream_t s) {
  const long Nc = (long)(nf >> 1) * (nf >> 1) * (nf >> 1);
  hipLaunchKernelGGL(mg_restrict_kernel, dim3(grid_for(Nc)), dim3(256), 0, s, res_f, rhs_c, u1_c, nf);
  return hipGetLastError();
}
hipError_t mg_launch_interp(double *phi_f, const double *corr_c, int nf, hipStream_t s) {
  const int nc = nf >> 1;
  hipLaunchKernelGGL(mg_interp_kernel, dim3(grid_for((long)nc * nc * nc, 8192)), dim3(256), 0, s, phi_f, corr_c, nf);
  return hipGetLastError();
}
hipError_t mg_launch_coarse_tail(const MgTailArgs &T, hipStream_t s) {
  hipLaunchKernelGGL(mg_coarse_tail_kernel, dim3(1), dim3(1024), 0, s, T);
  return hipGetLastError();
}
hipError_t mg_launch_restrict_ghost(const double *res_f, double *rhs_c, int nfx, int nfy, int nfz, int ngf, int ngc, hipStream_t s) {
  const long Nc = (long)(nfx >> 1) * (nfy >> 1) * (nfz >> 1);
  hipLaunchKernelGGL(mg_restrict_ghost_kernel, dim3(grid_for(Nc)), dim3(256), 0, s, res_f, rhs_c, BrickDims{nfx, nfy, nfz}, ngf, ngc);
  return hipGetLastError();
}
hipError_t mg_launch_interp_ghost(double *phi_f, int nfx, int nfy, int nfz, int ngf, const double *corr_c, int ngc, int cglob,
                                  int cox, int coy, int coz, hipStream_t s) {
  const long Nc = (long)(nfx >> 1) * (nfy >> 1) * (nfz >> 1);
  hipLaunchKernelGGL(mg_interp_ghost_kernel, dim3(grid_for(Nc, 8192)), dim3(256), 0, s, phi_f, BrickDims{nfx, nfy, nfz},
                     ngf, corr_c, ngc, cglob, cox, coy, coz);
  return hipGetLastError();
}
hipError_t mg_launch_gradient_ghost(const double *phi, double *f, int nx, int ny, int nz, int ng, double a, double b, hipStream_t s) {
  hipLaunchKernelGGL(mg_gradient_ghost_kernel, dim3(grid_for((long)nx * ny * nz, 8192)), dim3(256), 0, s, phi, f,
                     BrickDims{nx, ny, nz}, ng, a, b);
  return hipGetLastError();
}
hipError_t mg_launch_gradient(const double *phi, double *f, int n, double a, double b, hipStream_t s) {
  hipLaunchKernelGGL(mg_gradient_kernel, dim3(grid_for((long)n * n * n, 8192)), dim3(256), 0, s, phi, f, n, a, b);
  return hipGetLastError();
}

// ===========================================================================
// Fused, time-skewed smoother: P colour passes (P/2 full red-black sweeps) and
// optionally the residual with its norm in ONE pass over the level.
//
// A workgroup owns an (LX-2P) x (LY-2P) column tile and marches along z.  The
// P colour passes run as pipeline stages lagging 2 planes each: at step m stage
// s applies pass s to plane m-2(s-1), reading planes z-1, z, z+1 that stage s-1
// finished one step earlier.  Planes live in an LDS ring of 2P+4 slots; every
// pass updates only cells whose whole dependency cone was loaded (the region
// shrinks by one cell per pass), so the values are exactly those of the global
// red-black sweeps -- bit-identical -- while phi is read ~1.7x and written once
// per TWO sweeps instead of 4 reads + 4 partial-line writes.  One barrier per
// step.  Within a step stage s writes only plane m-2(s-1) (cells of its colour)
// and reads that plane's other colour plus planes at odd offsets from m, which
// no stage writes in that step.  Out of place (phi_in -> phi_out): tile halos
// must see the values from before the sweeps while other workgroups store.
// ===========================================================================
// H = halo width: P for the smoother alone; P+1 when the residual is fused (it
// reads FINAL values one cell beyond the tile interior).
template <int P, bool RESID, int LYT = 24>
struct SmoothGeom {
  static constexpr int LX = 64, LY = LYT;
  static constexpr int H = RESID ? P + 1 : P;
  static constexpr int IX = LX - 2 * H, IY = LY - 2 * H;
  static constexpr int R = 2 * P + 4;
  static constexpr int PLANE = LX * LY;
};

// 4 tile rows per wavefront: 24 rows = 6 wavefronts (384 threads), 16 = 4, 12 = 3

// RESTR (with RESID, dense periodic levels): the residual does not leave the chip at all -- every plane of it is parked in
// LDS for one step, 377 threads add the four children of their coarse cell of that plane in octant order (the second
// plane of a pair continues the sum of the first, which waits in a register: the reference's order, child by child), and
// the coarse right-hand side is stored and the coarse correction zeroed: restrict_residual_fine_reverse + the reset of
// multigrid_fine_commons.f90:217-238 without a pass of their own (0.25 ms + the 1.07 GB residual store at 512^3).
//
// PROL (without RESID, dense periodic levels): the planes that enter the ring are phi + the trilinear interpolation of the
// coarse correction (interpolate_and_correct_fine, multigrid_fine_fine.f90:596-698: 8 of the 27 parents, weights
// 1,3,3,9,3,9,9,27 / 64 in the reference's order) -- the prolongation without a pass of its own (0.51 ms at 512^3: phi read
// and written once more).  The (LX/2 + 2) x (LY/2 + 2) coarse cells under a tile plane wait in an LDS ring of three coarse
// planes (K - 1, K, K + 1 around the parent plane); a new one is fetched two steps ahead of the step that needs it.
template <int P, bool RESID, int LYT = 24, bool RESTR = false, bool PROL = false>
__global__ __launch_bounds__(LYT * 16) void mg_smooth_fused_kernel(const double *__restrict__ phi_in,
                                                                          double *__restrict__ phi_out,
                                                                          const double *__restrict__ rhs,
                                                                          double *__restrict__ res,
                                                                          double *__restrict__ partial, int n,
                                                                          int ng, double dx2, double oneoverdx2,
                                                                          int zchunk, int ntx, int nty,
                                                                          double *__restrict__ rhs_c,
                                                                          double *__restrict__ u1_c,
                                                                          const double *__restrict__ corr_c,
                                                                          int ny, int nz) {
  // n = cells along x; ny, nz along y, z (a rank's brick of a distributed level need not be a cube; dense levels: all equal)
  using G = SmoothGeom<P, RESID, LYT>;
  static_assert(!PROL || (!RESID && P == 2 && (LYT * 16) * 2 >= (G::LX / 2 + 2) * (G::LY / 2 + 2)), "fused prolongation: the smoother without residual");
  static_assert(!RESTR || RESID, "the fused restriction restricts the fused residual");
  static_assert(!RESTR || ((G::IX % 2 == 0) && (G::IY % 2 == 0) && (G::IX / 2) * (G::IY / 2) <= LYT * 16), "coarse cells of a tile plane: one per thread");
  constexpr int H = G::H;
  constexpr int SMOOTH_THREADS = LYT * 16;
  constexpr int NW = SMOOTH_THREADS / 64;                  // 6 waves at 24 rows
  constexpr int NROW = G::LY / NW;                         // 4 full rows per wave (loads, final stage)
  constexpr int NPAIR = G::LY / (2 * NW);                  // 2 row pairs per wave (colour passes)
  static_assert(G::LY % (2 * NW) == 0, "tile rows must split evenly over the wavefronts");
  extern __shared__ __attribute__((aligned(16))) double ring[];  // [R][LY][LX]
  __shared__ double sm[512];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wv = tid >> 6;
  // XCD-aware: workgroup b runs on XCD b mod 8; every XCD is handed a contiguous run of tiles (x first, then y, then z-chunk), so
  // that the tiles whose halos overlap -- 36 % of what a 58 x 26 tile of 64 x 32 reads -- ask the SAME L2 (round 6; the norm's
  // partial sums stay indexed by tile: the same sum as before)
#ifndef MG_XCD_REMAP
#define MG_XCD_REMAP 1
#endif
  int bid = blockIdx.x;
  if (MG_XCD_REMAP) {
    const int nb = gridDim.x, x = bid & 7, q = nb >> 3, r = nb & 7;
    bid = x * q + (x < r ? x : r) + (bid >> 3);
  }
  const int tix = bid % ntx, tiy = (bid / ntx) % nty, tiz = bid / (ntx * nty);
  const int x0 = tix * G::IX - H, y0 = tiy * G::IY - H;   // global coords of tile cell (0,0)
  const int z0 = tiz * zchunk;
  const int z1 = min(z0 + zchunk, nz);                    // planes [z0,z1) are produced
  // ng = 0: dense periodic level, neighbours wrap; ng >= H: one rank's brick of a
  // distributed level with ghost layers filled by the halo exchange (addresses
  // outside the allocation are clamped: those values are never used)
  const int pitch = n + 2 * ng;
  const long nn = (long)pitch * (ny + 2 * ng);
  // every extent >= LX (the launcher guarantees it): one conditional wrap is enough
  auto wrapd = [&](int v, int nd) {
    if (ng == 0) return v < 0 ? v + nd : (v >= nd ? v - nd : v);
    return min(max(v, -ng), nd + ng - 1) + ng;
  };
  auto wrapx = [&](int v) { return wrapd(v, n); };
  auto wrapy = [&](int v) { return wrapd(v, ny); };
  auto wrapz = [&](int v) { return wrapd(v, nz); };
  auto slot = [&](int z) { int s = z % G::R; return s < 0 ? s + G::R : s; };

  // (a) row mapping: lane = x, wave wv owns rows wv + NW*i -- coalesced loads/stores
  const int gxu = x0 + lane;
  int goffR[NROW], lofsR[NROW];
#pragma unroll
  for (int i = 0; i < NROW; i++) {
    const int ly = wv + NW * i;
    goffR[i] = wrapy(y0 + ly) * pitch + wrapx(gxu);
    lofsR[i] = ly * G::LX + lane;
  }
  // (b) colour mapping: a wave owns row pairs; lanes 0-31 take the even row of the
  // pair, lanes 32-63 the odd row, each lane one (2p, 2p+1) cell pair: every lane
  // updates exactly one cell per colour pass (no half-masked wavefronts)
  const int pr = lane & 31, sub = lane >> 5;
  int lyC[NPAIR], goffC[NPAIR][2];
#pragma unroll
  for (int j = 0; j < NPAIR; j++) {
    lyC[j] = 2 * (wv + NW * j) + sub;
    const int gy = wrapy(y0 + lyC[j]) * pitch;
    goffC[j][0] = gy + wrapx(x0 + 2 * pr);
    goffC[j][1] = gy + wrapx(x0 + 2 * pr + 1);
  }
  double acc = 0.0;

  const int m_begin = z0 - H - 2;
  const int m_end = (z1 - 1) + 2 * P;

  // fused restriction: two residual planes [IY][IX] behind the ring (written in step m, read in step m + 1), this thread's
  // coarse cell and the sum of its first four children
  double *resbuf = ring + G::R * G::PLANE;
  constexpr int RB = G::IX * G::IY;
  const int cxl = tid % (G::IX / 2), cyl = tid / (G::IX / 2);
  const int cgx = x0 + H + 2 * cxl, cgy = y0 + H + 2 * cyl;
  const bool coarse_on = RESTR && (tid < (G::IX / 2) * (G::IY / 2)) && cgx < n && cgy < ny;
  double cacc = 0.0;
  auto restrict_plane = [&](int zr, int buf) {
    if (!coarse_on || zr < z0 || zr > z1 - 1) return;
    const double *rb = resbuf + buf * RB + (2 * cyl) * G::IX + 2 * cxl;
    if ((zr & 1) == 0) cacc = 0.0;
    cacc = cacc + rb[0] / 8.0;
    cacc = cacc + rb[1] / 8.0;
    cacc = cacc + rb[G::IX] / 8.0;
    cacc = cacc + rb[G::IX + 1] / 8.0;
    if (zr & 1) {
      const int nc = n >> 1;
      const long c = (long)(cgx >> 1) + (long)nc * ((cgy >> 1) + (long)nc * (zr >> 1));
      rhs_c[c] = cacc;
      u1_c[c] = 0.0;
    }
  };

  // All global reads of a step (the new phi plane, the rhs of the P colour
  // passes and of the residual plane) are issued one step AHEAD into registers:
  // every address is valid (wrapped), so the loads are unconditional and their
  // latency hides behind the previous step's stencil work.
  // rhs of colour pass s on plane z is the value pass s-2 used on the same plane 4 steps
  // earlier (same cell: the pair offset depends on z only through its parity), so only
  // passes 1 and 2 load; later passes take it from a register FIFO
  struct StepLoads { double ph[NROW]; double rv[2][NPAIR]; };
  // which cell of the pair has colour c on plane z in row ly: lx = 2p + off
  auto pair_off = [&](int ly, int z, int color) { return ((x0 + y0 + ly + z) & 1) ^ color; };
  auto issue = [&](int m, StepLoads &L) {
    {
      const double *__restrict__ base = phi_in + (long)wrapz(m + 2) * nn;
#pragma unroll
      for (int i = 0; i < NROW; i++) L.ph[i] = base[goffR[i]];
    }
#pragma unroll
    for (int s = 1; s <= 2; s++) {
      const int z = m - 2 * (s - 1);
      const double *__restrict__ base = rhs + (long)wrapz(z) * nn;
#pragma unroll
      for (int j = 0; j < NPAIR; j++) L.rv[s - 1][j] = base[goffC[j][pair_off(lyC[j], z, (s & 1) ? 0 : 1)]];
    }
    // (the residual of plane m-2P needs no load of its own: its red cells' rhs was loaded by pass 1
    //  2P steps ago, its black cells' by pass 2 2P-2 steps ago -- both are still in the register FIFO)
  };
  // prefetch distance 2 steps: LDS capacity limits the CU to 6 wavefronts, so
  // registers are plentiful and memory-level parallelism has to come from here
  StepLoads cur, nxt, nx2;
  // rhs values of passes 1,2 of the last FD steps; [FD-1] = previous step, [0] = FD steps ago.
  // Passes 3,4 reuse the values of 4 steps ago; the fused residual those of 2P (red) and 2P-2 (black)
  constexpr int FD = RESID ? 2 * P : 4;
  double fifo[FD][2][NPAIR];
#pragma unroll
  for (int a = 0; a < FD; a++)
#pragma unroll
    for (int b = 0; b < 2; b++)
#pragma unroll
      for (int j = 0; j < NPAIR; j++) fifo[a][b][j] = 0.0;
  // fused prolongation: the coarse planes under the tile
  constexpr int CW = G::LX / 2 + 2, CH = G::LY / 2 + 2, CPL = CW * CH;
  double *cring = ring + G::R * G::PLANE;            // [3][CH][CW]  (PROL and RESTR never come together)
  const int nc = n >> 1;
  double cpre[2] = {0.0, 0.0};
  auto cwrap = [&](int v) { return v < 0 ? v + nc : (v >= nc ? v - nc : v); };
  auto cslot = [&](int K) { int r = K % 3; return r < 0 ? r + 3 : r; };
  auto coarse_fetch = [&](int K, double (&dst)[2]) {   // this thread's (up to) two cells of coarse plane K
    const long zo = (long)cwrap(K) * nc * nc;
#pragma unroll
    for (int h = 0; h < 2; h++) {
      const int e = tid + h * SMOOTH_THREADS;
      if (e < CPL) dst[h] = corr_c[zo + (long)cwrap((y0 >> 1) - 1 + e / CW) * nc + cwrap((x0 >> 1) - 1 + e % CW)];
    }
  };
  auto coarse_put = [&](int K, const double (&src)[2]) {
    double *cp = cring + cslot(K) * CPL;
#pragma unroll
    for (int h = 0; h < 2; h++) {
      const int e = tid + h * SMOOTH_THREADS;
      if (e < CPL) cp[e] = src[h];
    }
  };
  if constexpr (PROL) {
    // the first plane that enters the ring is z0 - H (even): parents K0 = (z0 - H) / 2 and K0 - 1; K0 + 1 follows in the loop
    const int K0 = (z0 - H) >> 1;
    coarse_fetch(K0 - 1, cpre); coarse_put(K0 - 1, cpre);
    coarse_fetch(K0, cpre); coarse_put(K0, cpre);
    coarse_fetch(K0 + 1, cpre);
    __syncthreads();
  }
  issue(m_begin, cur);
  issue(m_begin + 1, nxt);
  for (int m = m_begin; m <= m_end; m++) {
    issue(m + 2, nx2);
    // the residual plane the previous step parked (plane m - 1 - 2P) joins its coarse cells
    if constexpr (RESTR) restrict_plane(m - 1 - 2 * P, (m + 1) & 1);
    const int zl = m + 2;
    const bool do_load = (zl >= z0 - H) && (zl <= z1 - 1 + H);
    // ---- gather phase: every LDS read of this step is issued before any LDS
    // write.  No stage reads a cell that another stage (or another lane of the
    // same stage) writes in the same step, so hoisting the reads is exact and
    // gives the scheduler independent stencil sums instead of a serial chain.
    double nbv[P][NPAIR];
    int cidx[P][NPAIR];
    bool on[P][NPAIR];
#pragma unroll
    for (int s = 1; s <= P; s++) {
      const int z = m - 2 * (s - 1);
      const bool zin = !(z < z0 - (H - s) || z > z1 - 1 + (H - s));
      const int color = (s & 1) ? 0 : 1;                  // odd passes red, even black
      const double *pc = ring + slot(z) * G::PLANE;
      const double *pm = ring + slot(z - 1) * G::PLANE;
      const double *pp = ring + slot(z + 1) * G::PLANE;
#pragma unroll
      for (int j = 0; j < NPAIR; j++) {
        const int ly = lyC[j];
        const int lx = 2 * pr + pair_off(ly, z, color);
        const int c = ly * G::LX + lx;
        cidx[s - 1][j] = c;
        on[s - 1][j] = zin && (lx >= s) && (lx < G::LX - s) && (ly >= s) && (ly < G::LY - s);
        double nb = 0.0;
        if (on[s - 1][j]) {
          nb = nb + pc[c - 1];
          nb = nb + pc[c - G::LX];
          nb = nb + pm[c];
          nb = nb + pc[c + 1];
          nb = nb + pc[c + G::LX];
          nb = nb + pp[c];
        }
        nbv[s - 1][j] = nb;
      }
    }
    // final stage.  Without the residual: row mapping (a), one phi per lane and row.  With it: pair
    // mapping (b), the lane's (2p, 2p+1) cells of its row pairs, because that is where the rhs sits.
    double nbf[NROW], phf[NROW];
    bool onf[NROW];
    double nbb[NPAIR][2], phb[NPAIR][2];
    bool onb[NPAIR][2];
    const int zf = m - 2 * P;
    {
      const bool zin = (zf >= z0) && (zf <= z1 - 1);
      const double *pc = ring + slot(zf) * G::PLANE;
      const double *pm = ring + slot(zf - 1) * G::PLANE;
      const double *pp = ring + slot(zf + 1) * G::PLANE;
      if (!RESID) {
        const bool xin = zin && (lane >= H) && (lane < G::LX - H) && (gxu < n);
#pragma unroll
        for (int i = 0; i < NROW; i++) {
          const int ly = wv + NW * i;
          onf[i] = xin && ly >= H && ly < G::LY - H && (y0 + ly) < ny;
          phf[i] = onf[i] ? pc[lofsR[i]] : 0.0;
          nbf[i] = 0.0;
        }
      } else {
#pragma unroll
        for (int j = 0; j < NPAIR; j++) {
          const int ly = lyC[j];
          const bool yin = zin && ly >= H && ly < G::LY - H && (y0 + ly) < ny;
#pragma unroll
          for (int e = 0; e < 2; e++) {
            const int lx = 2 * pr + e;
            onb[j][e] = yin && lx >= H && lx < G::LX - H && (x0 + lx) < n;
            double nb = 0.0, ph = 0.0;
            if (onb[j][e]) {
              const int c = ly * G::LX + lx;
              ph = pc[c];
              nb = nb + pc[c - 1];
              nb = nb + pc[c - G::LX];
              nb = nb + pm[c];
              nb = nb + pc[c + 1];
              nb = nb + pc[c + G::LX];
              nb = nb + pp[c];
            }
            nbb[j][e] = nb; phb[j][e] = ph;
          }
        }
      }
    }
    // ---- update phase: colour passes write their cells -------------------------
#pragma unroll
    for (int s = 1; s <= P; s++) {
      double *pc = ring + slot(m - 2 * (s - 1)) * G::PLANE;
#pragma unroll
      for (int j = 0; j < NPAIR; j++)
        if (on[s - 1][j]) {
          const double r = s <= 2 ? cur.rv[(s - 1) & 1][j] : fifo[FD - 4][(s - 1) & 1][j];
          pc[cidx[s - 1][j]] = div6(nbv[s - 1][j] - dx2 * r);
        }
    }
    // ---- final stage: store phi (+ residual and its norm) of plane m-2P -------
    {
      const long zoff = (long)wrapz(zf) * nn;
      if (!RESID) {
#pragma unroll
        for (int i = 0; i < NROW; i++)
          if (onf[i]) phi_out[zoff + goffR[i]] = phf[i];
      } else {
#pragma unroll
        for (int j = 0; j < NPAIR; j++) {
          const int ered = pair_off(lyC[j], zf, 0);       // which cell of the pair is red on this plane
#pragma unroll
          for (int e = 0; e < 2; e++) {
            if (onb[j][e]) {
              const long g = zoff + goffC[j][e];
              phi_out[g] = phb[j][e];
              const double rr = (e == ered) ? fifo[0][0][j] : fifo[2][1][j];
              const double r = -oneoverdx2 * (nbb[j][e] - 6.0 * phb[j][e]) + rr;
              if (res) res[g] = r;      // norm-only callers pass NULL: the residual never leaves the chip
              if constexpr (RESTR) resbuf[(m & 1) * RB + (lyC[j] - H) * G::IX + (2 * pr + e - H)] = r;
              acc = acc + r * r;
            }
          }
        }
      }
    }
    // ---- stage 0: plane m+2 (values from before the sweeps) into its ring slot --
    if (do_load) {
      double *pl = ring + slot(zl) * G::PLANE;
      if constexpr (PROL) {
        // phi + interpolated correction: parent (lane/2, ly/2, zl/2) of the coarse tile (which starts one cell earlier), the
        // neighbour on the child's side where bit t is clear; t = 0..7 in the reference's order
        const int K = zl >> 1, Kn = K + ((zl & 1) ? 1 : -1);
        const double *cK = cring + cslot(K) * CPL, *cN = cring + cslot(Kn) * CPL;
        const double wa = 1.0 / 64.0, wb = 3 * wa, wc = 9 * wa, wd = 27 * wa;
        const int pxl = (lane >> 1) + 1, nxl = pxl + ((lane & 1) ? 1 : -1);
#pragma unroll
        for (int i = 0; i < NROW; i++) {
          const int ly = wv + NW * i;
          const int pyl = (ly >> 1) + 1, nyl = pyl + ((ly & 1) ? 1 : -1);
          double corr = 0.0;
          corr = corr + wa * cN[nyl * CW + nxl];
          corr = corr + wb * cN[nyl * CW + pxl];
          corr = corr + wb * cN[pyl * CW + nxl];
          corr = corr + wc * cN[pyl * CW + pxl];
          corr = corr + wb * cK[nyl * CW + nxl];
          corr = corr + wc * cK[nyl * CW + pxl];
          corr = corr + wc * cK[pyl * CW + nxl];
          corr = corr + wd * cK[pyl * CW + pxl];
          pl[lofsR[i]] = cur.ph[i] + corr;
        }
        // the coarse plane the next odd fine plane needs: fetched two steps ago, parked now; the one after it is asked for
        if ((zl & 1) == 0) {
          coarse_put(K + 1, cpre);
          coarse_fetch(K + 2, cpre);
        }
      } else {
#pragma unroll
        for (int i = 0; i < NROW; i++) pl[lofsR[i]] = cur.ph[i];
      }
    }
    __syncthreads();
    if (P > 2 || RESID) {
#pragma unroll
      for (int b = 0; b < 2; b++)
#pragma unroll
        for (int j = 0; j < NPAIR; j++) {
#pragma unroll
          for (int a = 0; a + 1 < FD; a++) fifo[a][b][j] = fifo[a + 1][b][j];
          fifo[FD - 1][b][j] = cur.rv[b][j];
        }
    }
    cur = nxt;
    nxt = nx2;
  }
  if constexpr (RESTR) restrict_plane(m_end - 2 * P, m_end & 1);       // the last plane (the loop's last barrier is behind us)
  if (RESID && partial) {
    sm[tid] = acc;
    for (int i = SMOOTH_THREADS + tid; i < 512; i += SMOOTH_THREADS) sm[i] = 0.0;
    __syncthreads();
    // deterministic fixed-order tree over 512 slots (the threads' sums, zero padded)
    for (int s = 256; s > 0; s >>= 1) {
      if (tid < s) sm[tid] = sm[tid] + sm[tid + s];
      __syncthreads();
    }
    if (tid == 0) partial[bid] = sm[0];
  }
}

// tile rows of the fused smoother with P=2 (P=4 always uses 24: one 6-wave workgroup per CU).
// 32 (default): one 8-wave workgroup per CU, interior 58x26 of 64x32; 12/16: three or two
// workgroups per CU -- a tuning knob, results do not depend on it.  Measured at 512^3 per V-cycle:
// one 4-pass launch 7.75 ms, 2+2 passes on 24 rows 7.73, on 32 rows 7.27.
static int g_smooth_ly = 32;
void mg_set_smooth_rows(int ly) { g_smooth_ly = (ly == 12 || ly == 16 || ly == 32) ? ly : 24; }

// can the fused smoother of this level restrict its residual itself (mg_launch_smooth_fused with rhs_c / u1_c)?
bool mg_smooth_can_restrict(int n, int npass) { return npass == 2 && g_smooth_ly == 32 && n >= 64; }

hipError_t mg_launch_smooth_fused(const double *phi_in, double *phi_out, const double *rhs, double *res,
                                  double *partial, double *norm_out, int n, double dx, int npass,
                                  hipStream_t s, int ng, double *rhs_c, double *u1_c, const double *corr_c, int ny, int nz) {
  if (npass != 4 && npass != 2) return hipErrorInvalidValue;
  if (ny <= 0) ny = n;
  if (nz <= 0) nz = n;
  if ((ny != n || nz != n) && (ng == 0 || rhs_c || corr_c)) return hipErrorInvalidValue;   // dense periodic levels are cubes
  if (ny < 64 || nz < 64) return hipErrorInvalidValue;
  const bool restr = rhs_c != nullptr;
  const bool prol = corr_c != nullptr;
  if (prol && (restr || res || norm_out || ng != 0 || !mg_smooth_can_restrict(n, npass) || (n & 1))) return hipErrorInvalidValue;
  if (restr && (!u1_c || ng != 0 || !mg_smooth_can_restrict(n, npass) || (n & 1))) return hipErrorInvalidValue;
  if (n < 64) return hipErrorInvalidValue;   // tile wider than the level: use the per-colour kernels
  const int P = npass;
  const bool resid = restr || (res != nullptr) || (norm_out != nullptr);
  const int H = resid ? P + 1 : P;
  if (ng != 0 && ng < H) return hipErrorInvalidValue;   // ghost layers must cover the dependency cone
  const int LY = (P == 2) ? g_smooth_ly : 24;
  const int IX = 64 - 2 * H, IY = LY - 2 * H;
  const int ntx = (n + IX - 1) / IX, nty = (ny + IY - 1) / IY;
  int zchunk = nz >= 512 ? 128 : (nz >= 128 ? 64 : nz);   // (256^3: 64 planes per workgroup measured faster than 128, round 2)
  if (const char *e = getenv("RAMSES_AMD_MG_ZCHUNK")) { const int z = atoi(e); if (z >= 8 && nz >= 256) zchunk = z; }   // tuning aid
  if (const char *e = getenv("RAMSES_AMD_MG_ZCHUNK_SMALL")) { const int z = atoi(e); if (z >= 8 && nz < 256) zchunk = z < nz ? z : nz; }   // tuning aid
  const int ntz = (nz + zchunk - 1) / zchunk;
  const int blocks = ntx * nty * ntz;
  if (resid && blocks > MG_MAX_PARTIALS) return hipErrorInvalidValue;
  const size_t lds = sizeof(double) * ((size_t)(2 * P + 4) * 64 * LY + (restr ? 2 * (size_t)IX * IY : 0) + (prol ? 3 * (size_t)(64 / 2 + 2) * (LY / 2 + 2) : 0));
  const double dx2 = dx * dx, oneoverdx2 = 1.0 / (dx * dx);
  hipError_t e;
#define SM_LAUNCH(PP, RR, LL)                                                                                 \
  do {                                                                                                        \
    auto k = mg_smooth_fused_kernel<PP, RR, LL>;                                                              \
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize,    \
                            (int)lds);                                                                        \
    if (e != hipSuccess) return e;                                                                            \
    hipLaunchKernelGGL(k, dim3(blocks), dim3(LL * 16), lds, s, phi_in, phi_out, rhs, res, partial, n, ng, dx2,    \
                       oneoverdx2, zchunk, ntx, nty, (double *)nullptr, (double *)nullptr, (const double *)nullptr, ny, nz); \
  } while (0)
  if (restr) {
    // (the residual itself is not stored: the restricted right-hand side is all the coarse level needs of it)
    auto k = mg_smooth_fused_kernel<2, true, 32, true>;
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k, dim3(blocks), dim3(32 * 16), lds, s, phi_in, phi_out, rhs, res, partial, n, ng, dx2, oneoverdx2, zchunk, ntx, nty,
                       rhs_c, u1_c, (const double *)nullptr, ny, nz);
  } else if (prol) {
    auto k = mg_smooth_fused_kernel<2, false, 32, false, true>;
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k, dim3(blocks), dim3(32 * 16), lds, s, phi_in, phi_out, rhs, res, partial, n, ng, dx2, oneoverdx2, zchunk, ntx, nty,
                       (double *)nullptr, (double *)nullptr, corr_c, ny, nz);
  } else
  if (P == 4) { if (resid) SM_LAUNCH(4, true, 24); else SM_LAUNCH(4, false, 24); }
  else if (LY == 12) { if (resid) SM_LAUNCH(2, true, 12); else SM_LAUNCH(2, false, 12); }
  else if (LY == 16) { if (resid) SM_LAUNCH(2, true, 16); else SM_LAUNCH(2, false, 16); }
  else if (LY == 32) { if (resid) SM_LAUNCH(2, true, 32); else SM_LAUNCH(2, false, 32); }
  else { if (resid) SM_LAUNCH(2, true, 24); else SM_LAUNCH(2, false, 24); }
#undef SM_LAUNCH
  if (norm_out)
    hipLaunchKernelGGL(mg_sum_partials_kernel, dim3(1), dim3(256), 0, s, partial, blocks, dx * dx * dx, norm_out);
  return hipGetLastError();
}

}  // namespace ramses_amd

#include "warm.hpp"
RAMSES_AMD_TU_WARM(mg_kernels)
