// amr_ops.hip -- coarse<->fine hydro operators on level bricks:
//   interpol_hydro (+ limiters)   hydro/interpol_hydro.f90:268-444, 449-637
//   upl / upload_fine             hydro/interpol_hydro.f90:5-68, 73-263
// A coarse brick (n^3, periodic) and its fully refined child brick ((2n)^3).
// One thread per coarse cell: 7-point coarse stencil in, 8 children out
// (prolongation), or 8 children in, 1 parent out (restriction).  Streaming,
// HBM-bound; the reference's operation order is kept (bit parity).
#include <hip/hip_runtime.h>

#include "amr_args.hpp"

namespace ramses_amd {

__device__ __forceinline__ int wrapc(int i, int n) { return i < 0 ? i + n : (i >= n ? i - n : i); }
__device__ __forceinline__ double dmx(double a, double b) { return __builtin_fmax(a, b); }
__device__ __forceinline__ double dmn(double a, double b) { return __builtin_fmin(a, b); }

__device__ __forceinline__ void lim_central_raw(const double (&a)[7], double (&w)[3]) {
#pragma unroll
  for (int d = 0; d < 3; d++) w[d] = 0.25 * (a[2 * d + 2] - a[2 * d + 1]);
}
__device__ __forceinline__ void lim_minmod(const double (&a)[7], double (&w)[3]) {
#pragma unroll
  for (int d = 0; d < 3; d++) {
    const double dl = 0.5 * (a[2 * d + 2] - a[0]);
    const double dr = 0.5 * (a[0] - a[2 * d + 1]);
    double mm = 0.0;
    if (!(dl * dr <= 0.0)) mm = dmn(__builtin_fabs(dl), __builtin_fabs(dr)) * dl / __builtin_fabs(dl);
    w[d] = mm;
  }
}
__device__ __forceinline__ void lim_central(const double (&a)[7], double (&w)[3]) {
  lim_central_raw(a, w);
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
  for (int j = 1; j < 8; j++) corner = dmx(corner, ac[j]);
#pragma unroll
  for (int j = 2; j <= 6; j++) kernel = dmx(kernel, a[j]);
  double dk = a[0] - kernel, dc = a[0] - corner;
  double max_lim = 0.0;
  if (dk * dc > 0.0) max_lim = dmn(1.0, dk / dc);
  corner = ac[0]; kernel = a[1];
#pragma unroll
  for (int j = 1; j < 8; j++) corner = dmn(corner, ac[j]);
#pragma unroll
  for (int j = 2; j <= 6; j++) kernel = dmn(kernel, a[j]);
  dk = a[0] - kernel; dc = a[0] - corner;
  double min_lim = 0.0;
  if (dk * dc > 0.0) min_lim = dmn(1.0, dk / dc);
  const double lim = dmn(min_lim, max_lim);
#pragma unroll
  for (int d = 0; d < 3; d++) w[d] = w[d] * lim;
}

template <int NV>
__global__ __launch_bounds__(256) void interpol_hydro_kernel(AmrOpArgs A) {
  const int n = A.nc;
  const long N = (long)n * n * n;
  const int nf = 2 * n;
  const long Nf = (long)nf * nf * nf;
  for (long c = (long)blockIdx.x * blockDim.x + threadIdx.x; c < N; c += (long)gridDim.x * blockDim.x) {
    const int I = (int)(c % n), J = (int)((c / n) % n), K = (int)(c / ((long)n * n));
    // stencil order of getnborfather: 0 = cell, then -x,+x,-y,+y,-z,+z
    long nb[7];
    nb[0] = c;
    nb[1] = (long)wrapc(I - 1, n) + (long)n * (J + (long)n * K);
    nb[2] = (long)wrapc(I + 1, n) + (long)n * (J + (long)n * K);
    nb[3] = (long)I + (long)n * (wrapc(J - 1, n) + (long)n * K);
    nb[4] = (long)I + (long)n * (wrapc(J + 1, n) + (long)n * K);
    nb[5] = (long)I + (long)n * (J + (long)n * wrapc(K - 1, n));
    nb[6] = (long)I + (long)n * (J + (long)n * wrapc(K + 1, n));
    double u1[7][NV];
#pragma unroll
    for (int j = 0; j < 7; j++)
#pragma unroll
      for (int v = 0; v < NV; v++) u1[j][v] = A.coarse[nb[j] + (long)v * N];
    if (A.interpol_var == 1 || A.interpol_var == 2) {
#pragma unroll
      for (int j = 0; j < 7; j++) {
        double ekin = 0.0;
#pragma unroll
        for (int d = 0; d < 3; d++) ekin = ekin + 0.5 * (u1[j][d + 1] * u1[j][d + 1]) / dmx(u1[j][0], A.smallr);
        u1[j][4] = u1[j][4] - ekin - 0.0;
        if (A.interpol_var == 2) {
#pragma unroll
          for (int d = 0; d < 3; d++) u1[j][d + 1] = u1[j][d + 1] / dmx(u1[j][0], A.smallr);
        }
      }
    }
    double u2[8][NV];
#pragma unroll
    for (int v = 0; v < NV; v++) {
      double a[7], w[3] = {0.0, 0.0, 0.0};
#pragma unroll
      for (int j = 0; j < 7; j++) a[j] = u1[j][v];
      if (A.interpol_type == 1) lim_minmod(a, w);
      else if (A.interpol_type == 2) lim_central(a, w);
      else if (A.interpol_type == 3) lim_central_raw(a, w);
      else if (A.interpol_type == 4) {
        if (v >= 1 && v <= 3) lim_central_raw(a, w);
        else lim_central(a, w);
      }
#pragma unroll
      for (int ind = 0; ind < 8; ind++) {
        double val = a[0];
#pragma unroll
        for (int d = 0; d < 3; d++) val = val + w[d] * ((double)((ind >> d) & 1) - 0.5);
        u2[ind][v] = val;
      }
    }
    if (A.interpol_var == 1 || A.interpol_var == 2) {
      if (A.interpol_var == 2) {
#pragma unroll
        for (int ind = 0; ind < 8; ind++)
#pragma unroll
          for (int d = 0; d < 3; d++) u2[ind][d + 1] = u2[ind][d + 1] * u2[ind][0];
#pragma unroll
        for (int d = 0; d < 3; d++) {
          double mom = 0.0;
#pragma unroll
          for (int ind = 0; ind < 8; ind++) mom = mom + u2[ind][d + 1] * 0.125;
          mom = mom - u1[0][d + 1] * u1[0][0];
#pragma unroll
          for (int ind = 0; ind < 8; ind++) u2[ind][d + 1] = u2[ind][d + 1] - mom;
        }
      }
#pragma unroll
      for (int ind = 0; ind < 8; ind++) {
        double ekin = 0.0;
#pragma unroll
        for (int d = 0; d < 3; d++) ekin = ekin + 0.5 * (u2[ind][d + 1] * u2[ind][d + 1]) / dmx(u2[ind][0], A.smallr);
        u2[ind][4] = u2[ind][4] + ekin + 0.0;
      }
    }
#pragma unroll
    for (int ind = 0; ind < 8; ind++) {
      const long f = (long)(2 * I + (ind & 1)) + (long)nf * ((2 * J + ((ind >> 1) & 1)) + (long)nf * (2 * K + ((ind >> 2) & 1)));
#pragma unroll
      for (int v = 0; v < NV; v++) A.fine[f + (long)v * Nf] = u2[ind][v];
    }
  }
}

template <int NV>
__global__ __launch_bounds__(256) void upload_fine_kernel(AmrOpArgs A) {
  const int n = A.nc;
  const long N = (long)n * n * n;
  const int nf = 2 * n;
  const long Nf = (long)nf * nf * nf;
  for (long c = (long)blockIdx.x * blockDim.x + threadIdx.x; c < N; c += (long)gridDim.x * blockDim.x) {
    const int I = (int)(c % n), J = (int)((c / n) % n), K = (int)(c / ((long)n * n));
    double ch[8][NV];
#pragma unroll
    for (int ind = 0; ind < 8; ind++) {
      const long f = (long)(2 * I + (ind & 1)) + (long)nf * ((2 * J + ((ind >> 1) & 1)) + (long)nf * (2 * K + ((ind >> 2) & 1)));
#pragma unroll
      for (int v = 0; v < NV; v++) ch[ind][v] = A.fine[f + (long)v * Nf];
    }
    double pa[NV];
    double getx = 0.0;
#pragma unroll
    for (int ind = 0; ind < 8; ind++) getx = getx + dmx(ch[ind][0], A.smallr);
    pa[0] = getx / 8.0;
#pragma unroll
    for (int v = 1; v < NV; v++) {
      getx = 0.0;
#pragma unroll
      for (int ind = 0; ind < 8; ind++) getx = getx + ch[ind][v];
      pa[v] = getx / 8.0;
    }
    if (A.interpol_var == 1 || A.interpol_var == 2) {
      getx = 0.0;
#pragma unroll
      for (int ind = 0; ind < 8; ind++) {
        double ekin = 0.0;
#pragma unroll
        for (int d = 0; d < 3; d++) ekin = ekin + 0.5 * (ch[ind][1 + d] * ch[ind][1 + d]) / dmx(ch[ind][0], A.smallr);
        getx = getx + ch[ind][4] - ekin - 0.0;
      }
      double ekin = 0.0;
#pragma unroll
      for (int d = 0; d < 3; d++) ekin = ekin + 0.5 * (pa[1 + d] * pa[1 + d]) / dmx(pa[0], A.smallr);
      pa[4] = getx / 8.0 + ekin + 0.0;
    }
#pragma unroll
    for (int v = 0; v < NV; v++) A.coarse[c + (long)v * N] = pa[v];
  }
}

hipError_t launch_amr_op(const AmrOpArgs &A, bool prolong, hipStream_t s) {
  const long N = (long)A.nc * A.nc * A.nc;
  long grid = (N + 255) / 256;
  if (grid > 8192) grid = 8192;
  if (A.nvar != 5) return hipErrorInvalidValue;
  if (prolong) hipLaunchKernelGGL(interpol_hydro_kernel<5>, dim3((int)grid), dim3(256), 0, s, A);
  else hipLaunchKernelGGL(upload_fine_kernel<5>, dim3((int)grid), dim3(256), 0, s, A);
  return hipGetLastError();
}

}  // namespace ramses_amd
