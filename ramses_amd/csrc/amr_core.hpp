// amr_core.hpp -- cell-level coarse->fine interpolation shared by the brick
// operators (amr_ops.hip) and the AMR sweep (amr_sweep.hip):
//   interpol_hydro (+ limiters)   hydro/interpol_hydro.f90:268-444, 449-637
// u1[0] = the coarse cell, u1[1..6] = its -x,+x,-y,+y,-z,+z neighbours
// (getnborfather order); u2[ind] = the 8 children, ind-1 = ix+2*iy+4*iz.
// The reference's operation order is kept (bit parity, -ffp-contract=off).
#pragma once
#include <hip/hip_runtime.h>

namespace ramses_amd {

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
__device__ __forceinline__ void interpol_hydro_cell(double (&u1)[7][NV], double (&u2)[8][NV], int interpol_var,
                                                    int interpol_type, double smallr) {
  struct { int interpol_var, interpol_type; double smallr; } A = {interpol_var, interpol_type, smallr};
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
}

// upl (hydro/interpol_hydro.f90:73-263): a split cell = mean of its 8 children ch[ind][v], ind-1 = ix+2*iy+4*iz;
// density floored before averaging; with interpol_var 1|2 the internal energy is averaged instead of the total one.
template <int NV>
__device__ __forceinline__ void upl_cell(const double (&ch)[8][NV], int interpol_var, double smallr, double (&pa)[NV]) {
  double getx = 0.0;
#pragma unroll
  for (int ind = 0; ind < 8; ind++) getx = getx + dmx(ch[ind][0], smallr);
  pa[0] = getx / 8.0;
#pragma unroll
  for (int v = 1; v < NV; v++) {
    getx = 0.0;
#pragma unroll
    for (int ind = 0; ind < 8; ind++) getx = getx + ch[ind][v];
    pa[v] = getx / 8.0;
  }
  if (interpol_var == 1 || interpol_var == 2) {
    getx = 0.0;
#pragma unroll
    for (int ind = 0; ind < 8; ind++) {
      double ekin = 0.0;
#pragma unroll
      for (int d = 0; d < 3; d++) ekin = ekin + 0.5 * (ch[ind][1 + d] * ch[ind][1 + d]) / dmx(ch[ind][0], smallr);
      getx = getx + ch[ind][4] - ekin - 0.0;
    }
    double ekin = 0.0;
#pragma unroll
    for (int d = 0; d < 3; d++) ekin = ekin + 0.5 * (pa[1 + d] * pa[1 + d]) / dmx(pa[0], smallr);
    pa[4] = getx / 8.0 + ekin + 0.0;
  }
}

}  // namespace ramses_amd
