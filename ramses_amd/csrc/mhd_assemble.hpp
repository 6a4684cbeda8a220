// mhd_assemble.hpp -- which neighbours feed which step of mag_unsplit (mhd/umuscl.f90:31-238), written once over an
// accessor so that the device kernels (csrc/mhd_sweep.hip: a periodic brick in HBM) and the host check of the header
// (tests/native/mhd_host_check.cpp: the reference's 6^3 stencil) walk the same code.
//
// An accessor A provides, for integer cell coordinates (i, j, k):
//   A.q(n, i, j, k)    primitive variable n = 0..7 of the cell (ctoprim's q)
//   A.bf(c, i, j, k)   face field c = 0,1,2 on the cell's LOW face of direction c (ctoprim's bf)
//   A.E(c, i, j, k)    edge field c = 0,1,2 (trace3d's Ex, Ey, Ez) on the cell's low edge of direction c
#pragma once
#include "mhd_core.hpp"

namespace ramses_amd {
namespace mhd {

// trace3d :811-838: the three edge-centred electric fields at the low x-, y-, z-edge of cell (i,j,k)
template <class A>
MHD_FN double efield(const A &a, int c, int i, int j, int k) {
  if (c == 0) {
    const double v = 0.25 * (a.q(2, i, j - 1, k - 1) + a.q(2, i, j - 1, k) + a.q(2, i, j, k - 1) + a.q(2, i, j, k));
    const double w = 0.25 * (a.q(3, i, j - 1, k - 1) + a.q(3, i, j - 1, k) + a.q(3, i, j, k - 1) + a.q(3, i, j, k));
    const double B = 0.5 * (a.bf(1, i, j, k - 1) + a.bf(1, i, j, k));
    const double C = 0.5 * (a.bf(2, i, j - 1, k) + a.bf(2, i, j, k));
    return v * C - w * B;
  }
  if (c == 1) {
    const double u = 0.25 * (a.q(1, i - 1, j, k - 1) + a.q(1, i - 1, j, k) + a.q(1, i, j, k - 1) + a.q(1, i, j, k));
    const double w = 0.25 * (a.q(3, i - 1, j, k - 1) + a.q(3, i - 1, j, k) + a.q(3, i, j, k - 1) + a.q(3, i, j, k));
    const double Af = 0.5 * (a.bf(0, i, j, k - 1) + a.bf(0, i, j, k));
    const double C = 0.5 * (a.bf(2, i - 1, j, k) + a.bf(2, i, j, k));
    return w * Af - u * C;
  }
  const double u = 0.25 * (a.q(1, i - 1, j - 1, k) + a.q(1, i - 1, j, k) + a.q(1, i, j - 1, k) + a.q(1, i, j, k));
  const double v = 0.25 * (a.q(2, i - 1, j - 1, k) + a.q(2, i - 1, j, k) + a.q(2, i, j - 1, k) + a.q(2, i, j, k));
  const double Af = 0.5 * (a.bf(0, i, j - 1, k) + a.bf(0, i, j, k));
  const double B = 0.5 * (a.bf(1, i - 1, j, k) + a.bf(1, i, j, k));
  return u * B - v * Af;
}

// uslope + the gathers of trace3d (:841-925) for cell (i,j,k)
// (S3: slope_type = 3 is compiled in -- its loop over the 27 neighbours indexes the slope arrays dynamically, which costs the
// device kernel its registers; the device instantiates the trace with and without it)
template <bool S3 = true, class A>
MHD_FN void trace_inputs(const A &a, int i, int j, int k, const MhdConst &P, TraceIn &I) {
  const int st = P.slope_type, sm = P.slope_mag_type;
  const double th = P.slope_theta;
  for (int n = 0; n < 8; n++) {
    const double q0 = a.q(n, i, j, k);
    I.q[n] = q0;
    if (S3 && st == 3) {
      // positivity-preserving 3-D unsplit slope (uslope :2420-2484): the central differences, scaled back so that no corner
      // value leaves the range of the 27 neighbours
      double vmin = 0.0, vmax = 0.0;      // (the centre's own difference is 0)
      for (int dk = -1; dk <= 1; dk++)
        for (int dj = -1; dj <= 1; dj++)
          for (int di = -1; di <= 1; di++) {
            const double df = a.q(n, i + di, j + dj, k + dk) - q0;
            vmin = fmin2(vmin, df);
            vmax = fmax2(vmax, df);
          }
      const double dfx = 0.5 * (a.q(n, i + 1, j, k) - a.q(n, i - 1, j, k));
      const double dfy = 0.5 * (a.q(n, i, j + 1, k) - a.q(n, i, j - 1, k));
      const double dfz = 0.5 * (a.q(n, i, j, k + 1) - a.q(n, i, j, k - 1));
      const double dff = 0.5 * (__builtin_fabs(dfx) + __builtin_fabs(dfy) + __builtin_fabs(dfz));
      const double slop = dff > 0.0 ? fmin2(1.0, fmin2(__builtin_fabs(vmin), __builtin_fabs(vmax)) / dff) : 1.0;
      I.dq[0][n] = slop * dfx;
      I.dq[1][n] = slop * dfy;
      I.dq[2][n] = slop * dfz;
      continue;
    }
    I.dq[0][n] = slope(st, th, a.q(n, i - 1, j, k), q0, a.q(n, i + 1, j, k));
    I.dq[1][n] = slope(st, th, a.q(n, i, j - 1, k), q0, a.q(n, i, j + 1, k));
    I.dq[2][n] = slope(st, th, a.q(n, i, j, k - 1), q0, a.q(n, i, j, k + 1));
  }
  I.AL = a.bf(0, i, j, k); I.AR = a.bf(0, i + 1, j, k);
  I.BL = a.bf(1, i, j, k); I.BR = a.bf(1, i, j + 1, k);
  I.CL = a.bf(2, i, j, k); I.CR = a.bf(2, i, j, k + 1);
  // dbf(:,:,:,1,1:2): Bx along y and z on the faces i and i+1 (uslope :2576-2603) ...
  I.dALy = slope(sm, th, a.bf(0, i, j - 1, k), I.AL, a.bf(0, i, j + 1, k));
  I.dALz = slope(sm, th, a.bf(0, i, j, k - 1), I.AL, a.bf(0, i, j, k + 1));
  I.dARy = slope(sm, th, a.bf(0, i + 1, j - 1, k), I.AR, a.bf(0, i + 1, j + 1, k));
  I.dARz = slope(sm, th, a.bf(0, i + 1, j, k - 1), I.AR, a.bf(0, i + 1, j, k + 1));
  // ... By along x and z on the faces j and j+1 (:2604-2632) ...
  I.dBLx = slope(sm, th, a.bf(1, i - 1, j, k), I.BL, a.bf(1, i + 1, j, k));
  I.dBLz = slope(sm, th, a.bf(1, i, j, k - 1), I.BL, a.bf(1, i, j, k + 1));
  I.dBRx = slope(sm, th, a.bf(1, i - 1, j + 1, k), I.BR, a.bf(1, i + 1, j + 1, k));
  I.dBRz = slope(sm, th, a.bf(1, i, j + 1, k - 1), I.BR, a.bf(1, i, j + 1, k + 1));
  // ... Bz along x and y on the faces k and k+1 (:2633-2662)
  I.dCLx = slope(sm, th, a.bf(2, i - 1, j, k), I.CL, a.bf(2, i + 1, j, k));
  I.dCLy = slope(sm, th, a.bf(2, i, j - 1, k), I.CL, a.bf(2, i, j + 1, k));
  I.dCRx = slope(sm, th, a.bf(2, i - 1, j, k + 1), I.CR, a.bf(2, i + 1, j, k + 1));
  I.dCRy = slope(sm, th, a.bf(2, i, j - 1, k + 1), I.CR, a.bf(2, i, j + 1, k + 1));
  // trace3d :927-940
  I.ELL = a.E(0, i, j, k); I.ELR = a.E(0, i, j, k + 1); I.ERL = a.E(0, i, j + 1, k); I.ERR = a.E(0, i, j + 1, k + 1);
  I.FLL = a.E(1, i, j, k); I.FLR = a.E(1, i, j, k + 1); I.FRL = a.E(1, i + 1, j, k); I.FRR = a.E(1, i + 1, j, k + 1);
  I.GLL = a.E(2, i, j, k); I.GLR = a.E(2, i, j + 1, k); I.GRL = a.E(2, i + 1, j, k); I.GRR = a.E(2, i + 1, j + 1, k);
}

// mag_unsplit's three cmpflxm calls (:95-157): the flux through the LOW face of direction d of cell (i,j,k) joins the
// +d state of the cell below (qm shifted by one) and the -d state of the cell itself.
// mag_unsplit's three cmp_mag_flx calls (:160-236): which cell's qRT / qRB / qLT / qLB are the routine's four states at the
// LOW edge of direction e of cell (i,j,k) -- offsets (di,dj,dk) of the cell relative to (i,j,k):
//   e = 2 (z):  RT_ = qRT(i-1,j-1,k)  RB_ = qRB(i-1,j,k)  LT_ = qLT(i,j-1,k)  LB_ = qLB(i,j,k)
//   e = 1 (y):  RT_ = qRT(i-1,j,k-1)  RB_ = qLT(i,j,k-1)  LT_ = qRB(i-1,j,k)  LB_ = qLB(i,j,k)     (the call swaps qLT and qRB)
//   e = 0 (x):  RT_ = qRT(i,j-1,k-1)  RB_ = qRB(i,j-1,k)  LT_ = qLT(i,j,k-1)  LB_ = qLB(i,j,k)
// kind: 0 qRT, 1 qRB, 2 qLT, 3 qLB
struct EdgeSource { int kind[4]; int off[4][3]; };
MHD_FN EdgeSource edge_sources(int e) {
  EdgeSource S;
  if (e == 2) {
    S = {{0, 1, 2, 3}, {{-1, -1, 0}, {-1, 0, 0}, {0, -1, 0}, {0, 0, 0}}};
  } else if (e == 1) {
    S = {{0, 2, 1, 3}, {{-1, 0, -1}, {0, 0, -1}, {-1, 0, 0}, {0, 0, 0}}};
  } else {
    S = {{0, 1, 2, 3}, {{0, -1, -1}, {0, -1, 0}, {0, 0, -1}, {0, 0, 0}}};
  }
  return S;
}

}  // namespace mhd
}  // namespace ramses_amd
