// hydro_core.hpp -- per-cell / per-interface device math of the Godunov sweep.
//
// MI355X-native formulation: everything here works on one cell (or one
// interface) held in registers; the sweep kernel (hydro_sweep.hip) owns the
// data movement (HBM -> registers, LDS plane exchange, z-marching).  There is
// no per-oct 6^3 patch and no nvector batch as in the reference.
//
// Each function names the reference routine whose arithmetic it reproduces.
// In the strict build (-ffp-contract=off) the operation order is the
// reference's, so results are bit-identical to the reference's x86-64 build.
#pragma once
#include <hip/hip_runtime.h>

namespace ramses_amd {

constexpr int RIEMANN_LLF = 0, RIEMANN_HLLC = 1, RIEMANN_HLL = 2,
              RIEMANN_ACOUSTIC = 3, RIEMANN_EXACT = 4;

// Constants derived once on the host from &HYDRO_PARAMS.
struct HydroConst {
  double gamma;
  double smallr;
  double smallc;
  double smallp;       // smallc**2/gamma
  double smalle;       // smallc**2/gamma/(gamma-1)
  double entho;        // 1/(gamma-1)
  double gm1;          // gamma-1
  double smallc2;      // smallc**2
  double gamma6;       // (gamma+1)/(2 gamma)
  double smallpp;      // smallr*smallp
  double oneovergamma; // 1/gamma
  double slope_theta;
  int niter_riemann;
};

#define RA_DEV __device__ __forceinline__

RA_DEV double dmaxd(double a, double b) { return __builtin_fmax(a, b); }
RA_DEV double dmind(double a, double b) { return __builtin_fmin(a, b); }
RA_DEV double fsignd(double a, double b) { return __builtin_copysign(a, b); }

// ---------------------------------------------------------------------------
// Arithmetic policy.  Strict build: IEEE division and square root (correctly
// rounded, bit-identical to the reference's x86-64 build).  Fast build
// (-DRAMSES_AMD_FAST): v_rcp_f64 / v_rsq_f64 seeds + Newton steps, ~1 ulp, no
// denormal/overflow rescaling (the operands here are densities, pressures and
// their ratios); held to <=1e-12 relative L-infinity of the strict result.
// ---------------------------------------------------------------------------
#ifdef RAMSES_AMD_FAST
#ifndef RAMSES_AMD_RCP_ONE_STEP
#define RAMSES_AMD_RCP_ONE_STEP 1   // one Newton step after v_rcp_f64 (~2^-50): fast vs strict stays at ~2e-15 over 24 Sedov steps
#endif
RA_DEV double rcp_fast(double x) {
  double r = __builtin_amdgcn_rcp(x);
  double e = __builtin_fma(-x, r, 1.0);
  r = __builtin_fma(r, e, r);
#if !RAMSES_AMD_RCP_ONE_STEP
  e = __builtin_fma(-x, r, 1.0);
  r = __builtin_fma(r, e, r);
#endif
  return r;
}
RA_DEV double ddiv(double a, double b) { return a * rcp_fast(b); }
RA_DEV double dsqrt(double x) {
  // Goldschmidt: g -> sqrt(x), h -> 1/(2 sqrt(x))
  double y = __builtin_amdgcn_rsq(x);
  double g = x * y;
  double h = 0.5 * y;
  double r = __builtin_fma(-h, g, 0.5);
  g = __builtin_fma(g, r, g);
  h = __builtin_fma(h, r, h);
  const double d = __builtin_fma(-g, g, x);
  g = __builtin_fma(d, h, g);
  return x == 0.0 ? 0.0 : g;
}
#else
RA_DEV double ddiv(double a, double b) { return a / b; }
RA_DEV double dsqrt(double x) { return __builtin_sqrt(x); }
#endif

// ---------------------------------------------------------------------------
// ctoprim (hydro/umuscl.f90:861-965) for one cell, 3-D, NENER=0.
// u = (rho, mx, my, mz, E [, scalars]); g = gravity (or 0); q = (rho,u,v,w,P[,s])
// The sound speed is not produced: only scheme='plmde' reads it.
// ---------------------------------------------------------------------------
template <int NV, bool GRAV>
RA_DEV void ctoprim_cell(const double (&u)[NV], const double (&g)[3],
                         double dtxhalf, const HydroConst &P, double (&q)[NV]) {
  const double rho = dmaxd(u[0], P.smallr);
  const double oneoverrho = ddiv(1.0, rho);
  const double vx = u[1] * oneoverrho;
  const double vy = u[2] * oneoverrho;
  const double vz = u[3] * oneoverrho;
#ifdef RAMSES_AMD_FAST
  // every fused multiply-add of the fast build is written out (the unit is compiled with
  // -ffp-contract=off): each cell gets the same arithmetic whichever row role or tile computes it
  double k2 = vx * vx;
  k2 = __builtin_fma(vy, vy, k2);
  k2 = __builtin_fma(vz, vz, k2);
  const double eint = dmaxd(__builtin_fma(u[4], oneoverrho, -0.5 * k2), P.smalle);
#else
  double eken = 0.5 * vx * vx;
  eken = eken + 0.5 * vy * vy;
  eken = eken + 0.5 * vz * vz;
  const double eint = dmaxd(u[4] * oneoverrho - eken, P.smalle);
#endif
  q[0] = rho;
  q[4] = P.gm1 * rho * eint;
  if (GRAV) {
#ifdef RAMSES_AMD_FAST
    q[1] = __builtin_fma(g[0], dtxhalf, vx);
    q[2] = __builtin_fma(g[1], dtxhalf, vy);
    q[3] = __builtin_fma(g[2], dtxhalf, vz);
#else
    q[1] = vx + g[0] * dtxhalf;
    q[2] = vy + g[1] * dtxhalf;
    q[3] = vz + g[2] * dtxhalf;
#endif
  } else {
    // gravin is identically zero when poisson=.false.; v + 0*dt == v
    q[1] = vx; q[2] = vy; q[3] = vz;
  }
#pragma unroll
  for (int n = 5; n < NV; n++) q[n] = u[n] * oneoverrho;
}

// sound speed of ctoprim (umuscl.f90:924-930), needed by PLMDE only
RA_DEV double ctoprim_sound(double rho, double p, const HydroConst &P) {
  const double oneoverrho = 1.0 / rho;
  return __builtin_sqrt(P.gamma * p * oneoverrho);
}

// ---------------------------------------------------------------------------
// uslope (hydro/umuscl.f90:970-1480), 3-D branches, one variable, one direction
// ---------------------------------------------------------------------------
template <int ST>
RA_DEV double slope1(double qm1, double q0, double qp1, const HydroConst &P) {
  if (ST == 0) return 0.0;
  if (ST == 1) {  // minmod, umuscl.f90:1246-1279
    const double dlft = q0 - qm1;
    const double drgt = qp1 - q0;
#ifdef RAMSES_AMD_FAST
    // branch-free form; differs from the reference only when dlft*drgt underflows
    return dmaxd(dmind(dlft, drgt), dmind(dmaxd(dlft, drgt), 0.0));
#endif
    const double s = dlft > 0 ? dmind(dlft, drgt) : dmaxd(dlft, drgt);
    return (dlft * drgt) <= 0.0 ? 0.0 : s;
  }
  if (ST == 2) {  // moncen, umuscl.f90:1292-1325
    const double dlft = 2.0 * (q0 - qm1);
    const double drgt = 2.0 * (qp1 - q0);
    const double dcen = 0.5 * (dlft + drgt) / 2.0;
    const double dsgn = fsignd(1.0, dcen);
    double dlim = dmind(__builtin_fabs(dlft), __builtin_fabs(drgt));
    if ((dlft * drgt) <= 0.0) dlim = 0.0;
    return dsgn * dmind(dlim, __builtin_fabs(dcen));
  }
  if (ST == 7) {  // van Leer, umuscl.f90:1387-1418
    const double dlft = q0 - qm1;
    const double drgt = qp1 - q0;
    return (dlft * drgt) <= 0.0 ? 0.0 : (2 * dlft * drgt / (dlft + drgt));
  }
  if (ST == 8) {  // generalised moncen/minmod, umuscl.f90:1423-1460
    const double dlft = q0 - qm1;
    const double drgt = qp1 - q0;
    const double dcen = 0.5 * (dlft + drgt);
    const double dsgn = fsignd(1.0, dcen);
    double dlim = dmind(P.slope_theta * __builtin_fabs(dlft), P.slope_theta * __builtin_fabs(drgt));
    if ((dlft * drgt) <= 0.0) dlim = 0.0;
    return dsgn * dmind(dlim, __builtin_fabs(dcen));
  }
  return 0.0;
}

// The slope types only an NDIM=1 build of the reference has (hydro/umuscl.f90:1030-1090: 4 superbee, 5 ultrabee,
// 6 "unstable" central difference).  dcen = q(u)*dt/dx of the cell along the direction ((u*dt)/dx, the reference's
// order), n = variable index: types 5 and 6 limit the density only, every other slope is zero.  Strict arithmetic
// (the reference's divisions) in both builds.
template <int ST>
RA_DEV double slope1_1d(double qm1, double q0, double qp1, double dcen, int n) {
  if (ST == 4) {
    const double dlft = 2.0 / (1.0 + dcen) * (q0 - qm1);
    const double drgt = 2.0 / (1.0 - dcen) * (qp1 - q0);
    const double dsgn = fsignd(1.0, dlft);
    double dlim = dmind(__builtin_fabs(dlft), __builtin_fabs(drgt));
    if ((dlft * drgt) <= 0.0) dlim = 0.0;
    return dsgn * dlim;
  }
  if (ST == 5) {
    if (n != 0) return 0.0;
    double dlft, drgt;
    if (dcen >= 0) {
      dlft = 2.0 / (0.0 + dcen + 1e-10) * (q0 - qm1);
      drgt = 2.0 / (1.0 - dcen) * (qp1 - q0);
    } else {
      dlft = 2.0 / (1.0 + dcen) * (q0 - qm1);
      drgt = 2.0 / (0.0 - dcen + 1e-10) * (qp1 - q0);
    }
    const double dsgn = fsignd(1.0, dlft);
    double dlim = dmind(__builtin_fabs(dlft), __builtin_fabs(drgt));
    if ((dlft * drgt) <= 0.0) dlim = 0.0;
    return dsgn * dlim;
  }
  if (ST == 6) {
    if (n != 0) return 0.0;
    const double dlft = q0 - qm1;
    const double drgt = qp1 - q0;
    return 0.5 * (dlft + drgt);
  }
  return 0.0;
}

// positivity-preserving unsplit slope (slope_type=3, umuscl.f90:1326-1386):
// nb[27] = the 3x3x3 neighbourhood of one variable, index (di+1)+3*(dj+1)+9*(dk+1)
RA_DEV void slope3_var(const double (&nb)[27], double (&d)[3]) {
  const double q0 = nb[13];
  double vmin = nb[0] - q0, vmax = vmin;
#pragma unroll
  for (int t = 1; t < 27; t++) {
    const double df = nb[t] - q0;
    vmin = dmind(vmin, df);
    vmax = dmaxd(vmax, df);
  }
  const double dfx = 0.5 * (nb[14] - nb[12]);
  const double dfy = 0.5 * (nb[16] - nb[10]);
  const double dfz = 0.5 * (nb[22] - nb[4]);
  const double dff = 0.5 * (__builtin_fabs(dfx) + __builtin_fabs(dfy) + __builtin_fabs(dfz));
  double slop = 1.0;
  if (dff > 0.0) slop = dmind(1.0, dmind(__builtin_fabs(vmin), __builtin_fabs(vmax)) / dff);
  d[0] = slop * dfx;
  d[1] = slop * dfy;
  d[2] = slop * dfz;
}

// ---------------------------------------------------------------------------
// trace3d (hydro/umuscl.f90:483-708) for one cell.
// dq[d][n]: slope of variable n along d.  Outputs qm[d][n] (state on the +d
// face of the cell) and qp[d][n] (state on the -d face).
// ---------------------------------------------------------------------------
template <int NV>
RA_DEV void trace3d_cell(const double (&q)[NV], const double (&dq)[3][NV],
                         double dtdx, double dtdy, double dtdz,
                         const HydroConst &P, double (&qm)[3][NV],
                         double (&qp)[3][NV]) {
  const double r = q[0], u = q[1], v = q[2], w = q[3], p = q[4];
  const double drx = dq[0][0], dux = dq[0][1], dvx = dq[0][2], dwx = dq[0][3], dpx = dq[0][4];
  const double dry = dq[1][0], duy = dq[1][1], dvy = dq[1][2], dwy = dq[1][3], dpy = dq[1][4];
  const double drz = dq[2][0], duz = dq[2][1], dvz = dq[2][2], dwz = dq[2][3], dpz = dq[2][4];
  const double div = dux + dvy + dwz;
#ifdef RAMSES_AMD_FAST
  // s = -(u dq/dx + v dq/dy + w dq/dz + source) as explicit FMA chains
  const double rinv = rcp_fast(r);
  auto adv = [&](double ax, double ay, double az, double c0, double c1) {
    double t = u * ax;
    t = __builtin_fma(v, ay, t);
    t = __builtin_fma(w, az, t);
    return -__builtin_fma(c0, c1, t);
  };
  const double sr0 = adv(drx, dry, drz, div, r);
  const double sp0 = adv(dpx, dpy, dpz, div * P.gamma, p);
  const double su0 = adv(dux, duy, duz, dpx, rinv);
  const double sv0 = adv(dvx, dvy, dvz, dpy, rinv);
  const double sw0 = adv(dwx, dwy, dwz, dpz, rinv);
#else
  const double sr0 = -u * drx - v * dry - w * drz - (div)*r;
  const double sp0 = -u * dpx - v * dpy - w * dpz - (div)*P.gamma * p;
  const double su0 = -u * dux - v * duy - w * duz - (dpx) / r;
  const double sv0 = -u * dvx - v * dvy - w * dvz - (dpy) / r;
  const double sw0 = -u * dwx - v * dwy - w * dwz - (dpz) / r;
#endif
  const double s0[5] = {sr0, su0, sv0, sw0, sp0};
  const double dtd[3] = {dtdx, dtdy, dtdz};
#pragma unroll
  for (int d = 0; d < 3; d++) {
#ifdef RAMSES_AMD_FAST
    // q + s0*dt/(2dx) once, then one FMA per face state (the strict build keeps (q -+ dq/2) + s0*dt/(2dx))
    const double hdt = 0.5 * dtd[d];
#pragma unroll
    for (int n = 0; n < 5; n++) {
      const double base = __builtin_fma(s0[n], hdt, q[n]);
      qp[d][n] = __builtin_fma(-0.5, dq[d][n], base);
      qm[d][n] = __builtin_fma(0.5, dq[d][n], base);
    }
#else
#pragma unroll
    for (int n = 0; n < 5; n++) {
      const double hd = 0.5 * dq[d][n];
      const double st = s0[n] * dtd[d] * 0.5;
      qp[d][n] = q[n] - hd + st;
      qm[d][n] = q[n] + hd + st;
    }
#endif
    if (qp[d][0] < P.smallr) qp[d][0] = r;
    if (qm[d][0] < P.smallr) qm[d][0] = r;
  }
  // passive scalars, umuscl.f90:681-706
#pragma unroll
  for (int n = 5; n < NV; n++) {
    const double a = q[n];
    const double sa0 = -u * dq[0][n] - v * dq[1][n] - w * dq[2][n];
#pragma unroll
    for (int d = 0; d < 3; d++) {
      qp[d][n] = a - 0.5 * dq[d][n] + sa0 * dtd[d] * 0.5;
      qm[d][n] = a + 0.5 * dq[d][n] + sa0 * dtd[d] * 0.5;
    }
  }
}

// ---------------------------------------------------------------------------
// tracexyz (hydro/uplmde.f90:375-696): PLMDE characteristic tracing for one
// cell, 3-D.  cc = sound speed of ctoprim.  The reference's quirk is kept: all
// transverse terms are scaled with half*dtdx (uplmde.f90:453-469).
// ---------------------------------------------------------------------------
template <int NV>
RA_DEV void tracexyz_cell(const double (&q)[NV], const double (&dq)[3][NV], double cc,
                          double dtdx, double dtdy, double dtdz, const HydroConst &P,
                          double (&qm)[3][NV], double (&qp)[3][NV]) {
  const double r = q[0], p = q[4];
  const double vel[3] = {q[1], q[2], q[3]};
  const double csq = P.gamma * p / r;
  const double dtd[3] = {dtdx, dtdy, dtdz};
  const double fac = 0.5 * dtdx;
  // transverse direction pairs in the reference's order of subtraction
  constexpr int T[3][2] = {{1, 2}, {0, 2}, {1, 0}};
#pragma unroll
  for (int d = 0; d < 3; d++) {
    const int t0 = T[d][0], t1 = T[d][1];
    // transverse derivative terms
    const double ar = -vel[t0] * dq[t0][0] - vel[t1] * dq[t1][0];
    const double ap = -vel[t0] * dq[t0][4] - vel[t1] * dq[t1][4];
    const double divt = dq[t0][1 + t0] + dq[t1][1 + t1];
    const double sr = fac * (ar - (divt)*r);
    const double sp = fac * (ap - (divt)*P.gamma * p);
    double sv[3];
#pragma unroll
    for (int c = 0; c < 3; c++) {
      const double av = -vel[t0] * dq[t0][1 + c] - vel[t1] * dq[t1][1 + c];
      sv[c] = (c == d) ? fac * (av) : fac * (av - (dq[c][4]) / r);
    }
    // characteristic analysis along d
    const double vn = vel[d];
    const double dvn = dq[d][1 + d];
    const double alpham = 0.5 * (dq[d][4] / csq - dvn * r / cc);
    const double alphap = 0.5 * (dq[d][4] / csq + dvn * r / cc);
    const double alpha0r = dq[d][0] - dq[d][4] / csq;
    double ccc = cc;
    if (__builtin_fabs(dvn) > 3.0 * cc) ccc = 0.0;
#pragma unroll
    for (int side = 0; side < 2; side++) {
      double spminus = (vn - ccc) * dtd[d];
      double spplus = (vn + ccc) * dtd[d];
      double spzero = (vn)*dtd[d];
      double sg;
      if (side == 0) {  // right state at the left interface (qp)
        if ((vn + ccc) > 0.0) spplus = -1.0;
        if ((vn - ccc) > 0.0) spminus = -1.0;
        if (vn > 0.0) spzero = -1.0;
        sg = -1.0;
      } else {          // left state at the right interface (qm)
        if ((vn + ccc) <= 0.0) spplus = 1.0;
        if ((vn - ccc) <= 0.0) spminus = 1.0;
        if (vn <= 0.0) spzero = 1.0;
        sg = 1.0;
      }
      const double ap_ = 0.5 * (sg - spplus) * alphap;
      const double am_ = 0.5 * (sg - spminus) * alpham;
      const double azr = 0.5 * (sg - spzero) * alpha0r;
      double(&out)[3][NV] = side == 0 ? qp : qm;
      out[d][0] = dmaxd(P.smallr, r + (ap_ + am_ + azr) + sr);
      out[d][1 + d] = vn + (ap_ - am_) * cc / r + sv[d];
      out[d][4] = p + (ap_ + am_) * csq + sp;
#pragma unroll
      for (int c = 0; c < 3; c++) {
        if (c == d) continue;
        const double azt = 0.5 * (sg - spzero) * dq[d][1 + c];
        out[d][1 + c] = vel[c] + (azt) + sv[c];
      }
    }
  }
}

// ---------------------------------------------------------------------------
// Riemann solvers (hydro/godunov_utils.f90).  States are in cmpflxm's
// permuted order (rho, u_normal, P, u_t1, u_t2, scalars...); the flux comes
// back in the same order: (rho, mom_n, E, mom_t1, mom_t2, scalars...).
// f[NV] = internal-energy flux (only used with pressure_fix).
// ---------------------------------------------------------------------------
template <int NV>
RA_DEV void riemann_llf(const double (&ql)[NV], const double (&qr)[NV],
                        const HydroConst &P, double (&f)[NV + 1]) {
  // godunov_utils.f90:660-820
  const double rl = dmaxd(ql[0], P.smallr), ul = ql[1];
  const double pl = dmaxd(ql[2], rl * P.smallp);
  const double cl = dsqrt(ddiv(P.gamma * pl, rl));
  const double rr = dmaxd(qr[0], P.smallr), ur = qr[1];
  const double pr = dmaxd(qr[2], rr * P.smallp);
  const double cr = dsqrt(ddiv(P.gamma * pr, rr));
  const double cmax = dmaxd(__builtin_fabs(ul) + cl, __builtin_fabs(ur) + cr);
  double uL[NV + 1], uR[NV + 1], fL[NV + 1], fR[NV + 1];
  uL[0] = ql[0]; uR[0] = qr[0];
  uL[1] = ql[0] * ql[1]; uR[1] = qr[0] * qr[1];
  uL[2] = ql[2] * P.entho + 0.5 * ql[0] * (ql[1] * ql[1]);
  uR[2] = qr[2] * P.entho + 0.5 * qr[0] * (qr[1] * qr[1]);
  uL[2] = uL[2] + 0.5 * ql[0] * (ql[3] * ql[3]);
  uR[2] = uR[2] + 0.5 * qr[0] * (qr[3] * qr[3]);
  uL[2] = uL[2] + 0.5 * ql[0] * (ql[4] * ql[4]);
  uR[2] = uR[2] + 0.5 * qr[0] * (qr[4] * qr[4]);
#pragma unroll
  for (int n = 3; n < NV; n++) { uL[n] = ql[0] * ql[n]; uR[n] = qr[0] * qr[n]; }
  uL[NV] = ql[2] * P.entho; uR[NV] = qr[2] * P.entho;
  fL[0] = ql[1] * uL[0]; fR[0] = qr[1] * uR[0];
  fL[1] = ql[1] * uL[1] + ql[2]; fR[1] = qr[1] * uR[1] + qr[2];
  fL[2] = ql[1] * (uL[2] + ql[2]); fR[2] = qr[1] * (uR[2] + qr[2]);
#pragma unroll
  for (int n = 3; n <= NV; n++) { fL[n] = ql[1] * uL[n]; fR[n] = qr[1] * uR[n]; }
#pragma unroll
  for (int n = 0; n <= NV; n++) f[n] = 0.5 * (fL[n] + fR[n] - cmax * (uR[n] - uL[n]));
}

template <int NV>
RA_DEV void riemann_hll(const double (&ql)[NV], const double (&qr)[NV],
                        const HydroConst &P, double (&f)[NV + 1]) {
  // godunov_utils.f90:825-983
  const double rl = dmaxd(ql[0], P.smallr), ul = ql[1];
  const double pl = dmaxd(ql[2], rl * P.smallp);
  const double cl = dsqrt(ddiv(P.gamma * pl, rl));
  const double rr = dmaxd(qr[0], P.smallr), ur = qr[1];
  const double pr = dmaxd(qr[2], rr * P.smallp);
  const double cr = dsqrt(ddiv(P.gamma * pr, rr));
  const double SL = dmind(dmind(ul, ur) - dmaxd(cl, cr), 0.0);
  const double SR = dmaxd(dmaxd(ul, ur) + dmaxd(cl, cr), 0.0);
  double uL[NV + 1], uR[NV + 1], fL[NV + 1], fR[NV + 1];
  uL[0] = ql[0]; uR[0] = qr[0];
  uL[1] = ql[0] * ql[1]; uR[1] = qr[0] * qr[1];
  uL[2] = ql[2] * P.entho + 0.5 * ql[0] * (ql[1] * ql[1]);
  uR[2] = qr[2] * P.entho + 0.5 * qr[0] * (qr[1] * qr[1]);
  uL[2] = uL[2] + 0.5 * ql[0] * (ql[3] * ql[3]);
  uR[2] = uR[2] + 0.5 * qr[0] * (qr[3] * qr[3]);
  uL[2] = uL[2] + 0.5 * ql[0] * (ql[4] * ql[4]);
  uR[2] = uR[2] + 0.5 * qr[0] * (qr[4] * qr[4]);
#pragma unroll
  for (int n = 3; n < NV; n++) { uL[n] = ql[0] * ql[n]; uR[n] = qr[0] * qr[n]; }
  uL[NV] = ql[2] * P.entho; uR[NV] = qr[2] * P.entho;
  fL[0] = uL[1]; fR[0] = uR[1];
  fL[1] = ql[2] + uL[1] * ql[1]; fR[1] = qr[2] + uR[1] * qr[1];
  fL[2] = ql[1] * (uL[2] + ql[2]); fR[2] = qr[1] * (uR[2] + qr[2]);
#pragma unroll
  for (int n = 3; n <= NV; n++) { fL[n] = ql[1] * uL[n]; fR[n] = qr[1] * uR[n]; }
#pragma unroll
  for (int n = 0; n <= NV; n++)
    f[n] = ddiv(SR * fL[n] - SL * fR[n] + SR * SL * (uR[n] - uL[n]), SR - SL);
}

template <int NV>
RA_DEV void riemann_hllc(const double (&ql)[NV], const double (&qr)[NV],
                         const HydroConst &P, double (&f)[NV + 1]) {
  // godunov_utils.f90:988-1209
  const double rl = dmaxd(ql[0], P.smallr);
  const double Pl = dmaxd(ql[2], rl * P.smallp);
  const double ul = ql[1];
  const double el = Pl * P.entho;
  double ecinl = 0.5 * rl * ul * ul;
  ecinl = ecinl + 0.5 * rl * (ql[3] * ql[3]);
  ecinl = ecinl + 0.5 * rl * (ql[4] * ql[4]);
  const double etotl = el + ecinl;
  const double rr = dmaxd(qr[0], P.smallr);
  const double Pr = dmaxd(qr[2], rr * P.smallp);
  const double ur = qr[1];
  const double er = Pr * P.entho;
  double ecinr = 0.5 * rr * ur * ur;
  ecinr = ecinr + 0.5 * rr * (qr[3] * qr[3]);
  ecinr = ecinr + 0.5 * rr * (qr[4] * qr[4]);
  const double etotr = er + ecinr;
  const double cfastl = dsqrt(dmaxd(ddiv(P.gamma * Pl, rl), P.smallc2));
  const double cfastr = dsqrt(dmaxd(ddiv(P.gamma * Pr, rr), P.smallc2));
  const double SL = dmind(ul, ur) - dmaxd(cfastl, cfastr);
  const double SR = dmaxd(ul, ur) + dmaxd(cfastl, cfastr);
  const double rcl = rl * (ul - SL);
  const double rcr = rr * (SR - ur);
  // (ddiv: the IEEE division in the strict build, v_rcp_f64 + a Newton step in the fast one -- the passive-scalar
  //  instantiations of the fast build come through here; NV = 5 takes hllc_flux_fast below)
  const double ustar = ddiv(rcr * ur + rcl * ul + (Pl - Pr), rcr + rcl);
  const double Ptotstar = ddiv(rcr * Pl + rcl * Pr + rcl * rcr * (ul - ur), rcr + rcl);
  const double rstarl = ddiv(rl * (SL - ul), SL - ustar);
  const double etotstarl = ddiv((SL - ul) * etotl - Pl * ul + Ptotstar * ustar, SL - ustar);
  const double estarl = ddiv(el * (SL - ul), SL - ustar);
  const double rstarr = ddiv(rr * (SR - ur), SR - ustar);
  const double etotstarr = ddiv((SR - ur) * etotr - Pr * ur + Ptotstar * ustar, SR - ustar);
  const double estarr = ddiv(er * (SR - ur), SR - ustar);
  double ro, uo, Ptoto, etoto, eo;
  if (SL > 0.0) { ro = rl; uo = ul; Ptoto = Pl; etoto = etotl; eo = el; }
  else if (ustar > 0.0) { ro = rstarl; uo = ustar; Ptoto = Ptotstar; etoto = etotstarl; eo = estarl; }
  else if (SR > 0.0) { ro = rstarr; uo = ustar; Ptoto = Ptotstar; etoto = etotstarr; eo = estarr; }
  else { ro = rr; uo = ur; Ptoto = Pr; etoto = etotr; eo = er; }
  f[0] = ro * uo;
  f[1] = ro * uo * uo + Ptoto;
  f[2] = (etoto + Ptoto) * uo;
#pragma unroll
  for (int n = 3; n < NV; n++) f[n] = ustar > 0 ? ro * uo * ql[n] : ro * uo * qr[n];
  f[NV] = uo * eo;
}

// shared tail of riemann_approx/acoustic: godunov_utils.f90:465-493, 627-653
template <int NV>
RA_DEV void gdnv_to_flux(const double (&qg)[NV + 1], const HydroConst &P,
                         double (&f)[NV + 1]) {
  f[0] = qg[0] * qg[1];
  f[1] = qg[2] + qg[0] * (qg[1] * qg[1]);
  double etot = qg[2] * P.entho + 0.5 * qg[0] * (qg[1] * qg[1]);
  etot = etot + 0.5 * qg[0] * (qg[3] * qg[3]);
  etot = etot + 0.5 * qg[0] * (qg[4] * qg[4]);
  f[2] = qg[1] * (etot + qg[2]);
#pragma unroll
  for (int n = 3; n <= NV; n++) f[n] = f[0] * qg[n];
}

template <int NV>
RA_DEV void riemann_acoustic(const double (&ql)[NV], const double (&qr)[NV],
                             const HydroConst &P, double (&f)[NV + 1]) {
  // godunov_utils.f90:500-655
  const double rl = dmaxd(ql[0], P.smallr), ul = ql[1], pl = dmaxd(ql[2], rl * P.smallp);
  const double rr = dmaxd(qr[0], P.smallr), ur = qr[1], pr = dmaxd(qr[2], rr * P.smallp);
  const double cl = dsqrt(ddiv(P.gamma * pl, rl));
  const double cr = dsqrt(ddiv(P.gamma * pr, rr));
  const double wl = cl * rl, wr = cr * rr;
  const double pstar = ddiv((wr * pl + wl * pr) + wl * wr * (ul - ur), wl + wr);   // (ddiv / dsqrt: IEEE in the strict build)
  const double ustar = ddiv((wr * ur + wl * ul) + (pl - pr), wl + wr);
  const double sgnm = fsignd(1.0, ustar);
  const bool left = sgnm == 1.0;
  const double ro = left ? rl : rr, uo = left ? ul : ur, po = left ? pl : pr, co = left ? cl : cr;
  double rstar = ro + ddiv(pstar - po, co * co);
  rstar = dmaxd(rstar, P.smallr);
  double cstar = dsqrt(__builtin_fabs(ddiv(P.gamma * pstar, rstar)));
  cstar = dmaxd(cstar, P.smallc);
  double spout = co - sgnm * uo;
  double spin = cstar - sgnm * ustar;
  double ushock = 0.5 * (spin + spout);
  ushock = dmaxd(ushock, -sgnm * ustar);
  if (pstar >= po) { spout = ushock; spin = spout; }
  double qg[NV + 1];
  if (spout < 0.0) { qg[0] = ro; qg[1] = uo; qg[2] = po; }
  else if (spin >= 0.0) { qg[0] = rstar; qg[1] = ustar; qg[2] = pstar; }
  else {
    const double frac = ddiv(spout, spout - spin);
    qg[0] = frac * rstar + (1.0 - frac) * ro;
    qg[1] = frac * ustar + (1.0 - frac) * uo;
    qg[2] = frac * pstar + (1.0 - frac) * po;
  }
#pragma unroll
  for (int n = 3; n < NV; n++) qg[n] = left ? ql[n] : qr[n];
  qg[NV] = ddiv(po, ro) * P.entho;
  gdnv_to_flux<NV>(qg, P, f);
}

template <int NV>
RA_DEV void riemann_exact(const double (&ql)[NV], const double (&qr)[NV],
                          const HydroConst &P, double (&f)[NV + 1]) {
  // riemann_approx, godunov_utils.f90:268-495.  The reference compacts the
  // not-yet-converged lanes of its nvector batch; per interface that is:
  // Newton steps until |delp/(p+smallpp)| <= 1e-6, at most niter_riemann.
  const double gamma = P.gamma;
  const double rl = dmaxd(ql[0], P.smallr), ul = ql[1], pl = dmaxd(ql[2], rl * P.smallp);
  const double rr = dmaxd(qr[0], P.smallr), ur = qr[1], pr = dmaxd(qr[2], rr * P.smallp);
  const double cl = gamma * pl * rl, cr = gamma * pr * rr;
  double wl = __builtin_sqrt(cl), wr = __builtin_sqrt(cr);
  double pstar = ((wr * pl + wl * pr) + wl * wr * (ul - ur)) / (wl + wr);
  pstar = dmaxd(pstar, 0.0);
  double pold = pstar;
  bool live = true;
  for (int iter = 0; iter < P.niter_riemann; iter++) {
    if (live) {
      const double wwl = __builtin_sqrt(cl * (1.0 + P.gamma6 * (pold - pl) / pl));
      const double wwr = __builtin_sqrt(cr * (1.0 + P.gamma6 * (pold - pr) / pr));
      const double qql = 2.0 * (wwl * wwl * wwl) / (wwl * wwl + cl);
      const double qqr = 2.0 * (wwr * wwr * wwr) / (wwr * wwr + cr);
      const double usl = ul - (pold - pl) / wwl;
      const double usr = ur + (pold - pr) / wwr;
      const double delp = dmaxd(qqr * qql / (qqr + qql) * (usl - usr), -pold);
      pold = pold + delp;
      const double conv = __builtin_fabs(delp / (pold + P.smallpp));
      if (!(conv > 1e-06)) live = false;
    }
  }
  pstar = pold;
  wl = __builtin_sqrt(cl * (1.0 + P.gamma6 * (pstar - pl) / pl));
  wr = __builtin_sqrt(cr * (1.0 + P.gamma6 * (pstar - pr) / pr));
  const double ustar = 0.5 * (ul + (pl - pstar) / wl + ur - (pr - pstar) / wr);
  const double sgnm = fsignd(1.0, ustar);
  const bool left = sgnm == 1.0;
  const double ro = left ? rl : rr, uo = left ? ul : ur, po = left ? pl : pr, wo = left ? wl : wr;
  const double co = dmaxd(P.smallc, __builtin_sqrt(__builtin_fabs(gamma * po / ro)));
  double rstar;
  if (pstar >= po) rstar = ro / (1.0 + ro * (po - pstar) / (wo * wo));
  else rstar = ro * pow(pstar / po, P.oneovergamma);
  rstar = dmaxd(rstar, P.smallr);
  double cstar = __builtin_sqrt(__builtin_fabs(gamma * pstar / rstar));
  cstar = dmaxd(cstar, P.smallc);
  double spout = co - sgnm * uo;
  double spin = cstar - sgnm * ustar;
  const double ushock = wo / ro - sgnm * uo;
  if (pstar >= po) { spout = ushock; spin = spout; }
  double qg[NV + 1];
  if (spout <= 0.0) { qg[0] = ro; qg[1] = uo; qg[2] = po; }
  else if (spin >= 0.0) { qg[0] = rstar; qg[1] = ustar; qg[2] = pstar; }
  else {
    const double frac = spout / (spout - spin);
    qg[1] = frac * ustar + (1.0 - frac) * uo;
    qg[2] = frac * pstar + (1.0 - frac) * po;
    qg[0] = ro * pow(qg[2] / po, P.oneovergamma);
  }
#pragma unroll
  for (int n = 3; n < NV; n++) qg[n] = left ? ql[n] : qr[n];
  qg[NV] = po / ro * P.entho;
  gdnv_to_flux<NV>(qg, P, f);
}

template <int RS, int NV>
RA_DEV void riemann_solve(const double (&ql)[NV], const double (&qr)[NV],
                          const HydroConst &P, double (&f)[NV + 1]) {
  if (RS == RIEMANN_LLF) riemann_llf<NV>(ql, qr, P, f);
  else if (RS == RIEMANN_HLLC) riemann_hllc<NV>(ql, qr, P, f);
  else if (RS == RIEMANN_HLL) riemann_hll<NV>(ql, qr, P, f);
  else if (RS == RIEMANN_ACOUSTIC) riemann_acoustic<NV>(ql, qr, P, f);
  else riemann_exact<NV>(ql, qr, P, f);
}

// ---------------------------------------------------------------------------
// cmpflxm (hydro/umuscl.f90:714-856) for one interface normal to DIR.
// qL = qm of the cell on the low side, qR = qp of the cell on the high side,
// both in natural order (rho,u,v,w,P,...).  flux comes back in natural order
// (rho, mx, my, mz, E, ...), unscaled; eflux = internal energy flux,
// unorm = half*(uL_n+uR_n)  (the reference's tmp(:,1:2)).
// ---------------------------------------------------------------------------
template <int RS, int NV, int DIR>
RA_DEV void interface_flux(const double (&qL)[NV], const double (&qR)[NV],
                           const HydroConst &P, double (&flux)[NV],
                           double &unorm, double &eflux) {
  constexpr int ln = DIR == 0 ? 1 : (DIR == 1 ? 2 : 3);
  constexpr int lt1 = DIR == 0 ? 2 : 1;
  constexpr int lt2 = DIR == 2 ? 2 : 3;
  double a[NV], b[NV], f[NV + 1];
  a[0] = qL[0]; b[0] = qR[0];
  a[1] = qL[ln]; b[1] = qR[ln];
  a[2] = qL[4]; b[2] = qR[4];
  a[3] = qL[lt1]; b[3] = qR[lt1];
  a[4] = qL[lt2]; b[4] = qR[lt2];
#pragma unroll
  for (int n = 5; n < NV; n++) { a[n] = qL[n]; b[n] = qR[n]; }
  riemann_solve<RS, NV>(a, b, P, f);
  flux[0] = f[0];
  flux[ln] = f[1];
  flux[lt1] = f[3];
  flux[lt2] = f[4];
  flux[4] = f[2];
#pragma unroll
  for (int n = 5; n < NV; n++) flux[n] = f[n];
  unorm = 0.5 * (a[1] + b[1]);
  eflux = f[NV];
}

// ---------------------------------------------------------------------------
// Scaled interface flux  flux*dt/dx  (hydro/umuscl.f90:101-163).  Strict build:
// the reference's two operations ((f*dt)/dx; an exact multiply by 1/dx when dx
// is a power of two).  Fast build + LLF: one fused routine (FMA forms, rsq-based
// sound speed, the dt/dx factor folded into the Lax-Friedrichs average).
// ---------------------------------------------------------------------------
#ifdef RAMSES_AMD_FAST
RA_DEV double rsqrt_fast(double x) {
  double y = __builtin_amdgcn_rsq(x);
  const double e = __builtin_fma(-(x * y), y, 1.0);
  return __builtin_fma(0.5 * y, e, y);
}
template <int DIR>
RA_DEV void llf_flux_fast(const double (&qL)[5], const double (&qR)[5],
                          const HydroConst &P, double dtdx, double (&flux)[5]) {
  constexpr int ln = DIR == 0 ? 1 : (DIR == 1 ? 2 : 3);
  constexpr int lt1 = DIR == 0 ? 2 : 1;
  constexpr int lt2 = DIR == 2 ? 2 : 3;
  const double rl = dmaxd(qL[0], P.smallr), pl = dmaxd(qL[4], rl * P.smallp);
  const double rr = dmaxd(qR[0], P.smallr), pr = dmaxd(qR[4], rr * P.smallp);
  const double gl = P.gamma * pl, gr = P.gamma * pr;
  const double cl = gl * rsqrt_fast(gl * rl);   // sqrt(gamma p / rho)
  const double cr = gr * rsqrt_fast(gr * rr);
  const double ul = qL[ln], ur = qR[ln];
  const double cmax = dmaxd(__builtin_fabs(ul) + cl, __builtin_fabs(ur) + cr);
  const double hs = 0.5 * dtdx;           // 0.5*dt/dx
  const double hc = cmax * hs;            // 0.5*cmax*dt/dx
  // conserved states
  const double mL = qL[0] * ul, mR = qR[0] * ur;
  double kL = ul * ul; kL = __builtin_fma(qL[lt1], qL[lt1], kL); kL = __builtin_fma(qL[lt2], qL[lt2], kL);
  double kR = ur * ur; kR = __builtin_fma(qR[lt1], qR[lt1], kR); kR = __builtin_fma(qR[lt2], qR[lt2], kR);
  const double eL = __builtin_fma(0.5 * qL[0], kL, qL[4] * P.entho);
  const double eR = __builtin_fma(0.5 * qR[0], kR, qR[4] * P.entho);
  const double t1L = qL[0] * qL[lt1], t1R = qR[0] * qR[lt1];
  const double t2L = qL[0] * qL[lt2], t2R = qR[0] * qR[lt2];
  // physical fluxes
  const double fnL = __builtin_fma(ul, mL, qL[4]), fnR = __builtin_fma(ur, mR, qR[4]);
  const double feR = ur * (eR + qR[4]);
  const double fesum = __builtin_fma(ul, eL + qL[4], feR);
  flux[0] = __builtin_fma(hc, qL[0] - qR[0], hs * (mL + mR));
  flux[ln] = __builtin_fma(hc, mL - mR, hs * (fnL + fnR));
  flux[lt1] = __builtin_fma(hc, t1L - t1R, hs * __builtin_fma(ul, t1L, ur * t1R));
  flux[lt2] = __builtin_fma(hc, t2L - t2R, hs * __builtin_fma(ul, t2L, ur * t2R));
  flux[4] = __builtin_fma(hc, eL - eR, hs * fesum);
}
// HLLC the same way (round 6; RAMSES_AMD_HLLC_FUSED=0 gives the generic routine back).  The fast build ran HLLC -- the solver most
// production namelists choose -- through riemann_hllc with eight IEEE divisions: 5.24 ms per 512^3 sweep against LLF's 2.97.
// Step 1, the divisions through ddiv: 4.21 ms.  Step 2, this fused form: one reciprocal of the wave-speed sum, ONE star state
// -- the side the contact speed points to -- with one reciprocal, FMA forms, rsq-based sound speeds, dt/dx folded in: 11 % fewer
// instructions, <= 5e-15 of the strict build over 40 steps.  It keeps both sides' values alive up to the choice, and the 12-row
// kernels sit at their 168-register limit: with the plane also held in registers (hydro_sweep.hip KEEP) they spilled 44 - 100 B
// per lane and the sweep took 7.3 ms; the HLLC kernels therefore re-read the plane from L2 (SWEEP_KEEP_NOT_HLLC), spill the
// generic routine's 8 B, and the sweep takes 3.51 ms (minmod) / 4.59 ms (moncen)  (profiles/r06_hllc_fast.txt).
template <int DIR>
RA_DEV void hllc_flux_fast(const double (&qL)[5], const double (&qR)[5],
                           const HydroConst &P, double dtdx, double (&flux)[5]) {
  constexpr int ln = DIR == 0 ? 1 : (DIR == 1 ? 2 : 3);
  constexpr int lt1 = DIR == 0 ? 2 : 1;
  constexpr int lt2 = DIR == 2 ? 2 : 3;
  const double rl = dmaxd(qL[0], P.smallr), Pl = dmaxd(qL[4], rl * P.smallp);
  const double rr = dmaxd(qR[0], P.smallr), Pr = dmaxd(qR[4], rr * P.smallp);
  const double ul = qL[ln], ur = qR[ln];
  const double gl = P.gamma * Pl, gr = P.gamma * Pr;
  const double cl = dmaxd(gl * rsqrt_fast(gl * rl), P.smallc);   // sqrt(max(gamma P / rho, smallc^2))
  const double cr = dmaxd(gr * rsqrt_fast(gr * rr), P.smallc);
  const double cm = dmaxd(cl, cr);
  const double SL = dmind(ul, ur) - cm, SR = dmaxd(ul, ur) + cm;
  const double rcl = rl * (ul - SL), rcr = rr * (SR - ur);
  const double rsum = rcp_fast(rcr + rcl);
  const double ustar = __builtin_fma(rcr, ur, __builtin_fma(rcl, ul, Pl - Pr)) * rsum;
  const double Pstar = __builtin_fma(rcl * rcr, ul - ur, __builtin_fma(rcr, Pl, rcl * Pr)) * rsum;
  // (SL > 0 implies u* > 0 with a margin of c (1 - 1/gamma): the side is the sign of u* in every branch of the chain)
  const bool left = ustar > 0.0;
  const bool star = left ? !(SL > 0.0) : (SR > 0.0);
  const double rk = left ? rl : rr, uk = left ? ul : ur, Pk = left ? Pl : Pr, S = left ? SL : SR;
  // (values first: a conditional between two array ELEMENTS is a conditional between two addresses, which parks the arrays in scratch)
  const double vl1 = qL[lt1], vr1 = qR[lt1], vl2 = qL[lt2], vr2 = qR[lt2];
  const double w1 = left ? vl1 : vr1, w2 = left ? vl2 : vr2;
  double k2 = uk * uk; k2 = __builtin_fma(w1, w1, k2); k2 = __builtin_fma(w2, w2, k2);
  const double etk = __builtin_fma(0.5 * rk, k2, Pk * P.entho);
  const double inv = rcp_fast(S - ustar);
  const double dS = S - uk;
  const double rst = rk * dS * inv;
  const double ets = __builtin_fma(Pstar, ustar, __builtin_fma(dS, etk, -(Pk * uk))) * inv;
  const double ro = star ? rst : rk, uo = star ? ustar : uk, Po = star ? Pstar : Pk, eto = star ? ets : etk;
  const double m = ro * uo * dtdx;
  flux[0] = m;
  flux[ln] = __builtin_fma(m, uo, Po * dtdx);
  flux[lt1] = m * w1;
  flux[lt2] = m * w2;
  flux[4] = (eto + Po) * (uo * dtdx);
#if RAMSES_AMD_HLLC_FUSED == 2
  __builtin_amdgcn_sched_barrier(0);     // (the three interface fluxes of a cell one after the other, not interleaved)
#endif
}
#endif

// FUSE_HLLC = false: the sweep of a level in tiles and its surface pass (which must agree with each other flux by flux) take
// the generic HLLC routine -- their kernels carry the tile bookkeeping on top and the fused flux's live values spill there:
// 256^3 level in tiles 1.005 ms fused, 0.830 ms generic; shell level 2.40 / 1.98 ms (profiles/r06_hllc_fast.txt)
template <int RS, int NV, int DIR, bool FUSE_HLLC = true>
RA_DEV void scaled_interface_flux(const double (&qL)[NV], const double (&qR)[NV],
                                  const HydroConst &P, double dt, double dx, double rdx,
                                  double dtdx, bool DXPOW2, double (&flux)[NV]) {
#ifdef RAMSES_AMD_FAST
  if constexpr (RS == RIEMANN_LLF && NV == 5) {
    llf_flux_fast<DIR>(qL, qR, P, dtdx, flux);
    return;
  }
#ifndef RAMSES_AMD_HLLC_FUSED
#define RAMSES_AMD_HLLC_FUSED 1     // (2: + a scheduling barrier after each interface flux -- measured, no better)
#endif
  if constexpr (RAMSES_AMD_HLLC_FUSED != 0 && FUSE_HLLC && RS == RIEMANN_HLLC && NV == 5) {
    hllc_flux_fast<DIR>(qL, qR, P, dtdx, flux);
    return;
  }
#endif
  double un_, ef_;
  interface_flux<RS, NV, DIR>(qL, qR, P, flux, un_, ef_);
#pragma unroll
  for (int n = 0; n < NV; n++) {
#ifdef RAMSES_AMD_FAST
    flux[n] = flux[n] * dtdx;
#else
    flux[n] = DXPOW2 ? flux[n] * dt * rdx : flux[n] * dt / dx;
#endif
  }
}

// The same plus cmpflxm's two extra face quantities (hydro/umuscl.f90:843-850, scaled like
// the fluxes in unsplit :136-163): tmp[0] = half*(uL+uR) normal velocity, tmp[1] = internal
// energy flux -- the divu/enew updates of pressure_fix.  Strict arithmetic only.
template <int RS, int NV, int DIR>
RA_DEV void scaled_interface_flux_tmp(const double (&qL)[NV], const double (&qR)[NV],
                                      const HydroConst &P, double dt, double dx, double rdx,
                                      bool DXPOW2, double (&flux)[NV], double (&tmp)[2]) {
  double un_, ef_;
  interface_flux<RS, NV, DIR>(qL, qR, P, flux, un_, ef_);
#pragma unroll
  for (int n = 0; n < NV; n++) flux[n] = DXPOW2 ? flux[n] * dt * rdx : flux[n] * dt / dx;
  tmp[0] = DXPOW2 ? un_ * dt * rdx : un_ * dt / dx;
  tmp[1] = DXPOW2 ? ef_ * dt * rdx : ef_ * dt / dx;
}

// ---------------------------------------------------------------------------
// cmpdt (hydro/godunov_utils.f90:5-120) for one cell, 3-D.
// ---------------------------------------------------------------------------
template <int NV, bool GRAV>
RA_DEV double cmpdt_cell(const double (&u)[NV], const double (&g)[3], double dx,
                         double courant_factor, const HydroConst &P, double ndimf = 3.0) {
  const double rho = dmaxd(u[0], P.smallr);
  const double vx = u[1] / rho, vy = u[2] / rho, vz = u[3] / rho;
  double e = u[4];
  e = e - 0.5 * rho * (vx * vx);
  e = e - 0.5 * rho * (vy * vy);
  e = e - 0.5 * rho * (vz * vz);
  double pc = dmaxd(P.gm1 * e, rho * P.smallp);
  pc = P.gamma * pc;
  pc = __builtin_sqrt(pc / rho);
  pc = ndimf * pc;   // dble(ndim)*c: 1-D/2-D problems embedded in a brick keep their own NDIM
  pc = pc + __builtin_fabs(vx);
  pc = pc + __builtin_fabs(vy);
  pc = pc + __builtin_fabs(vz);
  double gs = 0.0;
  if (GRAV) {
    gs = gs + __builtin_fabs(g[0]);
    gs = gs + __builtin_fabs(g[1]);
    gs = gs + __builtin_fabs(g[2]);
  }
  gs = gs * dx / (pc * pc);
  gs = dmaxd(gs, 0.0001);
  return dx / pc * (__builtin_sqrt(1.0 + 2.0 * courant_factor * gs) - 1.0) / gs;
}

}  // namespace ramses_amd
