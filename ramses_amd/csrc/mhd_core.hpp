// mhd_core.hpp -- per-cell / per-face / per-edge arithmetic of the MHD Godunov sweep (SOLVER=mhd of the reference:
// mhd/umuscl.f90 mag_unsplit :31-238 = ctoprim :2029-2186, uslope :2187-2844, trace3d :750-1307, cmpflxm :1308-1448,
// cmp_mag_flx :1453-2028; Riemann solvers mhd/godunov_utils.f90: lax_friedrich :352-386, hll :391-421, hlld :426-699,
// find_mhd_flux :704-782, find_speed_info :787-818, find_speed_fast :823-852), NDIM = 3, NVAR = 8, NENER = 0.
//
// Every function restates the reference's operations in the reference's ORDER (IEEE double, no contraction: the units
// that include this header are compiled with -ffp-contract=off), so that the constrained-transport sweep built from
// them (csrc/mhd_sweep.hip) returns the reference's bits.  The header also compiles for the host (plain C++): the CPU
// test tests/test_mhd_core_host.py runs these very functions over the reference's 6^3 stencils and compares them with
// the compiled reference's mag_unsplit (oracle/_ref/libref_kernels3d_mhd.so) without a GPU.
//
// Variable order (the reference's): q[0] = rho, q[1..3] = u, v, w, q[4] = P, q[5..7] = A, B, C (cell-centred field).
// 1-D Riemann states (cmpflxm's qleft / qright): [0] rho, [1] P, [2] v_n, [3] B_n, [4] v_t1, [5] B_t1, [6] v_t2, [7] B_t2.
#pragma once
#include <cmath>

#if defined(__HIPCC__)
#define MHD_FN __host__ __device__ __forceinline__
#else
#define MHD_FN inline
#endif

namespace ramses_amd {
namespace mhd {

// riemann (iriemann, hydro/read_hydro_params.f90:184-199) and riemann2d (iriemann2d, :205-220) codes of the reference
enum { RIEMANN_LLF = 0, RIEMANN_ROE = 1, RIEMANN_HLL = 2, RIEMANN_HLLD = 3, RIEMANN_UPWIND = 4, RIEMANN_HYDRO = 5 };
enum { RIEMANN2D_LLF = 0, RIEMANN2D_ROE = 1, RIEMANN2D_UPWIND = 2, RIEMANN2D_HLL = 3, RIEMANN2D_HLLA = 4, RIEMANN2D_HLLD = 5 };

struct MhdConst {
  double gamma, smallr, smallc, slope_theta;
  int slope_type, slope_mag_type, riemann, riemann2d;
};

MHD_FN double fmax2(double a, double b) { return __builtin_fmax(a, b); }
MHD_FN double fmin2(double a, double b) { return __builtin_fmin(a, b); }
MHD_FN double fmax4(double a, double b, double c, double d) { return fmax2(fmax2(fmax2(a, b), c), d); }
MHD_FN double fmin4(double a, double b, double c, double d) { return fmin2(fmin2(fmin2(a, b), c), d); }

// ---- ctoprim (umuscl.f90:2029-2186), one cell: u[0..4] = rho, rho u, rho v, rho w, E; bl / br = the fields on the
// cell's left and right faces (uin(6:8), uin(nvar+1:nvar+3)); g = gravin or null -------------------------------------
MHD_FN void ctoprim_cell(const double (&u)[5], const double (&bl)[3], const double (&br)[3], const double *g, double dt,
                         const MhdConst &P, double (&q)[8]) {
  const double smallp = P.smallr * (P.smallc * P.smallc) / P.gamma;
  q[0] = fmax2(u[0], P.smallr);
  q[1] = u[1] / q[0];
  q[2] = u[2] / q[0];
  q[3] = u[3] / q[0];
  q[5] = (bl[0] + br[0]) * 0.5;
  q[6] = (bl[1] + br[1]) * 0.5;
  q[7] = (bl[2] + br[2]) * 0.5;
  const double eken = 0.5 * (q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  const double emag = 0.5 * (q[5] * q[5] + q[6] * q[6] + q[7] * q[7]);
  const double erad = 0.0;
  const double etot = u[4] - emag - erad;
  const double eint = etot / q[0] - eken;
  q[4] = fmax2((P.gamma - 1.0) * q[0] * eint, smallp);
  if (g) {
    q[1] = q[1] + g[0] * dt * 0.5;
    q[2] = q[2] + g[1] * dt * 0.5;
    q[3] = q[3] + g[2] * dt * 0.5;
  }
}

// ---- cmpdt (mhd/godunov_utils.f90:5-115, ischeme = 0, no gravity), one cell: u[0..10] = rho, rho u, rho v, rho w, E, the
// fields on the left faces, the fields on the right faces; returns dtcell (the caller starts from courant_factor*dx/smallc
// and takes the minimum) ----------------------------------------------------------------------------------------------
MHD_FN double cmpdt_cell(const double (&u)[11], double dx, double courant_factor, const MhdConst &P) {
  const double smallp = P.smallr * (P.smallc * P.smallc) / P.gamma;
  const double rho = fmax2(u[0], P.smallr);
  const double v[3] = {u[1] / rho, u[2] / rho, u[3] / rho};
  double B2 = 0.0, e = u[4];
  for (int d = 0; d < 3; d++) {
    const double Bc = 0.5 * (u[5 + d] + u[8 + d]);
    B2 = B2 + Bc * Bc;
    e = e - 0.5 * rho * (v[d] * v[d]) - 0.5 * (Bc * Bc);
  }
  const double p = fmax2((P.gamma - 1.0) * e, smallp);
  const double a2 = P.gamma * p / rho;
  double ctot = 0.0;
  for (int d = 0; d < 3; d++) {
    const double cc = 0.5 * (B2 / rho + a2);
    const double BN = 0.5 * (u[5 + d] + u[8 + d]);
    const double cf = __builtin_sqrt(cc + __builtin_sqrt(cc * cc - a2 * (BN * BN) / rho));
    ctot = ctot + __builtin_fabs(v[d]) + cf;
  }
  // gravity strength ratio: no gravity, rho = max(0 * dx / ctot**2, 1e-4)
  double g = 0.0;
  g = g * dx / (ctot * ctot);
  g = fmax2(g, 0.0001);
  return dx / ctot * (__builtin_sqrt(1.0 + 2.0 * courant_factor * g) - 1.0) / g;
}

// ---- one TVD slope (uslope :2375-2571 for the cell-centred variables, :2572-2842 for the face-centred fields: the same
// limiter, chosen by slope_type / slope_mag_type) ---------------------------------------------------------------------
MHD_FN bool slope_mag_type_supported(int st) { return st == 0 || st == 1 || st == 2 || st == 7 || st == 8; }
// (slope_type = 3, the positivity-preserving unsplit slope :2420-2484, needs the 27 neighbours: mhd_assemble.hpp trace_inputs;
//  uslope has no such branch for the face fields, so it goes with an explicit slope_mag_type)
MHD_FN bool slope_type_supported(int st) { return slope_mag_type_supported(st) || st == 3; }
MHD_FN double slope(int st, double theta, double qm1, double q0, double qp1) {
  if (st == 1 || st == 2) {
    const double s = (double)st;
    const double dlft = s * (q0 - qm1);
    const double drgt = s * (qp1 - q0);
    const double dcen = 0.5 * (dlft + drgt) / s;
    const double dsgn = __builtin_copysign(1.0, dcen);
    const double slop = fmin2(__builtin_fabs(dlft), __builtin_fabs(drgt));
    double dlim = slop;
    if ((dlft * drgt) <= 0.0) dlim = 0.0;
    return dsgn * fmin2(dlim, __builtin_fabs(dcen));
  }
  if (st == 7) {
    const double dlft = q0 - qm1;
    const double drgt = qp1 - q0;
    if ((dlft * drgt) <= 0.0) return 0.0;
    return 2.0 * dlft * drgt / (dlft + drgt);
  }
  if (st == 8) {
    const double dlft = q0 - qm1;
    const double drgt = qp1 - q0;
    const double dcen = 0.5 * (dlft + drgt);
    const double dsgn = __builtin_copysign(1.0, dcen);
    const double slop = fmin2(theta * __builtin_fabs(dlft), theta * __builtin_fabs(drgt));
    double dlim = slop;
    if ((dlft * drgt) <= 0.0) dlim = 0.0;
    return dsgn * fmin2(dlim, __builtin_fabs(dcen));
  }
  return 0.0;
}

// ---- the edge-centred electric fields of trace3d (:816-838); v4 / w4 ... = the four cells around the edge in the
// reference's order of addition ---------------------------------------------------------------------------------------
MHD_FN double avg4(double a, double b, double c, double d) { return 0.25 * (a + b + c + d); }

// ---- trace3d (:841-1276), one cell ----------------------------------------------------------------------------------
// In : q (cell), face fields AL..CR, dq[d][n] = RAW slopes (uslope's dq), the twelve raw face slopes, the twelve edge
//      fields E/F/G (Ex, Ey, Ez at the cell's four x-, y-, z-edges: LL, LR, RL, RR as in the reference).
// Out: qm[d][n], qp[d][n] (face states), qRT/qRB/qLT/qLB[e][n] (edge states; e = 0,1,2 for the x-, y-, z-edges).
struct TraceIn {
  double q[8];
  double AL, AR, BL, BR, CL, CR;
  double dq[3][8];
  double dALy, dARy, dALz, dARz, dBLx, dBRx, dBLz, dBRz, dCLx, dCRx, dCLy, dCRy;   // raw (not yet halved)
  double ELL, ELR, ERL, ERR, FLL, FLR, FRL, FRR, GLL, GLR, GRL, GRR;
};
// a sink receives the eighteen states one by one: put(kind, d, s) with kind 0 qm, 1 qp, 2 qRT, 3 qRB, 4 qLT, 5 qLB
enum { T_QM = 0, T_QP = 1, T_QRT = 2, T_QRB = 3, T_QLT = 4, T_QLB = 5 };
struct TraceOut {
  double st[6][3][8];
  MHD_FN void put(int kind, int d, const double (&s)[8]) { for (int n = 0; n < 8; n++) st[kind][d][n] = s[n]; }
  MHD_FN const double (&get(int kind, int d) const)[8] { return st[kind][d]; }
};

// trace3d in two steps, so that the sweep can keep 47 numbers per cell instead of the 144 of its eighteen traced states:
//   trace_predict  the predicted cell-centred state c[0..7], the predicted face fields f[0..5] = AL, AR, BL, BR, CL, CR, the
//                  21 half slopes of the cell-centred variables (x: r u v w p B C, y: r u v w p A C, z: r u v w p A B) and the
//                  12 half slopes of the face fields (dALy dARy dALz dARz dBLx dBRx dBLz dBRz dCLx dCRx dCLy dCRy)   (:841-983)
//   trace_state<KIND, D>  one traced state from those numbers, floors included (:985-1276); a source S offers c(n), f(n), h(n)
//                  (compile-time indices: a device source loads from HBM exactly what the state needs)
constexpr int NPRED = 8 + 6 + 21 + 12;
struct TracePred {
  double v[NPRED];
  MHD_FN double c(int n) const { return v[n]; }
  MHD_FN double f(int n) const { return v[8 + n]; }
  MHD_FN double h(int n) const { return v[14 + n]; }
};

MHD_FN void trace_predict(const TraceIn &I, double dtdx, double dtdy, double dtdz, const MhdConst &P, TracePred &O) {
  const double gamma = P.gamma;
  const double half = 0.5;
  double r = I.q[0], u = I.q[1], v = I.q[2], w = I.q[3], p = I.q[4], A = I.q[5], B = I.q[6], C = I.q[7];
  double AL = I.AL, AR = I.AR, BL = I.BL, BR = I.BR, CL = I.CL, CR = I.CR;
  const double drx = half * I.dq[0][0], dux = half * I.dq[0][1], dvx = half * I.dq[0][2], dwx = half * I.dq[0][3],
               dpx = half * I.dq[0][4], dBx = half * I.dq[0][6], dCx = half * I.dq[0][7];
  const double dry = half * I.dq[1][0], duy = half * I.dq[1][1], dvy = half * I.dq[1][2], dwy = half * I.dq[1][3],
               dpy = half * I.dq[1][4], dAy = half * I.dq[1][5], dCy = half * I.dq[1][7];
  const double drz = half * I.dq[2][0], duz = half * I.dq[2][1], dvz = half * I.dq[2][2], dwz = half * I.dq[2][3],
               dpz = half * I.dq[2][4], dAz = half * I.dq[2][5], dBz = half * I.dq[2][6];
  const double ELL = I.ELL, ELR = I.ELR, ERL = I.ERL, ERR = I.ERR;
  const double FLL = I.FLL, FLR = I.FLR, FRL = I.FRL, FRR = I.FRR;
  const double GLL = I.GLL, GLR = I.GLR, GRL = I.GRL, GRR = I.GRR;

  // face-centred predicted states
  const double sAL0 = +(GLR - GLL) * dtdy * half - (FLR - FLL) * dtdz * half;
  const double sAR0 = +(GRR - GRL) * dtdy * half - (FRR - FRL) * dtdz * half;
  const double sBL0 = -(GRL - GLL) * dtdx * half + (ELR - ELL) * dtdz * half;
  const double sBR0 = -(GRR - GLR) * dtdx * half + (ERR - ERL) * dtdz * half;
  const double sCL0 = +(FRL - FLL) * dtdx * half - (ERL - ELL) * dtdy * half;
  const double sCR0 = +(FRR - FLR) * dtdx * half - (ERR - ELR) * dtdy * half;
  AL = AL + sAL0; AR = AR + sAR0;
  BL = BL + sBL0; BR = BR + sBR0;
  CL = CL + sCL0; CR = CR + sCR0;

  // source terms (including transverse derivatives)
  const double sr0 = (-u * drx - dux * r) * dtdx + (-v * dry - dvy * r) * dtdy + (-w * drz - dwz * r) * dtdz;
  const double su0 = (-u * dux - (dpx + B * dBx + C * dCx) / r) * dtdx + (-v * duy + B * dAy / r) * dtdy + (-w * duz + C * dAz / r) * dtdz;
  const double sv0 = (-u * dvx + A * dBx / r) * dtdx + (-v * dvy - (dpy + A * dAy + C * dCy) / r) * dtdy + (-w * dvz + C * dBz / r) * dtdz;
  const double sw0 = (-u * dwx + A * dCx / r) * dtdx + (-v * dwy + B * dCy / r) * dtdy + (-w * dwz - (dpz + A * dAz + B * dBz) / r) * dtdz;
  const double sp0 = (-u * dpx - dux * gamma * p) * dtdx + (-v * dpy - dvy * gamma * p) * dtdy + (-w * dpz - dwz * gamma * p) * dtdz;

  // cell-centred predicted states
  r = r + sr0; u = u + su0; v = v + sv0; w = w + sw0; p = p + sp0;
  A = 0.5 * (AL + AR); B = 0.5 * (BL + BR); C = 0.5 * (CL + CR);
  double *o = O.v;
  o[0] = r; o[1] = u; o[2] = v; o[3] = w; o[4] = p; o[5] = A; o[6] = B; o[7] = C;
  o[8] = AL; o[9] = AR; o[10] = BL; o[11] = BR; o[12] = CL; o[13] = CR;
  o[14] = drx; o[15] = dux; o[16] = dvx; o[17] = dwx; o[18] = dpx; o[19] = dBx; o[20] = dCx;
  o[21] = dry; o[22] = duy; o[23] = dvy; o[24] = dwy; o[25] = dpy; o[26] = dAy; o[27] = dCy;
  o[28] = drz; o[29] = duz; o[30] = dvz; o[31] = dwz; o[32] = dpz; o[33] = dAz; o[34] = dBz;
  o[35] = half * I.dALy; o[36] = half * I.dARy; o[37] = half * I.dALz; o[38] = half * I.dARz;
  o[39] = half * I.dBLx; o[40] = half * I.dBRx; o[41] = half * I.dBLz; o[42] = half * I.dBRz;
  o[43] = half * I.dCLx; o[44] = half * I.dCRx; o[45] = half * I.dCLy; o[46] = half * I.dCRy;
}

template <int KIND, int D, class Src>
MHD_FN void trace_state(const Src &S, const MhdConst &P, double (&s)[8]) {
  const double smallr = P.smallr;
  const double smallp = smallr * (P.smallc * P.smallc) / P.gamma;
  if constexpr (KIND == 1 && D == 0) { s[0] = S.c(0) - S.h(0); s[1] = S.c(1) - S.h(1); s[2] = S.c(2) - S.h(2); s[3] = S.c(3) - S.h(3); s[4] = S.c(4) - S.h(4); s[5] = S.f(0); s[6] = S.c(6) - S.h(5); s[7] = S.c(7) - S.h(6); }
  if constexpr (KIND == 0 && D == 0) { s[0] = S.c(0) + S.h(0); s[1] = S.c(1) + S.h(1); s[2] = S.c(2) + S.h(2); s[3] = S.c(3) + S.h(3); s[4] = S.c(4) + S.h(4); s[5] = S.f(1); s[6] = S.c(6) + S.h(5); s[7] = S.c(7) + S.h(6); }
  if constexpr (KIND == 1 && D == 1) { s[0] = S.c(0) - S.h(7); s[1] = S.c(1) - S.h(8); s[2] = S.c(2) - S.h(9); s[3] = S.c(3) - S.h(10); s[4] = S.c(4) - S.h(11); s[5] = S.c(5) - S.h(12); s[6] = S.f(2); s[7] = S.c(7) - S.h(13); }
  if constexpr (KIND == 0 && D == 1) { s[0] = S.c(0) + S.h(7); s[1] = S.c(1) + S.h(8); s[2] = S.c(2) + S.h(9); s[3] = S.c(3) + S.h(10); s[4] = S.c(4) + S.h(11); s[5] = S.c(5) + S.h(12); s[6] = S.f(3); s[7] = S.c(7) + S.h(13); }
  if constexpr (KIND == 1 && D == 2) { s[0] = S.c(0) - S.h(14); s[1] = S.c(1) - S.h(15); s[2] = S.c(2) - S.h(16); s[3] = S.c(3) - S.h(17); s[4] = S.c(4) - S.h(18); s[5] = S.c(5) - S.h(19); s[6] = S.c(6) - S.h(20); s[7] = S.f(4); }
  if constexpr (KIND == 0 && D == 2) { s[0] = S.c(0) + S.h(14); s[1] = S.c(1) + S.h(15); s[2] = S.c(2) + S.h(16); s[3] = S.c(3) + S.h(17); s[4] = S.c(4) + S.h(18); s[5] = S.c(5) + S.h(19); s[6] = S.c(6) + S.h(20); s[7] = S.f(5); }
  if constexpr (KIND == 2 && D == 0) { s[0] = S.c(0) + (+S.h(7) + S.h(14)); s[1] = S.c(1) + (+S.h(8) + S.h(15)); s[2] = S.c(2) + (+S.h(9) + S.h(16)); s[3] = S.c(3) + (+S.h(10) + S.h(17)); s[4] = S.c(4) + (+S.h(11) + S.h(18)); s[5] = S.c(5) + (+S.h(12) + S.h(19)); s[6] = S.f(3) + (+S.h(28)); s[7] = S.f(5) + (+S.h(32)); }
  if constexpr (KIND == 3 && D == 0) { s[0] = S.c(0) + (+S.h(7) - S.h(14)); s[1] = S.c(1) + (+S.h(8) - S.h(15)); s[2] = S.c(2) + (+S.h(9) - S.h(16)); s[3] = S.c(3) + (+S.h(10) - S.h(17)); s[4] = S.c(4) + (+S.h(11) - S.h(18)); s[5] = S.c(5) + (+S.h(12) - S.h(19)); s[6] = S.f(3) + (-S.h(28)); s[7] = S.f(4) + (+S.h(31)); }
  if constexpr (KIND == 4 && D == 0) { s[0] = S.c(0) + (-S.h(7) + S.h(14)); s[1] = S.c(1) + (-S.h(8) + S.h(15)); s[2] = S.c(2) + (-S.h(9) + S.h(16)); s[3] = S.c(3) + (-S.h(10) + S.h(17)); s[4] = S.c(4) + (-S.h(11) + S.h(18)); s[5] = S.c(5) + (-S.h(12) + S.h(19)); s[6] = S.f(2) + (+S.h(27)); s[7] = S.f(5) + (-S.h(32)); }
  if constexpr (KIND == 5 && D == 0) { s[0] = S.c(0) + (-S.h(7) - S.h(14)); s[1] = S.c(1) + (-S.h(8) - S.h(15)); s[2] = S.c(2) + (-S.h(9) - S.h(16)); s[3] = S.c(3) + (-S.h(10) - S.h(17)); s[4] = S.c(4) + (-S.h(11) - S.h(18)); s[5] = S.c(5) + (-S.h(12) - S.h(19)); s[6] = S.f(2) + (-S.h(27)); s[7] = S.f(4) + (-S.h(31)); }
  if constexpr (KIND == 2 && D == 1) { s[0] = S.c(0) + (+S.h(0) + S.h(14)); s[1] = S.c(1) + (+S.h(1) + S.h(15)); s[2] = S.c(2) + (+S.h(2) + S.h(16)); s[3] = S.c(3) + (+S.h(3) + S.h(17)); s[4] = S.c(4) + (+S.h(4) + S.h(18)); s[5] = S.f(1) + (+S.h(24)); s[6] = S.c(6) + (+S.h(5) + S.h(20)); s[7] = S.f(5) + (+S.h(30)); }
  if constexpr (KIND == 3 && D == 1) { s[0] = S.c(0) + (+S.h(0) - S.h(14)); s[1] = S.c(1) + (+S.h(1) - S.h(15)); s[2] = S.c(2) + (+S.h(2) - S.h(16)); s[3] = S.c(3) + (+S.h(3) - S.h(17)); s[4] = S.c(4) + (+S.h(4) - S.h(18)); s[5] = S.f(1) + (-S.h(24)); s[6] = S.c(6) + (+S.h(5) - S.h(20)); s[7] = S.f(4) + (+S.h(29)); }
  if constexpr (KIND == 4 && D == 1) { s[0] = S.c(0) + (-S.h(0) + S.h(14)); s[1] = S.c(1) + (-S.h(1) + S.h(15)); s[2] = S.c(2) + (-S.h(2) + S.h(16)); s[3] = S.c(3) + (-S.h(3) + S.h(17)); s[4] = S.c(4) + (-S.h(4) + S.h(18)); s[5] = S.f(0) + (+S.h(23)); s[6] = S.c(6) + (-S.h(5) + S.h(20)); s[7] = S.f(5) + (-S.h(30)); }
  if constexpr (KIND == 5 && D == 1) { s[0] = S.c(0) + (-S.h(0) - S.h(14)); s[1] = S.c(1) + (-S.h(1) - S.h(15)); s[2] = S.c(2) + (-S.h(2) - S.h(16)); s[3] = S.c(3) + (-S.h(3) - S.h(17)); s[4] = S.c(4) + (-S.h(4) - S.h(18)); s[5] = S.f(0) + (-S.h(23)); s[6] = S.c(6) + (-S.h(5) - S.h(20)); s[7] = S.f(4) + (-S.h(29)); }
  if constexpr (KIND == 2 && D == 2) { s[0] = S.c(0) + (+S.h(0) + S.h(7)); s[1] = S.c(1) + (+S.h(1) + S.h(8)); s[2] = S.c(2) + (+S.h(2) + S.h(9)); s[3] = S.c(3) + (+S.h(3) + S.h(10)); s[4] = S.c(4) + (+S.h(4) + S.h(11)); s[5] = S.f(1) + (+S.h(22)); s[6] = S.f(3) + (+S.h(26)); s[7] = S.c(7) + (+S.h(6) + S.h(13)); }
  if constexpr (KIND == 3 && D == 2) { s[0] = S.c(0) + (+S.h(0) - S.h(7)); s[1] = S.c(1) + (+S.h(1) - S.h(8)); s[2] = S.c(2) + (+S.h(2) - S.h(9)); s[3] = S.c(3) + (+S.h(3) - S.h(10)); s[4] = S.c(4) + (+S.h(4) - S.h(11)); s[5] = S.f(1) + (-S.h(22)); s[6] = S.f(2) + (+S.h(25)); s[7] = S.c(7) + (+S.h(6) - S.h(13)); }
  if constexpr (KIND == 4 && D == 2) { s[0] = S.c(0) + (-S.h(0) + S.h(7)); s[1] = S.c(1) + (-S.h(1) + S.h(8)); s[2] = S.c(2) + (-S.h(2) + S.h(9)); s[3] = S.c(3) + (-S.h(3) + S.h(10)); s[4] = S.c(4) + (-S.h(4) + S.h(11)); s[5] = S.f(0) + (+S.h(21)); s[6] = S.f(3) + (-S.h(26)); s[7] = S.c(7) + (-S.h(6) + S.h(13)); }
  if constexpr (KIND == 5 && D == 2) { s[0] = S.c(0) + (-S.h(0) - S.h(7)); s[1] = S.c(1) + (-S.h(1) - S.h(8)); s[2] = S.c(2) + (-S.h(2) - S.h(9)); s[3] = S.c(3) + (-S.h(3) - S.h(10)); s[4] = S.c(4) + (-S.h(4) - S.h(11)); s[5] = S.f(0) + (-S.h(21)); s[6] = S.f(2) + (-S.h(25)); s[7] = S.c(7) + (-S.h(6) - S.h(13)); }
  if (s[0] < smallr) s[0] = S.c(0);
  s[4] = fmax2(smallp, s[4]);
}

template <class Sink>
MHD_FN void trace3d_cell(const TraceIn &I, double dtdx, double dtdy, double dtdz, const MhdConst &P, Sink &O) {
  TracePred T;
  trace_predict(I, dtdx, dtdy, dtdz, P, T);
  double s[8];
#define MHD_PUT(K, D_) trace_state<K, D_>(T, P, s); O.put(K, D_, s);
  MHD_PUT(1, 0) MHD_PUT(0, 0) MHD_PUT(1, 1) MHD_PUT(0, 1) MHD_PUT(1, 2) MHD_PUT(0, 2)
  MHD_PUT(2, 0) MHD_PUT(3, 0) MHD_PUT(4, 0) MHD_PUT(5, 0)
  MHD_PUT(2, 1) MHD_PUT(3, 1) MHD_PUT(4, 1) MHD_PUT(5, 1)
  MHD_PUT(2, 2) MHD_PUT(3, 2) MHD_PUT(4, 2) MHD_PUT(5, 2)
#undef MHD_PUT
}

// ---- 1-D Riemann solvers (mhd/godunov_utils.f90) ----------------------------------------------------------------------
// find_mhd_flux :704-782: conservative variables cvar[0..8] and fluxes ff[0..8] (index 8 = the thermal energy)
MHD_FN void find_mhd_flux(const double (&qv)[8], double gamma, double (&cvar)[9], double (&ff)[9]) {
  const double entho = 1.0 / (gamma - 1.0);
  const double d = qv[0], P = qv[1], u = qv[2], A = qv[3], v = qv[4], B = qv[5], w = qv[6], C = qv[7];
  const double ecin = 0.5 * (u * u + v * v + w * w) * d;
  const double emag = 0.5 * (A * A + B * B + C * C);
  const double etot = P * entho + ecin + emag;
  const double Ptot = P + emag;
  cvar[0] = d; cvar[1] = etot; cvar[2] = d * u; cvar[3] = A; cvar[4] = d * v; cvar[5] = B; cvar[6] = d * w; cvar[7] = C;
  cvar[8] = P * entho;
  ff[0] = d * u;
  ff[1] = (etot + Ptot) * u - A * (A * u + B * v + C * w);
  ff[2] = d * u * u + Ptot - A * A;
  ff[3] = 0.0;
  ff[4] = d * u * v - A * B;
  ff[5] = B * u - A * v;
  ff[6] = d * u * w - A * C;
  ff[7] = C * u - A * w;
  ff[8] = P * entho * u;
}
// find_speed_fast :823-852 (find_speed_info :787-818 adds |u|)
MHD_FN double find_speed_fast(const double (&qv)[8], double gamma) {
  const double d = qv[0], P = qv[1], A = qv[3], B = qv[5], C = qv[7];
  const double B2 = A * A + B * B + C * C;
  const double c2 = gamma * P / d;
  const double d2 = 0.5 * (B2 / d + c2);
  return __builtin_sqrt(d2 + __builtin_sqrt(d2 * d2 - c2 * A * A / d));
}
MHD_FN double find_speed_info(const double (&qv)[8], double gamma) { return find_speed_fast(qv, gamma) + __builtin_fabs(qv[2]); }

// lax_friedrich :352-386
MHD_FN void lax_friedrich(double (&ql)[8], double (&qr)[8], double zero_flux, double gamma, double (&fg)[9]) {
  const double bx_mean = 0.5 * (ql[3] + qr[3]);
  ql[3] = bx_mean; qr[3] = bx_mean;
  double ul[9], fl[9], ur[9], fr[9];
  find_mhd_flux(ql, gamma, ul, fl);
  find_mhd_flux(qr, gamma, ur, fr);
  const double vleft = find_speed_info(ql, gamma), vright = find_speed_info(qr, gamma);
  const double vm = fmax2(vleft, vright);
  for (int n = 0; n < 9; n++) {
    const double fmean = 0.5 * (fr[n] + fl[n]) * zero_flux;
    const double udiff = 0.5 * (ur[n] - ul[n]);
    fg[n] = fmean - vm * udiff;
  }
}
// upwind :313-347 (the 2-D solver 'upwind' calls it; cmpflxm's own case 4 calls lax_friedrich, umuscl.f90:1409-1410)
MHD_FN void upwind(double (&ql)[8], double (&qr)[8], double zero_flux, double gamma, double (&fg)[9]) {
  const double bx_mean = 0.5 * (ql[3] + qr[3]);
  ql[3] = bx_mean; qr[3] = bx_mean;
  double ul[9], fl[9], ur[9], fr[9];
  find_mhd_flux(ql, gamma, ul, fl);
  find_mhd_flux(qr, gamma, ur, fr);
  const double vleft = 0.5 * (ql[2] + qr[2]);
  for (int n = 0; n < 9; n++) {
    const double fmean = 0.5 * (fr[n] + fl[n]) * zero_flux;
    const double udiff = 0.5 * (ur[n] - ul[n]);
    fg[n] = fmean - __builtin_fabs(vleft) * udiff;
  }
}
// find_speed_alfven :857-873
MHD_FN double find_speed_alfven(const double (&qv)[8]) { return __builtin_sqrt(qv[3] * qv[3] / qv[0]); }
// hll :391-421
MHD_FN void hll(double (&ql)[8], double (&qr)[8], double gamma, double (&fg)[9]) {
  const double bx_mean = 0.5 * (ql[3] + qr[3]);
  ql[3] = bx_mean; qr[3] = bx_mean;
  double ul[9], fl[9], ur[9], fr[9];
  find_mhd_flux(ql, gamma, ul, fl);
  find_mhd_flux(qr, gamma, ur, fr);
  const double cfl = find_speed_fast(ql, gamma), cfr = find_speed_fast(qr, gamma);
  const double vl = ql[2], vr = qr[2];
  const double SL = fmin2(fmin2(vl, vr) - fmax2(cfl, cfr), 0.0);
  const double SR = fmax2(fmax2(vl, vr) + fmax2(cfl, cfr), 0.0);
  for (int n = 0; n < 9; n++) fg[n] = (SR * fl[n] - SL * fr[n] + SR * SL * (ur[n] - ul[n])) / (SR - SL);
}
// hlld :426-699 (Miyoshi & Kusano 2005)
MHD_FN void hlld(double (&ql)[8], double (&qr)[8], double gamma, double (&fg)[9]) {
  const double entho = 1.0 / (gamma - 1.0);
  const double A = 0.5 * (ql[3] + qr[3]);
  const double sgnm = __builtin_copysign(1.0, A);
  ql[3] = A; qr[3] = A;
  const double rl = ql[0], Pl = ql[1], ul = ql[2], vl = ql[4], Bl = ql[5], wl = ql[6], Cl = ql[7];
  const double ecinl = 0.5 * (ul * ul + vl * vl + wl * wl) * rl;
  const double emagl = 0.5 * (A * A + Bl * Bl + Cl * Cl);
  const double etotl = Pl * entho + ecinl + emagl;
  const double Ptotl = Pl + emagl;
  const double vdotBl = ul * A + vl * Bl + wl * Cl;
  const double eintl = Pl * entho;
  const double rr = qr[0], Pr = qr[1], ur = qr[2], vr = qr[4], Br = qr[5], wr = qr[6], Cr = qr[7];
  const double ecinr = 0.5 * (ur * ur + vr * vr + wr * wr) * rr;
  const double emagr = 0.5 * (A * A + Br * Br + Cr * Cr);
  const double etotr = Pr * entho + ecinr + emagr;
  const double Ptotr = Pr + emagr;
  const double vdotBr = ur * A + vr * Br + wr * Cr;
  const double eintr = Pr * entho;
  const double cfastl = find_speed_fast(ql, gamma), cfastr = find_speed_fast(qr, gamma);
  const double SL = fmin2(ul, ur) - fmax2(cfastl, cfastr);
  const double SR = fmax2(ul, ur) + fmax2(cfastl, cfastr);
  const double rcl = rl * (ul - SL);
  const double rcr = rr * (SR - ur);
  const double ustar = (rcr * ur + rcl * ul + (Ptotl - Ptotr)) / (rcr + rcl);
  const double Ptotstar = (rcr * Ptotl + rcl * Ptotr + rcl * rcr * (ul - ur)) / (rcr + rcl);
  // left star region
  const double rstarl = rl * (SL - ul) / (SL - ustar);
  double estar = rl * (SL - ul) * (SL - ustar) - A * A;
  const double el = rl * (SL - ul) * (SL - ul) - A * A;
  const double eintstarl = eintl * (SL - ul) / (SL - ustar);
  double vstarl, Bstarl, wstarl, Cstarl;
  if (__builtin_fabs(estar) < (double)1e-4f * (A * A)) {
    vstarl = vl; Bstarl = Bl; wstarl = wl; Cstarl = Cl;
  } else {
    vstarl = vl - A * Bl * (ustar - ul) / estar;
    Bstarl = Bl * el / estar;
    wstarl = wl - A * Cl * (ustar - ul) / estar;
    Cstarl = Cl * el / estar;
  }
  const double vdotBstarl = ustar * A + vstarl * Bstarl + wstarl * Cstarl;
  const double etotstarl = ((SL - ul) * etotl - Ptotl * ul + Ptotstar * ustar + A * (vdotBl - vdotBstarl)) / (SL - ustar);
  const double sqrrstarl = __builtin_sqrt(rstarl);
  const double calfvenl = __builtin_fabs(A) / sqrrstarl;
  const double SAL = ustar - calfvenl;
  // right star region   ((double)1e-4f: the reference's literal 1e-4 is a default REAL)
  const double rstarr = rr * (SR - ur) / (SR - ustar);
  estar = rr * (SR - ur) * (SR - ustar) - A * A;
  const double er = rr * (SR - ur) * (SR - ur) - A * A;
  const double eintstarr = eintr * (SR - ur) / (SR - ustar);
  double vstarr, Bstarr, wstarr, Cstarr;
  if (__builtin_fabs(estar) < (double)1e-4f * (A * A)) {
    vstarr = vr; Bstarr = Br; wstarr = wr; Cstarr = Cr;
  } else {
    vstarr = vr - A * Br * (ustar - ur) / estar;
    Bstarr = Br * er / estar;
    wstarr = wr - A * Cr * (ustar - ur) / estar;
    Cstarr = Cr * er / estar;
  }
  const double vdotBstarr = ustar * A + vstarr * Bstarr + wstarr * Cstarr;
  const double etotstarr = ((SR - ur) * etotr - Ptotr * ur + Ptotstar * ustar + A * (vdotBr - vdotBstarr)) / (SR - ustar);
  const double sqrrstarr = __builtin_sqrt(rstarr);
  const double calfvenr = __builtin_fabs(A) / sqrrstarr;
  const double SAR = ustar + calfvenr;
  // double star region
  const double vstarstar = (sqrrstarl * vstarl + sqrrstarr * vstarr + sgnm * (Bstarr - Bstarl)) / (sqrrstarl + sqrrstarr);
  const double wstarstar = (sqrrstarl * wstarl + sqrrstarr * wstarr + sgnm * (Cstarr - Cstarl)) / (sqrrstarl + sqrrstarr);
  const double Bstarstar = (sqrrstarl * Bstarr + sqrrstarr * Bstarl + sgnm * sqrrstarl * sqrrstarr * (vstarr - vstarl)) / (sqrrstarl + sqrrstarr);
  const double Cstarstar = (sqrrstarl * Cstarr + sqrrstarr * Cstarl + sgnm * sqrrstarl * sqrrstarr * (wstarr - wstarl)) / (sqrrstarl + sqrrstarr);
  const double vdotBstarstar = ustar * A + vstarstar * Bstarstar + wstarstar * Cstarstar;
  const double etotstarstarl = etotstarl - sgnm * sqrrstarl * (vdotBstarl - vdotBstarstar);
  const double etotstarstarr = etotstarr + sgnm * sqrrstarr * (vdotBstarr - vdotBstarstar);
  // sample the solution at x/t = 0
  double ro, uo, vo, wo, Bo, Co, Ptoto, etoto, vdotBo, einto;
  if (SL > 0.0) {
    ro = rl; uo = ul; vo = vl; wo = wl; Bo = Bl; Co = Cl; Ptoto = Ptotl; etoto = etotl; vdotBo = vdotBl; einto = eintl;
  } else if (SAL > 0.0) {
    ro = rstarl; uo = ustar; vo = vstarl; wo = wstarl; Bo = Bstarl; Co = Cstarl; Ptoto = Ptotstar; etoto = etotstarl; vdotBo = vdotBstarl; einto = eintstarl;
  } else if (ustar > 0.0) {
    ro = rstarl; uo = ustar; vo = vstarstar; wo = wstarstar; Bo = Bstarstar; Co = Cstarstar; Ptoto = Ptotstar; etoto = etotstarstarl; vdotBo = vdotBstarstar; einto = eintstarl;
  } else if (SAR > 0.0) {
    ro = rstarr; uo = ustar; vo = vstarstar; wo = wstarstar; Bo = Bstarstar; Co = Cstarstar; Ptoto = Ptotstar; etoto = etotstarstarr; vdotBo = vdotBstarstar; einto = eintstarr;
  } else if (SR > 0.0) {
    ro = rstarr; uo = ustar; vo = vstarr; wo = wstarr; Bo = Bstarr; Co = Cstarr; Ptoto = Ptotstar; etoto = etotstarr; vdotBo = vdotBstarr; einto = eintstarr;
  } else {
    ro = rr; uo = ur; vo = vr; wo = wr; Bo = Br; Co = Cr; Ptoto = Ptotr; etoto = etotr; vdotBo = vdotBr; einto = eintr;
  }
  fg[0] = ro * uo;
  fg[1] = (etoto + Ptoto) * uo - A * vdotBo;
  fg[2] = ro * uo * uo + Ptoto - A * A;
  fg[3] = 0.0;
  fg[4] = ro * uo * vo - A * Bo;
  fg[5] = Bo * uo - A * vo;
  fg[6] = ro * uo * wo - A * Co;
  fg[7] = Co * uo - A * wo;
  fg[8] = uo * einto;
}

// eigenvalues :1207-1261 (MHD adiabatic eigenvalues of one state; athena_roe uses entries 1, 3, 5, 7 for its entropy fix)
MHD_FN void roe_eigenvalues(double d, double vx, double p, double bx, double by, double bz, double gamma, double smallc, double (&lambda)[7]) {
  const double btsq = by * by + bz * bz;
  const double vaxsq = bx * bx / d;
  const double vax = __builtin_sqrt(vaxsq);
  double asq = gamma * p / d;
  asq = fmax2(asq, smallc * smallc);
  const double astarsq = asq + vaxsq + btsq / d;
  const double disc = __builtin_sqrt(astarsq * astarsq - 4.0 * asq * vaxsq);
  const double cfsq = 0.5 * (astarsq + disc);
  const double cfast = __builtin_sqrt(cfsq);
  double cssq = 0.5 * (astarsq - disc);
  if (cssq <= 0.0) cssq = 0.0;
  const double cslow = __builtin_sqrt(cssq);
  lambda[0] = vx - cfast; lambda[1] = vx - vax; lambda[2] = vx - cslow; lambda[3] = vx;
  lambda[4] = vx + cslow; lambda[5] = vx + vax; lambda[6] = vx + cfast;
}
// eigen_cons :1266-1517: eigenvalues, right (rem[wave][component]) and left (lem[component][wave]) eigenmatrices of the
// Roe-averaged state in conserved variables (rho, rho vx, rho vy, rho vz, E, by, bz)
MHD_FN void roe_eigen_cons(double d, double vx, double vy, double vz, double h, double Bx, double by, double bz, double Xfac, double Yfac,
                           double gamma, double smallc, double (&lambda)[7], double (&rem)[7][7], double (&lem)[7][7]) {
  const double gm1 = gamma - 1.0, gm2 = gamma - 2.0;
  const double vsq = vx * vx + vy * vy + vz * vz;
  const double btsq = by * by + bz * bz;
  const double bt_starsq = (gm1 - gm2 * Yfac) * btsq;
  const double bt = __builtin_sqrt(btsq);
  const double bt_star = __builtin_sqrt(bt_starsq);
  const double vaxsq = Bx * Bx / d;
  const double vax = __builtin_sqrt(vaxsq);
  const double hp = h - (vaxsq + btsq / d);
  double twid_asq = (gm1 * (hp - 0.5 * vsq) - gm2 * Xfac);
  twid_asq = fmax2(twid_asq, smallc * smallc);
  const double q_starsq = twid_asq + (vaxsq + bt_starsq / d);
  const double disc = __builtin_sqrt(q_starsq * q_starsq - 4.0 * twid_asq * vaxsq);
  const double cfsq = 0.5 * (q_starsq + disc);
  const double cfast = __builtin_sqrt(cfsq);
  double cssq = 0.5 * (q_starsq - disc);
  if (cssq <= 0.0) cssq = 0.0;
  const double cslow = __builtin_sqrt(cssq);
  double beta_y, beta_z, beta_ystar, beta_zstar;
  if (bt == 0.0) {
    // the reference writes .5*sqrt(2.): default-REAL arithmetic, then promoted
    const double hs = (double)(0.5f * __builtin_sqrtf(2.0f));
    beta_y = hs; beta_z = hs; beta_ystar = hs; beta_zstar = hs;
  } else {
    beta_y = by / bt; beta_z = bz / bt; beta_ystar = by / bt_star; beta_zstar = bz / bt_star;
  }
  const double beta_starsq = beta_ystar * beta_ystar + beta_zstar * beta_zstar;
  const double vbeta = vy * beta_ystar + vz * beta_zstar;
  double alpha_f, alpha_s;
  if ((cfsq - cssq) == 0.0) { alpha_f = 1.0; alpha_s = 0.0; }
  else if ((twid_asq - cssq) <= 0.0) { alpha_f = 0.0; alpha_s = 1.0; }
  else if ((cfsq - twid_asq) <= 0.0) { alpha_f = 1.0; alpha_s = 0.0; }
  else {
    alpha_f = __builtin_sqrt((twid_asq - cssq) / (cfsq - cssq));
    alpha_s = __builtin_sqrt((cfsq - twid_asq) / (cfsq - cssq));
  }
  const double droot = __builtin_sqrt(d);
  const double s = __builtin_copysign(1.0, Bx);
  const double twid_a = __builtin_sqrt(twid_asq);
  double Qfast = s * cfast * alpha_f;
  double Qslow = s * cslow * alpha_s;
  const double af_prime = twid_a * alpha_f / droot;
  const double as_prime = twid_a * alpha_s / droot;
  const double Afpbb = af_prime * bt_star * beta_starsq;
  const double Aspbb = as_prime * bt_star * beta_starsq;
  lambda[0] = vx - cfast; lambda[1] = vx - vax; lambda[2] = vx - cslow; lambda[3] = vx;
  lambda[4] = vx + cslow; lambda[5] = vx + vax; lambda[6] = vx + cfast;
  // right eigenmatrix
  rem[0][0] = alpha_f;
  rem[0][1] = alpha_f * (vx - cfast);
  rem[0][2] = alpha_f * vy + Qslow * beta_ystar;
  rem[0][3] = alpha_f * vz + Qslow * beta_zstar;
  rem[0][4] = alpha_f * (hp - vx * cfast) + Qslow * vbeta + Aspbb;
  rem[0][5] = as_prime * beta_ystar;
  rem[0][6] = as_prime * beta_zstar;
  rem[1][0] = 0.0; rem[1][1] = 0.0;
  rem[1][2] = -beta_z;
  rem[1][3] = beta_y;
  rem[1][4] = -(vy * beta_z - vz * beta_y);
  rem[1][5] = -s * beta_z / droot;
  rem[1][6] = s * beta_y / droot;
  rem[2][0] = alpha_s;
  rem[2][1] = alpha_s * (vx - cslow);
  rem[2][2] = alpha_s * vy - Qfast * beta_ystar;
  rem[2][3] = alpha_s * vz - Qfast * beta_zstar;
  rem[2][4] = alpha_s * (hp - vx * cslow) - Qfast * vbeta - Afpbb;
  rem[2][5] = -af_prime * beta_ystar;
  rem[2][6] = -af_prime * beta_zstar;
  rem[3][0] = 1.0; rem[3][1] = vx; rem[3][2] = vy; rem[3][3] = vz;
  rem[3][4] = 0.5 * vsq + gm2 * Xfac / gm1;
  rem[3][5] = 0.0; rem[3][6] = 0.0;
  rem[4][0] = alpha_s;
  rem[4][1] = alpha_s * (vx + cslow);
  rem[4][2] = alpha_s * vy + Qfast * beta_ystar;
  rem[4][3] = alpha_s * vz + Qfast * beta_zstar;
  rem[4][4] = alpha_s * (hp + vx * cslow) + Qfast * vbeta - Afpbb;
  rem[4][5] = rem[2][5];
  rem[4][6] = rem[2][6];
  rem[5][0] = 0.0; rem[5][1] = 0.0;
  rem[5][2] = beta_z;
  rem[5][3] = -beta_y;
  rem[5][4] = -rem[1][4];
  rem[5][5] = rem[1][5];
  rem[5][6] = rem[1][6];
  rem[6][0] = alpha_f;
  rem[6][1] = alpha_f * (vx + cfast);
  rem[6][2] = alpha_f * vy - Qslow * beta_ystar;
  rem[6][3] = alpha_f * vz - Qslow * beta_zstar;
  rem[6][4] = alpha_f * (hp + vx * cfast) - Qslow * vbeta + Aspbb;
  rem[6][5] = rem[0][5];
  rem[6][6] = rem[0][6];
  // left eigenmatrix: some quantities normalised by 1/(2 a^2), some by (gamma-1)/(2 a^2)
  const double na = 0.5 / twid_asq;
  const double cff = na * alpha_f * cfast;
  const double css = na * alpha_s * cslow;
  Qfast = Qfast * na;
  Qslow = Qslow * na;
  const double af = na * af_prime * d;
  const double as = na * as_prime * d;
  const double Afpb = na * af_prime * bt_star;
  const double Aspb = na * as_prime * bt_star;
  alpha_f = gm1 * na * alpha_f;
  alpha_s = gm1 * na * alpha_s;
  const double Q_ystar = beta_ystar / beta_starsq;
  const double Q_zstar = beta_zstar / beta_starsq;
  const double vqstr = (vy * Q_ystar + vz * Q_zstar);
  const double norm = gm1 * 2.0 * na;
  lem[0][0] = alpha_f * (vsq - hp) + cff * (cfast + vx) - Qslow * vqstr - Aspb;
  lem[1][0] = -alpha_f * vx - cff;
  lem[2][0] = -alpha_f * vy + Qslow * Q_ystar;
  lem[3][0] = -alpha_f * vz + Qslow * Q_zstar;
  lem[4][0] = alpha_f;
  lem[5][0] = as * Q_ystar - alpha_f * by;
  lem[6][0] = as * Q_zstar - alpha_f * bz;
  lem[0][1] = 0.5 * (vy * beta_z - vz * beta_y);
  lem[1][1] = 0.0;
  lem[2][1] = -0.5 * beta_z;
  lem[3][1] = 0.5 * beta_y;
  lem[4][1] = 0.0;
  lem[5][1] = -0.5 * droot * beta_z * s;
  lem[6][1] = 0.5 * droot * beta_y * s;
  lem[0][2] = alpha_s * (vsq - hp) + css * (cslow + vx) + Qfast * vqstr + Afpb;
  lem[1][2] = -alpha_s * vx - css;
  lem[2][2] = -alpha_s * vy - Qfast * Q_ystar;
  lem[3][2] = -alpha_s * vz - Qfast * Q_zstar;
  lem[4][2] = alpha_s;
  lem[5][2] = -af * Q_ystar - alpha_s * by;
  lem[6][2] = -af * Q_zstar - alpha_s * bz;
  lem[0][3] = 1.0 - norm * (0.5 * vsq - gm2 * Xfac / gm1);
  lem[1][3] = norm * vx;
  lem[2][3] = norm * vy;
  lem[3][3] = norm * vz;
  lem[4][3] = -norm;
  lem[5][3] = norm * by;
  lem[6][3] = norm * bz;
  lem[0][4] = alpha_s * (vsq - hp) + css * (cslow - vx) - Qfast * vqstr + Afpb;
  lem[1][4] = -alpha_s * vx + css;
  lem[2][4] = -alpha_s * vy + Qfast * Q_ystar;
  lem[3][4] = -alpha_s * vz + Qfast * Q_zstar;
  lem[4][4] = alpha_s;
  lem[5][4] = lem[5][2];
  lem[6][4] = lem[6][2];
  lem[0][5] = -lem[0][1];
  lem[1][5] = 0.0;
  lem[2][5] = -lem[2][1];
  lem[3][5] = -lem[3][1];
  lem[4][5] = 0.0;
  lem[5][5] = lem[5][1];
  lem[6][5] = lem[6][1];
  lem[0][6] = alpha_f * (vsq - hp) + cff * (cfast - vx) + Qslow * vqstr - Aspb;
  lem[1][6] = -alpha_f * vx + cff;
  lem[2][6] = -alpha_f * vy - Qslow * Q_ystar;
  lem[3][6] = -alpha_f * vz - Qslow * Q_zstar;
  lem[4][6] = alpha_f;
  lem[5][6] = lem[5][0];
  lem[6][6] = lem[6][0];
}
// athena_roe :878-1087: the Roe flux with the entropy fix of the genuinely non-linear waves; falls back to the
// Lax-Friedrichs flux when an intermediate state has a negative density or thermal energy
MHD_FN void athena_roe(double (&ql)[8], double (&qr)[8], double zero_flux, double gamma, double smallc, double (&fg)[9]) {
  const double bx_mean = 0.5 * (ql[3] + qr[3]);
  ql[3] = bx_mean; qr[3] = bx_mean;
  double ul[9], fl[9], ur[9], fr[9];
  find_mhd_flux(ql, gamma, ul, fl);
  find_mhd_flux(qr, gamma, ur, fr);
  const double dl = ql[0], dr = qr[0], pl = ql[1], pr = qr[1], vxl = ql[2], vxr = qr[2];
  const double vyl = ql[4], vyr = qr[4], byl = ql[5], byr = qr[5], vzl = ql[6], vzr = qr[6], bzl = ql[7], bzr = qr[7];
  const double bx = 0.5 * (ql[3] + qr[3]);
  const double el = ul[1], er = ur[1], mxl = ul[2], mxr = ur[2], myl = ul[4], myr = ur[4], mzl = ul[6], mzr = ur[6];
  const double pbl = 0.5 * (bx * bx + byl * byl + bzl * bzl);
  const double pbr = 0.5 * (bx * bx + byr * byr + bzr * bzr);
  const double hl = (el + pl + pbl) / dl, hr = (er + pr + pbr) / dr;
  const double sqrtdl = __builtin_sqrt(dl), sqrtdr = __builtin_sqrt(dr);
  const double droe = sqrtdl * sqrtdr;
  const double vxroe = (sqrtdl * vxl + sqrtdr * vxr) / (sqrtdl + sqrtdr);
  const double vyroe = (sqrtdl * vyl + sqrtdr * vyr) / (sqrtdl + sqrtdr);
  const double vzroe = (sqrtdl * vzl + sqrtdr * vzr) / (sqrtdl + sqrtdr);
  const double byroe = (sqrtdr * byl + sqrtdl * byr) / (sqrtdl + sqrtdr);
  const double bzroe = (sqrtdr * bzl + sqrtdl * bzr) / (sqrtdl + sqrtdr);
  const double hroe = (sqrtdl * hl + sqrtdr * hr) / (sqrtdl + sqrtdr);
  const double Xfactor = ((byroe * byroe - byl * byr) + (bzroe * bzroe - bzl * bzr)) / (2.0 * droe);
  const double Yfactor = (dl + dr) / (2.0 * droe);
  double lambda[7], lambdal[7], lambdar[7], rem[7][7], lem[7][7], a[7];
  roe_eigen_cons(droe, vxroe, vyroe, vzroe, hroe, bx, byroe, bzroe, Xfactor, Yfactor, gamma, smallc, lambda, rem, lem);
  roe_eigenvalues(dl, vxl, pl, bx, byl, bzl, gamma, smallc, lambdal);
  roe_eigenvalues(dr, vxr, pr, bx, byr, bzr, gamma, smallc, lambdar);
  for (int n = 0; n < 7; n++) {
    double an = 0.0;
    an = an + (dr - dl) * lem[0][n];
    an = an + (mxr - mxl) * lem[1][n];
    an = an + (myr - myl) * lem[2][n];
    an = an + (mzr - mzl) * lem[3][n];
    an = an + (er - el) * lem[4][n];
    an = an + (byr - byl) * lem[5][n];
    an = an + (bzr - bzl) * lem[6][n];
    a[n] = an;
  }
  bool llf = false;
  double dim = dl, mxm = mxl, mym = myl, mzm = mzl, eim = el, bym = byl, bzm = bzl;
  for (int n = 0; n < 7; n++) {
    dim = dim + a[n] * rem[n][0];
    mxm = mxm + a[n] * rem[n][1];
    mym = mym + a[n] * rem[n][2];
    mzm = mzm + a[n] * rem[n][3];
    eim = eim + a[n] * rem[n][4];
    bym = bym + a[n] * rem[n][5];
    bzm = bzm + a[n] * rem[n][6];
    const double etm = eim - 0.5 * (mxm * mxm + mym * mym + mzm * mzm) / dim - 0.5 * (bx * bx + bym * bym + bzm * bzm);
    if (dim <= 0.0 || etm <= 0.0) llf = true;
  }
  if (llf) {
    const double vleft = find_speed_info(ql, gamma), vright = find_speed_info(qr, gamma);
    const double vm = fmax2(vleft, vright);
    for (int n = 0; n < 9; n++) {
      const double fmean = 0.5 * (fr[n] + fl[n]) * zero_flux;
      const double udiff = 0.5 * (ur[n] - ul[n]);
      fg[n] = fmean - vm * udiff;
    }
    return;
  }
  for (int n = 0; n < 7; n += 2) {
    const double l1 = fmin2(lambdal[n], lambda[n]);
    const double l2 = fmax2(lambdar[n], lambda[n]);
    if (l1 < 0.0 && l2 > 0.0) lambda[n] = (lambda[n] * (l2 + l1) - 2.0 * l2 * l1) / (l2 - l1);
  }
  double fluxd = fl[0] * zero_flux + fr[0] * zero_flux;
  double fluxe = fl[1] * zero_flux + fr[1] * zero_flux;
  double fluxmx = fl[2] * zero_flux + fr[2] * zero_flux;
  double fluxmy = fl[4] * zero_flux + fr[4] * zero_flux;
  double fluxby = fl[5] * zero_flux + fr[5] * zero_flux;
  double fluxmz = fl[6] * zero_flux + fr[6] * zero_flux;
  double fluxbz = fl[7] * zero_flux + fr[7] * zero_flux;
  for (int n = 0; n < 7; n++) {
    const double coef = __builtin_fabs(lambda[n]) * a[n];
    fluxd = fluxd - coef * rem[n][0];
    fluxe = fluxe - coef * rem[n][4];
    fluxmx = fluxmx - coef * rem[n][1];
    fluxmy = fluxmy - coef * rem[n][2];
    fluxby = fluxby - coef * rem[n][5];
    fluxmz = fluxmz - coef * rem[n][3];
    fluxbz = fluxbz - coef * rem[n][6];
  }
  fg[0] = 0.5 * fluxd; fg[1] = 0.5 * fluxe; fg[2] = 0.5 * fluxmx; fg[3] = 0.0; fg[4] = 0.5 * fluxmy; fg[5] = 0.5 * fluxby;
  fg[6] = 0.5 * fluxmz; fg[7] = 0.5 * fluxbz; fg[8] = 0.0;
}

// hydro_acoustic :1092-1201 (riemann = 'hydro': the acoustic hydro solver on density, pressure, normal velocity; the other
// components ride with the contact)
MHD_FN void hydro_acoustic(double (&ql)[8], double (&qr)[8], double gamma, double smallr, double smallc, double (&fg)[9]) {
  const double smallp = smallr * (smallc * smallc);
  const double bx_mean = 0.5 * (ql[3] + qr[3]);
  ql[3] = bx_mean; qr[3] = bx_mean;
  const double rl = fmax2(ql[0], smallr), rr = fmax2(qr[0], smallr);
  const double pl = fmax2(ql[1], smallp), pr = fmax2(qr[1], smallp);
  const double ul = ql[2], ur = qr[2];
  const double cl = __builtin_sqrt(gamma * pl / rl), cr = __builtin_sqrt(gamma * pr / rr);
  const double wl = cl * rl, wr = cr * rr;
  const double pstar = ((wr * pl + wl * pr) + wl * wr * (ul - ur)) / (wl + wr);
  const double ustar = ((wr * ur + wl * ul) + (pl - pr)) / (wl + wr);
  const double sgnm = __builtin_copysign(1.0, ustar);
  const bool left = sgnm == 1.0;
  const double ro = left ? rl : rr, uo = left ? ul : ur, po = left ? pl : pr, co = left ? cl : cr;
  double rstar = ro + (pstar - po) / (co * co);
  rstar = fmax2(rstar, smallr);
  double cstar = __builtin_sqrt(__builtin_fabs(gamma * pstar / rstar));
  cstar = fmax2(cstar, smallc);
  double spout = co - sgnm * uo;
  double spin = cstar - sgnm * ustar;
  double ushock = 0.5 * (spin + spout);
  ushock = fmax2(ushock, -sgnm * ustar);
  if (pstar >= po) { spout = ushock; spin = spout; }
  double qg[8];
  if (spout < 0.0) { qg[0] = ro; qg[1] = po; qg[2] = uo; }
  else if (spin >= 0.0) { qg[0] = rstar; qg[1] = pstar; qg[2] = ustar; }
  else {
    const double frac = spout / (spout - spin);
    qg[0] = frac * rstar + (1.0 - frac) * ro;
    qg[1] = frac * pstar + (1.0 - frac) * po;
    qg[2] = frac * ustar + (1.0 - frac) * uo;
  }
  for (int n = 3; n < 8; n++) qg[n] = left ? ql[n] : qr[n];
  double ug[9];
  find_mhd_flux(qg, gamma, ug, fg);
}

MHD_FN bool riemann_supported(int r) { return r >= RIEMANN_LLF && r <= RIEMANN_HYDRO; }
MHD_FN bool riemann2d_supported(int r) { return r >= RIEMANN2D_LLF && r <= RIEMANN2D_HLLD; }

// ---- cmpflxm :1308-1448, one face of direction d (0,1,2): qm_ = the state on the +d face of the cell below, qp_ = the state
// on the -d face of the cell above; flx[0..7] in the reference's variable order (1..8), before the dt/dx scaling ----------
// (RS = -1: P.riemann decides at run time; RS = -2: the same without the Roe solver, whose eigenmatrices would cost every
// solver its registers -- the device's general kernel; RS >= 0: the solver is a compile-time constant)
template <int RS = -1>
MHD_FN void cmpflxm_face(const double (&qm_)[8], const double (&qp_)[8], int d, const MhdConst &P, double (&flx)[8]) {
  // ln, lt1, lt2 / bn, bt1, bt2 (0-based) of mag_unsplit's three calls (:96-98, :118-120, :142-144)
  const int ln = 1 + d, lt1 = d == 0 ? 2 : 1, lt2 = d == 2 ? 2 : 3;
  const int bn = 5 + d, bt1 = d == 0 ? 6 : 5, bt2 = d == 2 ? 6 : 7;
  const double bn_mean = 0.5 * (qm_[bn] + qp_[bn]);
  double ql[8] = {qm_[0], qm_[4], qm_[ln], bn_mean, qm_[lt1], qm_[bt1], qm_[lt2], qm_[bt2]};
  double qr[8] = {qp_[0], qp_[4], qp_[ln], bn_mean, qp_[lt1], qp_[bt1], qp_[lt2], qp_[bt2]};
  double fg[9];
  switch (RS >= 0 ? RS : P.riemann) {
    case RIEMANN_HLL: hll(ql, qr, P.gamma, fg); break;
    case RIEMANN_HLLD: hlld(ql, qr, P.gamma, fg); break;
    case RIEMANN_HYDRO: hydro_acoustic(ql, qr, P.gamma, P.smallr, P.smallc, fg); break;
    case RIEMANN_ROE: if (RS != -2) { athena_roe(ql, qr, 1.0, P.gamma, P.smallc, fg); break; }
    default: lax_friedrich(ql, qr, 1.0, P.gamma, fg); break;     // llf (0) and 'upwind' (4, cmpflxm :1409-1410)
  }
  flx[0] = fg[0]; flx[4] = fg[1]; flx[ln] = fg[2]; flx[bn] = fg[3]; flx[lt1] = fg[4]; flx[bt1] = fg[5]; flx[lt2] = fg[6]; flx[bt2] = fg[7];
}

// ---- cmp_mag_flx :1453-2028, one edge of direction e (0,1,2 = x, y, z).  The four states around the edge, as the routine
// names them AFTER its dummy-argument shuffle: RT_, RB_, LT_, LB_ (which cell's qRT / qRB / qLT / qLB each one is depends on
// the direction: mag_unsplit :165-169, :199-203, :216-220 -- the caller knows) ---------------------------------------------
template <int R2 = -1>
MHD_FN double cmp_mag_flx_edge(const double (&RT_)[8], const double (&RB_)[8], const double (&LT_)[8], const double (&LB_)[8], int e,
                               const MhdConst &P) {
  const int riemann2d = R2 >= 0 ? R2 : P.riemann2d;
  // lp1, lp2, lor / bp1, bp2, bor (0-based) of the three calls: z: 2,3,4,6,7,8; y: 4,2,3,8,6,7; x: 3,4,2,7,8,6
  const int lp1 = e == 2 ? 1 : (e == 1 ? 3 : 2), lp2 = e == 2 ? 2 : (e == 1 ? 1 : 3), lor = e == 2 ? 3 : (e == 1 ? 2 : 1);
  const int bp1 = lp1 + 4, bp2 = lp2 + 4, bor = lor + 4;
  double qLL[8], qRL[8], qLR[8], qRR[8];   // [0] rho [1] P [2] v_p1 [3] v_p2 [4] v_or [5] B_p1 [6] B_p2 [7] B_or
  qLL[0] = RT_[0]; qRL[0] = LT_[0]; qLR[0] = RB_[0]; qRR[0] = LB_[0];
  qLL[1] = RT_[4]; qRL[1] = LT_[4]; qLR[1] = RB_[4]; qRR[1] = LB_[4];
  qLL[2] = RT_[lp1]; qRL[2] = LT_[lp1]; qLR[2] = RB_[lp1]; qRR[2] = LB_[lp1];
  qLL[3] = RT_[lp2]; qRL[3] = LT_[lp2]; qLR[3] = RB_[lp2]; qRR[3] = LB_[lp2];
  qLL[5] = 0.5 * (RT_[bp1] + LT_[bp1]); qRL[5] = 0.5 * (RT_[bp1] + LT_[bp1]);
  qLR[5] = 0.5 * (RB_[bp1] + LB_[bp1]); qRR[5] = 0.5 * (RB_[bp1] + LB_[bp1]);
  qLL[6] = 0.5 * (RT_[bp2] + RB_[bp2]); qRL[6] = 0.5 * (LT_[bp2] + LB_[bp2]);
  qLR[6] = 0.5 * (RT_[bp2] + RB_[bp2]); qRR[6] = 0.5 * (LT_[bp2] + LB_[bp2]);
  qLL[4] = RT_[lor]; qRL[4] = LT_[lor]; qLR[4] = RB_[lor]; qRR[4] = LB_[lor];
  qLL[7] = RT_[bor]; qRL[7] = LT_[bor]; qLR[7] = RB_[bor]; qRR[7] = LB_[bor];
  const double gamma = P.gamma;
  double ELL = qLL[2] * qLL[6] - qLL[3] * qLL[5];
  double ERL = qRL[2] * qRL[6] - qRL[3] * qRL[5];
  double ELR = qLR[2] * qLR[6] - qLR[3] * qLR[5];
  double ERR = qRR[2] * qRR[6] - qRR[3] * qRR[5];
  // the 1-D states find_speed_fast sees: relative to x (p1 normal) and to y (p2 normal)
  auto tmpx = [](const double (&s)[8], double (&t)[8]) { t[0] = s[0]; t[1] = s[1]; t[6] = s[4]; t[7] = s[7]; t[2] = s[2]; t[3] = s[5]; t[4] = s[3]; t[5] = s[6]; };
  auto tmpy = [](const double (&s)[8], double (&t)[8]) { t[0] = s[0]; t[1] = s[1]; t[6] = s[4]; t[7] = s[7]; t[2] = s[3]; t[3] = s[6]; t[4] = s[2]; t[5] = s[5]; };
  if (riemann2d == RIEMANN2D_HLLD) {
    const double rLL = qLL[0], pLL = qLL[1], uLL = qLL[2], vLL = qLL[3], ALL = qLL[5], BLL = qLL[6], CLL = qLL[7];
    const double rLR = qLR[0], pLR = qLR[1], uLR = qLR[2], vLR = qLR[3], ALR = qLR[5], BLR = qLR[6], CLR = qLR[7];
    const double rRL = qRL[0], pRL = qRL[1], uRL = qRL[2], vRL = qRL[3], ARL = qRL[5], BRL = qRL[6], CRL = qRL[7];
    const double rRR = qRR[0], pRR = qRR[1], uRR = qRR[2], vRR = qRR[3], ARR = qRR[5], BRR = qRR[6], CRR = qRR[7];
    double t[8];
    tmpx(qLL, t); const double cfastLLx = find_speed_fast(t, gamma);
    tmpx(qLR, t); const double cfastLRx = find_speed_fast(t, gamma);
    tmpx(qRL, t); const double cfastRLx = find_speed_fast(t, gamma);
    tmpx(qRR, t); const double cfastRRx = find_speed_fast(t, gamma);
    tmpy(qLL, t); const double cfastLLy = find_speed_fast(t, gamma);
    tmpy(qLR, t); const double cfastLRy = find_speed_fast(t, gamma);
    tmpy(qRL, t); const double cfastRLy = find_speed_fast(t, gamma);
    tmpy(qRR, t); const double cfastRRy = find_speed_fast(t, gamma);
    const double SL = fmin4(uLL, uLR, uRL, uRR) - fmax4(cfastLLx, cfastLRx, cfastRLx, cfastRRx);
    const double SR = fmax4(uLL, uLR, uRL, uRR) + fmax4(cfastLLx, cfastLRx, cfastRLx, cfastRRx);
    const double SB = fmin4(vLL, vLR, vRL, vRR) - fmax4(cfastLLy, cfastLRy, cfastRLy, cfastRRy);
    const double ST = fmax4(vLL, vLR, vRL, vRR) + fmax4(cfastLLy, cfastLRy, cfastRLy, cfastRRy);
    ELL = uLL * BLL - vLL * ALL;
    ELR = uLR * BLR - vLR * ALR;
    ERL = uRL * BRL - vRL * ARL;
    ERR = uRR * BRR - vRR * ARR;
    const double PtotLL = pLL + 0.5 * (ALL * ALL + BLL * BLL + CLL * CLL);
    const double PtotLR = pLR + 0.5 * (ALR * ALR + BLR * BLR + CLR * CLR);
    const double PtotRL = pRL + 0.5 * (ARL * ARL + BRL * BRL + CRL * CRL);
    const double PtotRR = pRR + 0.5 * (ARR * ARR + BRR * BRR + CRR * CRR);
    const double rcLLx = rLL * (uLL - SL), rcRLx = rRL * (SR - uRL);
    const double rcLRx = rLR * (uLR - SL), rcRRx = rRR * (SR - uRR);
    const double rcLLy = rLL * (vLL - SB), rcLRy = rLR * (ST - vLR);
    const double rcRLy = rRL * (vRL - SB), rcRRy = rRR * (ST - vRR);
    const double ustar = (rcLLx * uLL + rcLRx * uLR + rcRLx * uRL + rcRRx * uRR + (PtotLL - PtotRL + PtotLR - PtotRR)) / (rcLLx + rcLRx + rcRLx + rcRRx);
    const double vstar = (rcLLy * vLL + rcLRy * vLR + rcRLy * vRL + rcRRy * vRR + (PtotLL - PtotLR + PtotRL - PtotRR)) / (rcLLy + rcLRy + rcRLy + rcRRy);
    const double rstarLLx = rLL * (SL - uLL) / (SL - ustar), BstarLL = BLL * (SL - uLL) / (SL - ustar);
    const double rstarLLy = rLL * (SB - vLL) / (SB - vstar), AstarLL = ALL * (SB - vLL) / (SB - vstar);
    const double rstarLL = rLL * (SL - uLL) / (SL - ustar) * (SB - vLL) / (SB - vstar);
    const double EstarLLx = ustar * BstarLL - vLL * ALL;
    const double EstarLLy = uLL * BLL - vstar * AstarLL;
    const double EstarLL = ustar * BstarLL - vstar * AstarLL;
    const double rstarLRx = rLR * (SL - uLR) / (SL - ustar), BstarLR = BLR * (SL - uLR) / (SL - ustar);
    const double rstarLRy = rLR * (ST - vLR) / (ST - vstar), AstarLR = ALR * (ST - vLR) / (ST - vstar);
    const double rstarLR = rLR * (SL - uLR) / (SL - ustar) * (ST - vLR) / (ST - vstar);
    const double EstarLRx = ustar * BstarLR - vLR * ALR;
    const double EstarLRy = uLR * BLR - vstar * AstarLR;
    const double EstarLR = ustar * BstarLR - vstar * AstarLR;
    const double rstarRLx = rRL * (SR - uRL) / (SR - ustar), BstarRL = BRL * (SR - uRL) / (SR - ustar);
    const double rstarRLy = rRL * (SB - vRL) / (SB - vstar), AstarRL = ARL * (SB - vRL) / (SB - vstar);
    const double rstarRL = rRL * (SR - uRL) / (SR - ustar) * (SB - vRL) / (SB - vstar);
    const double EstarRLx = ustar * BstarRL - vRL * ARL;
    const double EstarRLy = uRL * BRL - vstar * AstarRL;
    const double EstarRL = ustar * BstarRL - vstar * AstarRL;
    const double rstarRRx = rRR * (SR - uRR) / (SR - ustar), BstarRR = BRR * (SR - uRR) / (SR - ustar);
    const double rstarRRy = rRR * (ST - vRR) / (ST - vstar), AstarRR = ARR * (ST - vRR) / (ST - vstar);
    const double rstarRR = rRR * (SR - uRR) / (SR - ustar) * (ST - vRR) / (ST - vstar);
    const double EstarRRx = ustar * BstarRR - vRR * ARR;
    const double EstarRRy = uRR * BRR - vstar * AstarRR;
    const double EstarRR = ustar * BstarRR - vstar * AstarRR;
    const double smallc = P.smallc;
    const double calfvenL = fmax2(fmax4(__builtin_fabs(ALR) / __builtin_sqrt(rstarLRx), __builtin_fabs(AstarLR) / __builtin_sqrt(rstarLR),
                                        __builtin_fabs(ALL) / __builtin_sqrt(rstarLLx), __builtin_fabs(AstarLL) / __builtin_sqrt(rstarLL)), smallc);
    const double calfvenR = fmax2(fmax4(__builtin_fabs(ARR) / __builtin_sqrt(rstarRRx), __builtin_fabs(AstarRR) / __builtin_sqrt(rstarRR),
                                        __builtin_fabs(ARL) / __builtin_sqrt(rstarRLx), __builtin_fabs(AstarRL) / __builtin_sqrt(rstarRL)), smallc);
    const double calfvenB = fmax2(fmax4(__builtin_fabs(BLL) / __builtin_sqrt(rstarLLy), __builtin_fabs(BstarLL) / __builtin_sqrt(rstarLL),
                                        __builtin_fabs(BRL) / __builtin_sqrt(rstarRLy), __builtin_fabs(BstarRL) / __builtin_sqrt(rstarRL)), smallc);
    const double calfvenT = fmax2(fmax4(__builtin_fabs(BLR) / __builtin_sqrt(rstarLRy), __builtin_fabs(BstarLR) / __builtin_sqrt(rstarLR),
                                        __builtin_fabs(BRR) / __builtin_sqrt(rstarRRy), __builtin_fabs(BstarRR) / __builtin_sqrt(rstarRR)), smallc);
    const double SAL = fmin2(ustar - calfvenL, 0.0), SAR = fmax2(ustar + calfvenR, 0.0);
    const double SAB = fmin2(vstar - calfvenB, 0.0), SAT = fmax2(vstar + calfvenT, 0.0);
    const double AstarT = (SAR * AstarRR - SAL * AstarLR) / (SAR - SAL), AstarB = (SAR * AstarRL - SAL * AstarLL) / (SAR - SAL);
    const double BstarR = (SAT * BstarRR - SAB * BstarRL) / (SAT - SAB), BstarL = (SAT * BstarLR - SAB * BstarLL) / (SAT - SAB);
    double E;
    if (SB > 0.0) {
      if (SL > 0.0) E = ELL;
      else if (SR < 0.0) E = ERL;
      else E = (SAR * EstarLLx - SAL * EstarRLx + SAR * SAL * (BRL - BLL)) / (SAR - SAL);
    } else if (ST < 0.0) {
      if (SL > 0.0) E = ELR;
      else if (SR < 0.0) E = ERR;
      else E = (SAR * EstarLRx - SAL * EstarRRx + SAR * SAL * (BRR - BLR)) / (SAR - SAL);
    } else if (SL > 0.0) {
      E = (SAT * EstarLLy - SAB * EstarLRy - SAT * SAB * (ALR - ALL)) / (SAT - SAB);
    } else if (SR < 0.0) {
      E = (SAT * EstarRLy - SAB * EstarRRy - SAT * SAB * (ARR - ARL)) / (SAT - SAB);
    } else {
      E = (SAL * SAB * EstarRR - SAL * SAT * EstarRL - SAR * SAB * EstarLR + SAR * SAT * EstarLL) / (SAR - SAL) / (SAT - SAB)
          - SAT * SAB / (SAT - SAB) * (AstarT - AstarB) + SAR * SAL / (SAR - SAL) * (BstarR - BstarL);
    }
    return E;      // (allow_switch_solver2D = .false., the default: :1771-1777)
  }
  if (riemann2d == RIEMANN2D_HLL) {
    double t[8];
    tmpx(qLL, t); const double vLLx = t[2], cLLx = find_speed_fast(t, gamma);
    tmpx(qLR, t); const double vLRx = t[2], cLRx = find_speed_fast(t, gamma);
    tmpx(qRL, t); const double vRLx = t[2], cRLx = find_speed_fast(t, gamma);
    tmpx(qRR, t); const double vRRx = t[2], cRRx = find_speed_fast(t, gamma);
    tmpy(qLL, t); const double vLLy = t[2], cLLy = find_speed_fast(t, gamma);
    tmpy(qLR, t); const double vLRy = t[2], cLRy = find_speed_fast(t, gamma);
    tmpy(qRL, t); const double vRLy = t[2], cRLy = find_speed_fast(t, gamma);
    tmpy(qRR, t); const double vRRy = t[2], cRRy = find_speed_fast(t, gamma);
    const double SL = fmin2(fmin4(vLLx, vLRx, vRLx, vRRx) - fmax4(cLLx, cLRx, cRLx, cRRx), 0.0);
    const double SR = fmax2(fmax4(vLLx, vLRx, vRLx, vRRx) + fmax4(cLLx, cLRx, cRLx, cRRx), 0.0);
    const double SB = fmin2(fmin4(vLLy, vLRy, vRLy, vRRy) - fmax4(cLLy, cLRy, cRLy, cRRy), 0.0);
    const double ST = fmax2(fmax4(vLLy, vLRy, vRLy, vRRy) + fmax4(cLLy, cLRy, cRLy, cRRy), 0.0);
    return (SL * SB * ERR - SL * ST * ERL - SR * SB * ELR + SR * ST * ELL) / (SR - SL) / (ST - SB)
           - ST * SB / (ST - SB) * (qRR[5] - qLL[5]) + SR * SL / (SR - SL) * (qRR[6] - qLL[6]);
  }
  if (riemann2d == RIEMANN2D_HLLA) {
    // :1859-1896: the HLL formula with the Alfven speeds of the four states
    double t[8];
    tmpx(qLL, t); const double vLLx = t[2], cLLx = find_speed_alfven(t);
    tmpx(qLR, t); const double vLRx = t[2], cLRx = find_speed_alfven(t);
    tmpx(qRL, t); const double vRLx = t[2], cRLx = find_speed_alfven(t);
    tmpx(qRR, t); const double vRRx = t[2], cRRx = find_speed_alfven(t);
    tmpy(qLL, t); const double vLLy = t[2], cLLy = find_speed_alfven(t);
    tmpy(qLR, t); const double vLRy = t[2], cLRy = find_speed_alfven(t);
    tmpy(qRL, t); const double vRLy = t[2], cRLy = find_speed_alfven(t);
    tmpy(qRR, t); const double vRRy = t[2], cRRy = find_speed_alfven(t);
    const double SL = fmin2(fmin4(vLLx, vLRx, vRLx, vRRx) - fmax4(cLLx, cLRx, cRLx, cRRx), 0.0);
    const double SR = fmax2(fmax4(vLLx, vLRx, vRLx, vRRx) + fmax4(cLLx, cLRx, cRLx, cRRx), 0.0);
    const double SB = fmin2(fmin4(vLLy, vLRy, vRLy, vRRy) - fmax4(cLLy, cLRy, cRLy, cRRy), 0.0);
    const double ST = fmax2(fmax4(vLLy, vLRy, vRLy, vRRy) + fmax4(cLLy, cLRy, cRLy, cRRy), 0.0);
    return (SL * SB * ERR - SL * ST * ERL - SR * SB * ELR + SR * ST * ELL) / (SR - SL) / (ST - SB)
           - ST * SB / (ST - SB) * (qRR[5] - qLL[5]) + SR * SL / (SR - SL) * (qRR[6] - qLL[6]);
  }
  // llf (iriemann2d = 0) and upwind (2): the mean of the four edge values plus the diffusive terms of two 1-D solves (:1898-2021)
  const bool up = riemann2d == RIEMANN2D_UPWIND, roe = R2 != -2 && riemann2d == RIEMANN2D_ROE;
  const double E = 0.25 * (ELL + ERL + ELR + ERR);
  double ql[8], qr[8], fx[9], fy[9];
  ql[0] = 0.5 * (qLL[0] + qLR[0]); qr[0] = 0.5 * (qRR[0] + qRL[0]);
  ql[1] = 0.5 * (qLL[1] + qLR[1]); qr[1] = 0.5 * (qRR[1] + qRL[1]);
  ql[2] = 0.5 * (qLL[2] + qLR[2]); qr[2] = 0.5 * (qRR[2] + qRL[2]);
  ql[3] = 0.5 * (qLL[5] + qLR[5]); qr[3] = 0.5 * (qRR[5] + qRL[5]);
  ql[4] = 0.5 * (qLL[3] + qLR[3]); qr[4] = 0.5 * (qRR[3] + qRL[3]);
  ql[5] = 0.5 * (qLL[6] + qLR[6]); qr[5] = 0.5 * (qRR[6] + qRL[6]);
  ql[6] = 0.5 * (qLL[4] + qLR[4]); qr[6] = 0.5 * (qRR[4] + qRL[4]);
  ql[7] = 0.5 * (qLL[7] + qLR[7]); qr[7] = 0.5 * (qRR[7] + qRL[7]);
  if (roe) athena_roe(ql, qr, 0.0, gamma, P.smallc, fx); else if (up) upwind(ql, qr, 0.0, gamma, fx); else lax_friedrich(ql, qr, 0.0, gamma, fx);
  ql[0] = 0.5 * (qLL[0] + qRL[0]); qr[0] = 0.5 * (qRR[0] + qLR[0]);
  ql[1] = 0.5 * (qLL[1] + qRL[1]); qr[1] = 0.5 * (qRR[1] + qLR[1]);
  ql[2] = 0.5 * (qLL[3] + qRL[3]); qr[2] = 0.5 * (qRR[3] + qLR[3]);
  ql[3] = 0.5 * (qLL[6] + qRL[6]); qr[3] = 0.5 * (qRR[6] + qLR[6]);
  ql[4] = 0.5 * (qLL[2] + qRL[2]); qr[4] = 0.5 * (qRR[2] + qLR[2]);
  ql[5] = 0.5 * (qLL[5] + qRL[5]); qr[5] = 0.5 * (qRR[5] + qLR[5]);
  ql[6] = 0.5 * (qLL[4] + qRL[4]); qr[6] = 0.5 * (qRR[4] + qLR[4]);
  ql[7] = 0.5 * (qLL[7] + qRL[7]); qr[7] = 0.5 * (qRR[7] + qLR[7]);
  if (roe) athena_roe(ql, qr, 0.0, gamma, P.smallc, fy); else if (up) upwind(ql, qr, 0.0, gamma, fy); else lax_friedrich(ql, qr, 0.0, gamma, fy);
  return E + (fx[5] - fy[5]);
}

}  // namespace mhd
}  // namespace ramses_amd
