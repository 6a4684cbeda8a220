// mhd_host_check.cpp -- TEST INFRASTRUCTURE ONLY (tests/test_mhd_core_host.py).
//
// The product's MHD headers (ramses_amd/csrc/mhd_core.hpp, mhd_assemble.hpp) compiled for the HOST and driven over the
// reference's own 6^3 stencils, with mag_unsplit's array layout on both sides, so that a CPU test can compare them
// with the compiled reference (oracle/_ref/libref_kernels3d_mhd.so: ref_mag_unsplit) bit for bit without a GPU.
// Built by the test with: g++ -O2 -std=c++17 -ffp-contract=off -fno-fast-math -shared -fPIC
#include <cstddef>
#include <vector>

#include "../../ramses_amd/csrc/mhd_assemble.hpp"

using namespace ramses_amd::mhd;

namespace {
// Fortran uin(nvector, -1:4, -1:4, -1:4, nvar+3)
struct Stencil {
  const double *uin;
  int nv, l;
  double u(int var, int i, int j, int k) const { return uin[l + (size_t)nv * ((i + 1) + 6 * ((j + 1) + 6 * ((k + 1) + 6 * (size_t)var)))]; }
};
struct Acc {
  // q: [8][6^3]; bf: [3][7^3]; E: [3][6^3] (indices -1..4 / -1..5)
  std::vector<double> Q, B, Ef;
  Acc() : Q(8 * 216), B(3 * 343), Ef(3 * 216) {}
  double &qw(int n, int i, int j, int k) { return Q[n * 216 + (i + 1) + 6 * ((j + 1) + 6 * (k + 1))]; }
  double &bw(int c, int i, int j, int k) { return B[c * 343 + (i + 1) + 7 * ((j + 1) + 7 * (k + 1))]; }
  double &ew(int c, int i, int j, int k) { return Ef[c * 216 + (i + 1) + 6 * ((j + 1) + 6 * (k + 1))]; }
  double q(int n, int i, int j, int k) const { return Q[n * 216 + (i + 1) + 6 * ((j + 1) + 6 * (k + 1))]; }
  double bf(int c, int i, int j, int k) const { return B[c * 343 + (i + 1) + 7 * ((j + 1) + 7 * (k + 1))]; }
  double E(int c, int i, int j, int k) const { return Ef[c * 216 + (i + 1) + 6 * ((j + 1) + 6 * (k + 1))]; }
};
}  // namespace

// gravin: null, or the acceleration gravin(nvector,-1:4,-1:4,-1:4,3) of the reference (ctoprim's half-step kick :2160-2170)
static const double *g_gravin = nullptr;
extern "C" void mhd_host_set_gravin(const double *gravin) { g_gravin = gravin; }

extern "C" int mhd_host_unsplit(const double *uin, int ngrid, int nvector, double dx, double dt, double gamma, double smallr, double smallc,
                                int slope_type, int slope_mag_type, double slope_theta, int iriemann, int iriemann2d, double *flux,
                                double *emfx, double *emfy, double *emfz) {
  MhdConst P;
  P.gamma = gamma; P.smallr = smallr; P.smallc = smallc; P.slope_theta = slope_theta;
  P.slope_type = slope_type; P.slope_mag_type = slope_mag_type; P.riemann = iriemann; P.riemann2d = iriemann2d;
  if (!slope_type_supported(slope_type) || !slope_mag_type_supported(slope_mag_type) || !riemann_supported(iriemann) || !riemann2d_supported(iriemann2d))
    return -2;
  const double dtdx = dt / dx;
  const int nv = nvector;
  // flux(nvector,1:3,1:3,1:3,1:8,1:3), emf*(nvector,1:3,1:3,1:3)
  auto F = [&](int l, int i, int j, int k, int var, int d) -> double & {
    return flux[l + (size_t)nv * ((i - 1) + 3 * ((j - 1) + 3 * ((k - 1) + 3 * (var + 8 * (size_t)d))))];
  };
  auto EM = [&](double *e, int l, int i, int j, int k) -> double & { return e[l + (size_t)nv * ((i - 1) + 3 * ((j - 1) + 3 * (k - 1)))]; };
  std::vector<TraceOut> T(216);
  auto tr = [&](int i, int j, int k) -> TraceOut & { return T[(i + 1) + 6 * ((j + 1) + 6 * (k + 1))]; };
  for (int l = 0; l < ngrid; l++) {
    Stencil S{uin, nv, l};
    Acc a;
    // ctoprim
    for (int k = -1; k <= 4; k++)
      for (int j = -1; j <= 4; j++)
        for (int i = -1; i <= 4; i++) {
          const double u[5] = {S.u(0, i, j, k), S.u(1, i, j, k), S.u(2, i, j, k), S.u(3, i, j, k), S.u(4, i, j, k)};
          const double bl[3] = {S.u(5, i, j, k), S.u(6, i, j, k), S.u(7, i, j, k)};
          const double br[3] = {S.u(8, i, j, k), S.u(9, i, j, k), S.u(10, i, j, k)};
          double q[8];
          double gv[3];
          if (g_gravin)
            for (int d = 0; d < 3; d++) gv[d] = g_gravin[l + (size_t)nv * ((i + 1) + 6 * ((j + 1) + 6 * ((k + 1) + 6 * (size_t)d)))];
          ctoprim_cell(u, bl, br, g_gravin ? gv : nullptr, dt, P, q);
          for (int n = 0; n < 8; n++) a.qw(n, i, j, k) = q[n];
        }
    for (int k = -1; k <= 5; k++)
      for (int j = -1; j <= 5; j++)
        for (int i = -1; i <= 5; i++) {
          if (j <= 4 && k <= 4) a.bw(0, i, j, k) = i <= 4 ? S.u(5, i, j, k) : S.u(8, i - 1, j, k);
          if (i <= 4 && k <= 4) a.bw(1, i, j, k) = j <= 4 ? S.u(6, i, j, k) : S.u(9, i, j - 1, k);
          if (i <= 4 && j <= 4) a.bw(2, i, j, k) = k <= 4 ? S.u(7, i, j, k) : S.u(10, i, j, k - 1);
        }
    // trace3d: edge fields on 0..4, traced states on 0..3
    for (int k = 0; k <= 4; k++)
      for (int j = 0; j <= 4; j++)
        for (int i = 0; i <= 4; i++)
          for (int c = 0; c < 3; c++) a.ew(c, i, j, k) = efield(a, c, i, j, k);
    for (int k = 0; k <= 3; k++)
      for (int j = 0; j <= 3; j++)
        for (int i = 0; i <= 3; i++) {
          TraceIn I;
          trace_inputs(a, i, j, k, P, I);
          trace3d_cell(I, dtdx, dtdx, dtdx, P, tr(i, j, k));
        }
    // fluxes (mag_unsplit :95-157)
    for (int d = 0; d < 3; d++) {
      const int hi[3] = {d == 0 ? 3 : 2, d == 1 ? 3 : 2, d == 2 ? 3 : 2};
      for (int k = 1; k <= hi[2]; k++)
        for (int j = 1; j <= hi[1]; j++)
          for (int i = 1; i <= hi[0]; i++) {
            const TraceOut &lo = tr(i - (d == 0), j - (d == 1), k - (d == 2));
            double f[8];
            cmpflxm_face(lo.get(T_QM, d), tr(i, j, k).get(T_QP, d), d, P, f);
            for (int n = 0; n < 8; n++) F(l, i, j, k, n, d) = f[n] * dt / dx;
          }
    }
    // EMFs (:160-236)
    for (int e = 0; e < 3; e++) {
      const EdgeSource src = edge_sources(e);
      const int hi[3] = {e == 0 ? 2 : 3, e == 1 ? 2 : 3, e == 2 ? 2 : 3};
      double *out = e == 0 ? emfx : (e == 1 ? emfy : emfz);
      for (int k = 1; k <= hi[2]; k++)
        for (int j = 1; j <= hi[1]; j++)
          for (int i = 1; i <= hi[0]; i++) {
            const double *s[4];
            for (int m = 0; m < 4; m++) {
              const TraceOut &t = tr(i + src.off[m][0], j + src.off[m][1], k + src.off[m][2]);
              s[m] = t.get(T_QRT + src.kind[m], e);
            }
            double a0[8], a1[8], a2[8], a3[8];
            for (int n = 0; n < 8; n++) { a0[n] = s[0][n]; a1[n] = s[1][n]; a2[n] = s[2][n]; a3[n] = s[3][n]; }
            EM(out, l, i, j, k) = cmp_mag_flx_edge(a0, a1, a2, a3, e, P) * dt / dx;
          }
    }
  }
  return 0;
}

// cmpdt of ncell cells: uu(nvector, 11) in Fortran order (cell index fastest); returns min(courant_factor*dx/smallc, dtcell...)
extern "C" double mhd_host_cmpdt(const double *uu, int ncell, int nvector, double dx, double courant_factor, double gamma, double smallr,
                                 double smallc) {
  MhdConst P;
  P.gamma = gamma; P.smallr = smallr; P.smallc = smallc; P.slope_theta = 1.5;
  P.slope_type = 1; P.slope_mag_type = 1; P.riemann = 0; P.riemann2d = 0;
  double dt = courant_factor * dx / smallc;
  for (int l = 0; l < ncell; l++) {
    double u[11];
    for (int n = 0; n < 11; n++) u[n] = uu[l + (size_t)nvector * n];
    const double d = cmpdt_cell(u, dx, courant_factor, P);
    dt = d < dt ? d : dt;
  }
  return dt;
}
