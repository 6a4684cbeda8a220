/* hydro_oracle_plmde.c -- TEST INFRASTRUCTURE ONLY (see hydro_oracle.h).
 *
 * PLMDE characteristic tracing: tracex / tracexy / tracexyz,
 * hydro/uplmde.f90:5-138, 144-368, 375-696, restated as one routine over the
 * active dimensions.  Quirks of the reference that are kept on purpose:
 *   - 3-D: every transverse term is scaled by half*dtdx (uplmde.f90:453-469),
 *     2-D: x-states use half*dtdy and y-states half*dtdx (:213-221);
 *   - 3-D passive scalars trace the z states with dtdy (:679,685).
 */
#include "hydro_oracle.h"

#include <math.h>
#include <stddef.h>

static inline double dmax_(double a, double b) { return a > b ? a : b; }

void ora_trace_plmde(const ora_hydro_params *p, int nv, const double *q,
                     const double *dq, const double *c, double *qm, double *qp,
                     const double dxs[3], double dt, int ngrid) {
  const int ndim = p->ndim, nvar = p->nvar;
  const int NI = 6, NJ = ndim > 1 ? 6 : 1, NK = ndim > 2 ? 6 : 1;
  const int LI = -1, LJ = ndim > 1 ? -1 : 1, LK = ndim > 2 ? -1 : 1;
  const int ilo = 0, ihi = 3;
  const int jlo = ndim > 1 ? 0 : 1, jhi = ndim > 1 ? 3 : 1;
  const int klo = ndim > 2 ? 0 : 1, khi = ndim > 2 ? 3 : 1;
  const int ir = 0, ip = ndim + 1;
  const double project_out = 1.0;
  double dtd[3] = {0, 0, 0};
  for (int d = 0; d < ndim; d++) dtd[d] = dt / dxs[d];
  /* transverse direction lists, in the reference's order of subtraction */
  static const int T3[3][2] = {{1, 2}, {0, 2}, {1, 0}};
  static const int T2[2][1] = {{1}, {0}};

#define CI(l, i, j, k, n) ((size_t)(l) + (size_t)nv * ((size_t)((i) - LI) + (size_t)NI * ((size_t)((j) - LJ) + (size_t)NJ * ((size_t)((k) - LK) + (size_t)NK * (size_t)(n)))))
#define CID(l, i, j, k, n, d) CI(l, i, j, k, (n) + nvar * (d))

  for (int k = klo; k <= khi; k++)
    for (int j = jlo; j <= jhi; j++)
      for (int i = ilo; i <= ihi; i++)
        for (int l = 0; l < ngrid; l++) {
          const double cc = c[CI(l, i, j, k, 0)];
          const double r = q[CI(l, i, j, k, ir)];
          const double pr = q[CI(l, i, j, k, ip)];
          double vel[3] = {0, 0, 0}, dr[3], dp[3], dv[3][3];
          for (int d = 0; d < ndim; d++) vel[d] = q[CI(l, i, j, k, 1 + d)];
          const double csq = p->gamma * pr / r;
          for (int d = 0; d < ndim; d++) {
            dr[d] = dq[CID(l, i, j, k, ir, d)];
            dp[d] = dq[CID(l, i, j, k, ip, d)];
            for (int cmp = 0; cmp < ndim; cmp++) dv[cmp][d] = dq[CID(l, i, j, k, 1 + cmp, d)];
          }
          for (int d = 0; d < ndim; d++) {
            /* transverse derivative terms */
            double sr = 0, sp = 0, sv[3] = {0, 0, 0};
            int nt = ndim - 1;
            if (nt > 0) {
              const int *T = ndim == 3 ? T3[d] : T2[d];
              double fac = ndim == 3 ? 0.5 * dtd[0] : 0.5 * dtd[T[0]];
              int t0 = T[0];
              double ar = -vel[t0] * dr[t0];
              double ap = -vel[t0] * dp[t0];
              double divt = dv[t0][t0];
              double av[3];
              for (int cmp = 0; cmp < ndim; cmp++) av[cmp] = -vel[t0] * dv[cmp][t0];
              if (nt > 1) {
                int t1 = T[1];
                ar = ar - vel[t1] * dr[t1];
                ap = ap - vel[t1] * dp[t1];
                divt = divt + dv[t1][t1];
                for (int cmp = 0; cmp < ndim; cmp++) av[cmp] = av[cmp] - vel[t1] * dv[cmp][t1];
              }
              sr = fac * (ar - (divt)*r);
              sp = fac * (ap - (divt)*p->gamma * pr);
              for (int cmp = 0; cmp < ndim; cmp++) {
                if (cmp == d) sv[cmp] = fac * (av[cmp]);
                else sv[cmp] = fac * (av[cmp] - (dp[cmp]) / r);
              }
            }
            /* characteristic analysis along d */
            const double vn = vel[d];
            const double dvn = dv[d][d];
            const double alpham = 0.5 * (dp[d] / csq - dvn * r / cc);
            const double alphap = 0.5 * (dp[d] / csq + dvn * r / cc);
            const double alpha0r = dr[d] - dp[d] / csq;
            double ccc = cc;
            if (fabs(dvn) > 3.0 * cc) ccc = 0.0;
            for (int side = 0; side < 2; side++) {
              /* side 0: "right" state qp (-one), side 1: "left" state qm (+one) */
              double spminus = (vn - ccc) * dtd[d];
              double spplus = (vn + ccc) * dtd[d];
              double spzero = (vn)*dtd[d];
              double sg;
              if (side == 0) {
                if ((vn + ccc) > 0.0) spplus = -project_out;
                if ((vn - ccc) > 0.0) spminus = -project_out;
                if (vn > 0.0) spzero = -project_out;
                sg = -1.0;
              } else {
                if ((vn + ccc) <= 0.0) spplus = +project_out;
                if ((vn - ccc) <= 0.0) spminus = +project_out;
                if (vn <= 0.0) spzero = +project_out;
                sg = +1.0;
              }
              double ap_ = 0.5 * (sg - spplus) * alphap;
              double am_ = 0.5 * (sg - spminus) * alpham;
              double azr = 0.5 * (sg - spzero) * alpha0r;
              double *out = side == 0 ? qp : qm;
              double vr = r + (ap_ + am_ + azr);
              double vu = vn + (ap_ - am_) * cc / r;
              double vp = pr + (ap_ + am_) * csq;
              if (nt > 0) { vr = vr + sr; vu = vu + sv[d]; vp = vp + sp; }
              out[CID(l, i, j, k, ir, d)] = dmax_(p->smallr, vr);
              out[CID(l, i, j, k, 1 + d, d)] = vu;
              out[CID(l, i, j, k, ip, d)] = vp;
              for (int cmp = 0; cmp < ndim; cmp++) {
                if (cmp == d) continue;
                double azt = 0.5 * (sg - spzero) * dv[cmp][d];
                out[CID(l, i, j, k, 1 + cmp, d)] = vel[cmp] + (azt) + sv[cmp];
              }
            }
          }
        }

  /* passive scalars: uplmde.f90:112-136, 327-366, 640-694 */
  for (int n = ndim + 2; n < nvar; n++)
    for (int k = klo; k <= khi; k++)
      for (int j = jlo; j <= jhi; j++)
        for (int i = ilo; i <= ihi; i++)
          for (int l = 0; l < ngrid; l++) {
            double a = q[CI(l, i, j, k, n)];
            double vel[3] = {0, 0, 0}, da[3];
            for (int d = 0; d < ndim; d++) { vel[d] = q[CI(l, i, j, k, 1 + d)]; da[d] = dq[CID(l, i, j, k, n, d)]; }
            for (int d = 0; d < ndim; d++) {
              double sa = 0.0;
              int nt = ndim - 1;
              if (nt > 0) {
                const int *T = ndim == 3 ? T3[d] : T2[d];
                double fac = ndim == 3 ? 0.5 * dtd[0] : 0.5 * dtd[T[0]];
                double acc = -vel[T[0]] * da[T[0]];
                if (nt > 1) acc = acc - vel[T[1]] * da[T[1]];
                sa = fac * (acc);
              }
              double dtn = (ndim == 3 && d == 2) ? dtd[1] : dtd[d];
              double spzero = (vel[d]) * dtn;
              if (vel[d] > 0.0) spzero = -project_out;
              double azr = 0.5 * (-1.0 - spzero) * da[d];
              qp[CID(l, i, j, k, n, d)] = nt > 0 ? a + azr + sa : a + azr;
              spzero = (vel[d]) * dtn;
              if (vel[d] <= 0.0) spzero = +project_out;
              double azl = 0.5 * (+1.0 - spzero) * da[d];
              qm[CID(l, i, j, k, n, d)] = nt > 0 ? a + azl + sa : a + azl;
            }
          }
#undef CI
#undef CID
}
