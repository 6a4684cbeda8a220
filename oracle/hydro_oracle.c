/*
 * hydro_oracle.c -- TEST INFRASTRUCTURE ONLY (see hydro_oracle.h).
 *
 * Plain-C restatement of the reference's Godunov hydro path, written to follow
 * the reference's floating-point operation order exactly (left-to-right
 * evaluation, no FMA contraction: build with -ffp-contract=off) so that it is
 * bit-identical to the reference's own unsplit() compiled for x86-64.
 *
 * Index conventions.  The reference dimensions its local patch arrays as
 *   (1:nvector, iu1:iu2, ju1:ju2, ku1:ku2, 1:nvar [,1:ndim])
 * with iu1=-1, iu2=4 for every ACTIVE dimension and 1:1 for inactive ones
 * (hydro/hydro_parameters.f90:19-30).  Here NI,NJ,NK are the extents (6 or 1)
 * and LO_I.. the lower bounds (-1 or 1).
 */
#include "hydro_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define MAXVAR 16

typedef struct {
  int nv;                 /* nvector                                   */
  int ndim, nvar;
  int NI, NJ, NK;         /* cell extents  (6 or 1)                    */
  int LI, LJ, LK;         /* cell lower bounds (-1 or 1)               */
  int FI, FJ, FK;         /* face extents  (3 or 1), lower bound 1     */
  int iu1, iu2, ju1, ju2, ku1, ku2;
  int if1, if2, jf1, jf2, kf1, kf2;
} patch_t;

static patch_t make_patch(int ndim, int nvar, int nvector) {
  patch_t g;
  g.nv = nvector; g.ndim = ndim; g.nvar = nvar;
  g.iu1 = -1; g.iu2 = 4;
  g.ju1 = ndim > 1 ? -1 : 1; g.ju2 = ndim > 1 ? 4 : 1;
  g.ku1 = ndim > 2 ? -1 : 1; g.ku2 = ndim > 2 ? 4 : 1;
  g.if1 = 1; g.if2 = 3;
  g.jf1 = 1; g.jf2 = ndim > 1 ? 3 : 1;
  g.kf1 = 1; g.kf2 = ndim > 2 ? 3 : 1;
  g.NI = g.iu2 - g.iu1 + 1; g.NJ = g.ju2 - g.ju1 + 1; g.NK = g.ku2 - g.ku1 + 1;
  g.LI = g.iu1; g.LJ = g.ju1; g.LK = g.ku1;
  g.FI = 3; g.FJ = g.jf2; g.FK = g.kf2;
  return g;
}

/* cell-array index: (l,i,j,k,n)  */
static inline size_t CI(const patch_t *g, int l, int i, int j, int k, int n) {
  return (size_t)l + (size_t)g->nv * ((size_t)(i - g->LI) + (size_t)g->NI * ((size_t)(j - g->LJ) + (size_t)g->NJ * ((size_t)(k - g->LK) + (size_t)g->NK * (size_t)n)));
}
/* cell-array index with a trailing dimension: (l,i,j,k,n,d), n in [0,nvar) */
static inline size_t CID(const patch_t *g, int l, int i, int j, int k, int n, int d) {
  return CI(g, l, i, j, k, n + g->nvar * d);
}
/* flux array (l, i,j,k in 1:3, n, d) with nn entries per face (nvar or 2) */
static inline size_t FIX(const patch_t *g, int l, int i, int j, int k, int n, int d, int nn) {
  return (size_t)l + (size_t)g->nv * ((size_t)(i - 1) + (size_t)g->FI * ((size_t)(j - 1) + (size_t)g->FJ * ((size_t)(k - 1) + (size_t)g->FK * ((size_t)n + (size_t)nn * (size_t)d))));
}

static inline double dmax(double a, double b) { return a > b ? a : b; }
static inline double dmin(double a, double b) { return a < b ? a : b; }
static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }
static inline double fsign(double a, double b) { return copysign(fabs(a), b); }

/* ------------------------------------------------------------------------
 * ctoprim -- hydro/umuscl.f90:861-965
 * ---------------------------------------------------------------------- */
static void ora_ctoprim(const ora_hydro_params *p, const patch_t *g,
                        const double *uin, double *q, double *c,
                        const double *gravin, double dt, int ngrid) {
  const int ndim = g->ndim;
  const double smalle = p->smallc * p->smallc / p->gamma / (p->gamma - 1.0);
  const double dtxhalf = dt * 0.5;
  for (int k = g->ku1; k <= g->ku2; k++)
    for (int j = g->ju1; j <= g->ju2; j++)
      for (int i = g->iu1; i <= g->iu2; i++)
        for (int l = 0; l < ngrid; l++) {
          double rho = dmax(uin[CI(g, l, i, j, k, 0)], p->smallr);
          q[CI(g, l, i, j, k, 0)] = rho;
          double oneoverrho = 1.0 / rho;
          double eken = 0.0;
          for (int d = 0; d < ndim; d++) {
            double v = uin[CI(g, l, i, j, k, 1 + d)] * oneoverrho;
            q[CI(g, l, i, j, k, 1 + d)] = v;
            if (d == 0) eken = 0.5 * v * v;
            else eken = eken + 0.5 * v * v;
          }
          double erad = 0.0;
          double eint = dmax(uin[CI(g, l, i, j, k, ndim + 1)] * oneoverrho - eken - erad, smalle);
          double pr = (p->gamma - 1.0) * rho * eint;
          q[CI(g, l, i, j, k, ndim + 1)] = pr;
          double cc = p->gamma * pr;
          c[CI(g, l, i, j, k, 0)] = sqrt(cc * oneoverrho);
          for (int d = 0; d < ndim; d++)
            q[CI(g, l, i, j, k, 1 + d)] = q[CI(g, l, i, j, k, 1 + d)] + gravin[CI(g, l, i, j, k, d)] * dtxhalf;
        }
  /* passive scalars: umuscl.f90:949-963 */
  for (int n = ndim + 2; n < g->nvar; n++)
    for (int k = g->ku1; k <= g->ku2; k++)
      for (int j = g->ju1; j <= g->ju2; j++)
        for (int i = g->iu1; i <= g->iu2; i++)
          for (int l = 0; l < ngrid; l++) {
            double oneoverrho = 1.0 / q[CI(g, l, i, j, k, 0)];
            q[CI(g, l, i, j, k, n)] = uin[CI(g, l, i, j, k, n)] * oneoverrho;
          }
}

/* ------------------------------------------------------------------------
 * uslope -- hydro/umuscl.f90:970-1480
 * ---------------------------------------------------------------------- */
/* the "dsgn*min(dlim,|dcen|)" limiter family with integer multiplier mm
 * (1-D: umuscl.f90:1009-1019, 2-D: :1112-1133, 3-D moncen: :1292-1325) */
static inline double slope_mm(double qm1, double q0, double qp1, double mm) {
  double dlft = mm * (q0 - qm1);
  double drgt = mm * (qp1 - q0);
  double dcen = 0.5 * (dlft + drgt) / mm;
  double dsgn = fsign(1.0, dcen);
  double slop = dmin(fabs(dlft), fabs(drgt));
  double dlim = slop;
  if ((dlft * drgt) <= 0.0) dlim = 0.0;
  return dsgn * dmin(dlim, fabs(dcen));
}
/* 3-D minmod, umuscl.f90:1246-1279 */
static inline double slope_minmod3(double qm1, double q0, double qp1) {
  double dlft = q0 - qm1;
  double drgt = qp1 - q0;
  if ((dlft * drgt) <= 0.0) return 0.0;
  else if (dlft > 0) return dmin(dlft, drgt);
  else return dmax(dlft, drgt);
}
/* van Leer, umuscl.f90:1085-1094,1387-1418 */
static inline double slope_vanleer(double qm1, double q0, double qp1) {
  double dlft = q0 - qm1;
  double drgt = qp1 - q0;
  if ((dlft * drgt) <= 0.0) return 0.0;
  return (2 * dlft * drgt / (dlft + drgt));
}
/* generalised moncen/minmod, umuscl.f90:1095-1106,1423-1460 */
static inline double slope_theta(double qm1, double q0, double qp1, double theta) {
  double dlft = q0 - qm1;
  double drgt = qp1 - q0;
  double dcen = 0.5 * (dlft + drgt);
  double dsgn = fsign(1.0, dcen);
  double slop = dmin(theta * fabs(dlft), theta * fabs(drgt));
  double dlim = slop;
  if ((dlft * drgt) <= 0.0) dlim = 0.0;
  return dsgn * dmin(dlim, fabs(dcen));
}

static void ora_uslope(const ora_hydro_params *p, const patch_t *g,
                       const double *q, double *dq, double dx, double dt,
                       int ngrid) {
  const int ndim = g->ndim, st = p->slope_type;
  const int ilo = imin(1, g->iu1 + 1), ihi = imax(1, g->iu2 - 1);
  const int jlo = imin(1, g->ju1 + 1), jhi = imax(1, g->ju2 - 1);
  const int klo = imin(1, g->ku1 + 1), khi = imax(1, g->ku2 - 1);
  size_t ntot = (size_t)g->nv * g->NI * g->NJ * g->NK * g->nvar * ndim;
  if (st == 0) { memset(dq, 0, ntot * sizeof(double)); return; }

  for (int n = 0; n < g->nvar; n++)
    for (int k = klo; k <= khi; k++)
      for (int j = jlo; j <= jhi; j++)
        for (int i = ilo; i <= ihi; i++)
          for (int l = 0; l < ngrid; l++) {
            const double q0 = q[CI(g, l, i, j, k, n)];
            /* ---- positivity preserving unsplit slope (2-D :1135-1171, 3-D :1326-1386) */
            if (st == 3 && ndim >= 2) {
              double vmin = 0, vmax = 0; int first = 1;
              int k0 = ndim > 2 ? -1 : 0, k1 = ndim > 2 ? 1 : 0;
              /* order of the min/max arguments is irrelevant to the result */
              for (int dk = k0; dk <= k1; dk++)
                for (int dj = -1; dj <= 1; dj++)
                  for (int di = -1; di <= 1; di++) {
                    double df = q[CI(g, l, i + di, j + dj, k + dk, n)] - q0;
                    if (first) { vmin = vmax = df; first = 0; }
                    else { vmin = dmin(vmin, df); vmax = dmax(vmax, df); }
                  }
              double dfx = 0.5 * (q[CI(g, l, i + 1, j, k, n)] - q[CI(g, l, i - 1, j, k, n)]);
              double dfy = 0.5 * (q[CI(g, l, i, j + 1, k, n)] - q[CI(g, l, i, j - 1, k, n)]);
              double dfz = 0.0, dff;
              if (ndim > 2) {
                dfz = 0.5 * (q[CI(g, l, i, j, k + 1, n)] - q[CI(g, l, i, j, k - 1, n)]);
                dff = 0.5 * (fabs(dfx) + fabs(dfy) + fabs(dfz));
              } else {
                dff = 0.5 * (fabs(dfx) + fabs(dfy));
              }
              double slop;
              if (dff > 0.0) slop = dmin(1.0, dmin(fabs(vmin), fabs(vmax)) / dff);
              else slop = 1.0;
              dq[CID(g, l, i, j, k, n, 0)] = slop * dfx;
              dq[CID(g, l, i, j, k, n, 1)] = slop * dfy;
              if (ndim > 2) dq[CID(g, l, i, j, k, n, 2)] = slop * dfz;
              continue;
            }
            for (int d = 0; d < ndim; d++) {
              int di = d == 0, dj = d == 1, dk = d == 2;
              const double qm1 = q[CI(g, l, i - di, j - dj, k - dk, n)];
              const double qp1 = q[CI(g, l, i + di, j + dj, k + dk, n)];
              double s;
              if (ndim == 1 && (st == 1 || st == 2 || st == 3)) {
                s = slope_mm(qm1, q0, qp1, (double)imin(st, 2));
              } else if (ndim == 2 && (st == 1 || st == 2)) {
                s = slope_mm(qm1, q0, qp1, (double)st);
              } else if (ndim == 3 && st == 1) {
                s = slope_minmod3(qm1, q0, qp1);
              } else if (ndim == 3 && st == 2) {
                s = slope_mm(qm1, q0, qp1, 2.0);
              } else if (st == 7) {
                s = slope_vanleer(qm1, q0, qp1);
              } else if (st == 8) {
                s = slope_theta(qm1, q0, qp1, p->slope_theta);
              } else if (ndim == 1 && st == 4) { /* superbee, umuscl.f90:1020-1031 */
                double dcen = q[CI(g, l, i, j, k, 1)] * dt / dx;
                double dlft = 2.0 / (1.0 + dcen) * (q0 - qm1);
                double drgt = 2.0 / (1.0 - dcen) * (qp1 - q0);
                double dsgn = fsign(1.0, dlft);
                double slop = dmin(fabs(dlft), fabs(drgt));
                double dlim = slop;
                if ((dlft * drgt) <= 0.0) dlim = 0.0;
                s = dsgn * dlim;
              } else if (ndim == 1 && st == 5) { /* ultrabee, :1032-1056 */
                if (n == 0) {
                  double dcen = q[CI(g, l, i, j, k, 1)] * dt / dx;
                  double dlft, drgt;
                  if (dcen >= 0) {
                    dlft = 2.0 / (0.0 + dcen + 1e-10) * (q0 - qm1);
                    drgt = 2.0 / (1.0 - dcen) * (qp1 - q0);
                  } else {
                    dlft = 2.0 / (1.0 + dcen) * (q0 - qm1);
                    drgt = 2.0 / (0.0 - dcen + 1e-10) * (qp1 - q0);
                  }
                  double dsgn = fsign(1.0, dlft);
                  double slop = dmin(fabs(dlft), fabs(drgt));
                  double dlim = slop;
                  if ((dlft * drgt) <= 0.0) dlim = 0.0;
                  s = dsgn * dlim;
                } else s = 0.0;
              } else if (ndim == 1 && st == 6) { /* unstable, :1057-1071 */
                if (n == 0) {
                  double dlft = (q0 - qm1), drgt = (qp1 - q0);
                  s = 0.5 * (dlft + drgt);
                } else s = 0.0;
              } else {
                fprintf(stderr, "ora_uslope: unknown slope type %d for ndim=%d\n", st, ndim);
                abort();
              }
              dq[CID(g, l, i, j, k, n, d)] = s;
            }
          }
}

/* ------------------------------------------------------------------------
 * trace1d/2d/3d -- hydro/umuscl.f90:176-299, 305-476, 483-708
 * ---------------------------------------------------------------------- */
static void ora_trace(const ora_hydro_params *p, const patch_t *g,
                      const double *q, const double *dq, double *qm, double *qp,
                      const double dxs[3], double dt, int ngrid) {
  const int ndim = g->ndim;
  const int ir = 0, ip = ndim + 1;
  double dtd[3];
  for (int d = 0; d < ndim; d++) dtd[d] = dt / dxs[d];
  const int ilo = imin(1, g->iu1 + 1), ihi = imax(1, g->iu2 - 1);
  const int jlo = imin(1, g->ju1 + 1), jhi = imax(1, g->ju2 - 1);
  const int klo = imin(1, g->ku1 + 1), khi = imax(1, g->ku2 - 1);
  for (int k = klo; k <= khi; k++)
    for (int j = jlo; j <= jhi; j++)
      for (int i = ilo; i <= ihi; i++)
        for (int l = 0; l < ngrid; l++) {
          double r = q[CI(g, l, i, j, k, ir)];
          double pr = q[CI(g, l, i, j, k, ip)];
          double vel[3] = {0, 0, 0};
          double dr[3], dp[3], dv[3][3]; /* dv[c][d] = slope of velocity c along d */
          for (int d = 0; d < ndim; d++) vel[d] = q[CI(g, l, i, j, k, 1 + d)];
          for (int d = 0; d < ndim; d++) {
            dr[d] = dq[CID(g, l, i, j, k, ir, d)];
            dp[d] = dq[CID(g, l, i, j, k, ip, d)];
            for (int c = 0; c < ndim; c++) dv[c][d] = dq[CID(g, l, i, j, k, 1 + c, d)];
          }
          /* divergence (dux+dvy+dwz), left to right */
          double div = dv[0][0];
          for (int d = 1; d < ndim; d++) div = div + dv[d][d];
          /* sr0 = -u*drx-v*dry-w*drz - div*r */
          double acc = -vel[0] * dr[0];
          for (int d = 1; d < ndim; d++) acc = acc - vel[d] * dr[d];
          double sr0 = acc - div * r;
          acc = -vel[0] * dp[0];
          for (int d = 1; d < ndim; d++) acc = acc - vel[d] * dp[d];
          double sp0 = acc - div * p->gamma * pr;
          double sv0[3];
          for (int c = 0; c < ndim; c++) {
            acc = -vel[0] * dv[c][0];
            for (int d = 1; d < ndim; d++) acc = acc - vel[d] * dv[c][d];
            sv0[c] = acc - (dp[c]) / r;
          }
          for (int d = 0; d < ndim; d++) {
            double v;
            /* right state at left interface */
            v = r - 0.5 * dr[d] + sr0 * dtd[d] * 0.5;
            if (v < p->smallr) v = r;
            qp[CID(g, l, i, j, k, ir, d)] = v;
            qp[CID(g, l, i, j, k, ip, d)] = pr - 0.5 * dp[d] + sp0 * dtd[d] * 0.5;
            for (int c = 0; c < ndim; c++)
              qp[CID(g, l, i, j, k, 1 + c, d)] = vel[c] - 0.5 * dv[c][d] + sv0[c] * dtd[d] * 0.5;
            /* left state at right interface */
            v = r + 0.5 * dr[d] + sr0 * dtd[d] * 0.5;
            if (v < p->smallr) v = r;
            qm[CID(g, l, i, j, k, ir, d)] = v;
            qm[CID(g, l, i, j, k, ip, d)] = pr + 0.5 * dp[d] + sp0 * dtd[d] * 0.5;
            for (int c = 0; c < ndim; c++)
              qm[CID(g, l, i, j, k, 1 + c, d)] = vel[c] + 0.5 * dv[c][d] + sv0[c] * dtd[d] * 0.5;
          }
        }
  /* passive scalars: umuscl.f90:276-296, 450-474, 681-706 */
  for (int n = ndim + 2; n < g->nvar; n++)
    for (int k = klo; k <= khi; k++)
      for (int j = jlo; j <= jhi; j++)
        for (int i = ilo; i <= ihi; i++)
          for (int l = 0; l < ngrid; l++) {
            double a = q[CI(g, l, i, j, k, n)];
            double da[3];
            for (int d = 0; d < ndim; d++) da[d] = dq[CID(g, l, i, j, k, n, d)];
            double sa0 = -q[CI(g, l, i, j, k, 1)] * da[0];
            for (int d = 1; d < ndim; d++) sa0 = sa0 - q[CI(g, l, i, j, k, 1 + d)] * da[d];
            for (int d = 0; d < ndim; d++) {
              qp[CID(g, l, i, j, k, n, d)] = a - 0.5 * da[d] + sa0 * dtd[d] * 0.5;
              qm[CID(g, l, i, j, k, n, d)] = a + 0.5 * da[d] + sa0 * dtd[d] * 0.5;
            }
          }
}

/* ------------------------------------------------------------------------
 * Riemann solvers -- hydro/godunov_utils.f90
 * States are (rho, u_n, P, u_t1, u_t2, scalars...) per cmpflxm's permutation.
 * ---------------------------------------------------------------------- */
#define QL(n) qleft[(size_t)i + (size_t)nvector * (n)]
#define QR(n) qright[(size_t)i + (size_t)nvector * (n)]
#define FG(n) fgdnv[(size_t)i + (size_t)nvector * (n)]

/* godunov_utils.f90:660-820 */
static void riemann_llf(const ora_hydro_params *p, const double *qleft,
                        const double *qright, double *fgdnv, int ngrid,
                        int nvector) {
  const int ndim = p->ndim, nvar = p->nvar;
  const double smallp = p->smallc * p->smallc / p->gamma;
  const double entho = 1.0 / (p->gamma - 1.0);
  for (int i = 0; i < ngrid; i++) {
    double rl = dmax(QL(0), p->smallr), ul = QL(1);
    double pl = dmax(QL(2), rl * smallp);
    double cl = p->gamma * pl;
    cl = sqrt(cl / rl);
    double rr = dmax(QR(0), p->smallr), ur = QR(1);
    double pr = dmax(QR(2), rr * smallp);
    double cr = p->gamma * pr;
    cr = sqrt(cr / rr);
    double cmax = dmax(fabs(ul) + cl, fabs(ur) + cr);
    double uleft[MAXVAR + 1], uright[MAXVAR + 1], fleft[MAXVAR + 1], fright[MAXVAR + 1];
    uleft[0] = QL(0); uright[0] = QR(0);
    uleft[1] = QL(0) * QL(1); uright[1] = QR(0) * QR(1);
    uleft[2] = QL(2) * entho + 0.5 * QL(0) * (QL(1) * QL(1));
    uright[2] = QR(2) * entho + 0.5 * QR(0) * (QR(1) * QR(1));
    if (ndim > 1) {
      uleft[2] = uleft[2] + 0.5 * QL(0) * (QL(3) * QL(3));
      uright[2] = uright[2] + 0.5 * QR(0) * (QR(3) * QR(3));
    }
    if (ndim > 2) {
      uleft[2] = uleft[2] + 0.5 * QL(0) * (QL(4) * QL(4));
      uright[2] = uright[2] + 0.5 * QR(0) * (QR(4) * QR(4));
    }
    for (int n = 3; n < nvar; n++) { uleft[n] = QL(0) * QL(n); uright[n] = QR(0) * QR(n); }
    uleft[nvar] = QL(2) * entho; uright[nvar] = QR(2) * entho;
    fleft[0] = QL(1) * uleft[0]; fright[0] = QR(1) * uright[0];
    fleft[1] = QL(1) * uleft[1] + QL(2); fright[1] = QR(1) * uright[1] + QR(2);
    fleft[2] = QL(1) * (uleft[2] + QL(2)); fright[2] = QR(1) * (uright[2] + QR(2));
    for (int n = 3; n <= nvar; n++) { fleft[n] = QL(1) * uleft[n]; fright[n] = QR(1) * uright[n]; }
    for (int n = 0; n <= nvar; n++)
      FG(n) = 0.5 * (fleft[n] + fright[n] - cmax * (uright[n] - uleft[n]));
  }
}

/* godunov_utils.f90:825-983 */
static void riemann_hll(const ora_hydro_params *p, const double *qleft,
                        const double *qright, double *fgdnv, int ngrid,
                        int nvector) {
  const int ndim = p->ndim, nvar = p->nvar;
  const double smallp = p->smallc * p->smallc / p->gamma;
  const double entho = 1.0 / (p->gamma - 1.0);
  for (int i = 0; i < ngrid; i++) {
    double rl = dmax(QL(0), p->smallr), ul = QL(1);
    double pl = dmax(QL(2), rl * smallp);
    double cl = p->gamma * pl;
    cl = sqrt(cl / rl);
    double rr = dmax(QR(0), p->smallr), ur = QR(1);
    double pr = dmax(QR(2), rr * smallp);
    double cr = p->gamma * pr;
    cr = sqrt(cr / rr);
    double SL = dmin(dmin(ul, ur) - dmax(cl, cr), 0.0);
    double SR = dmax(dmax(ul, ur) + dmax(cl, cr), 0.0);
    double uleft[MAXVAR + 1], uright[MAXVAR + 1], fleft[MAXVAR + 1], fright[MAXVAR + 1];
    uleft[0] = QL(0); uright[0] = QR(0);
    uleft[1] = QL(0) * QL(1); uright[1] = QR(0) * QR(1);
    uleft[2] = QL(2) * entho + 0.5 * QL(0) * (QL(1) * QL(1));
    uright[2] = QR(2) * entho + 0.5 * QR(0) * (QR(1) * QR(1));
    if (ndim > 1) {
      uleft[2] = uleft[2] + 0.5 * QL(0) * (QL(3) * QL(3));
      uright[2] = uright[2] + 0.5 * QR(0) * (QR(3) * QR(3));
    }
    if (ndim > 2) {
      uleft[2] = uleft[2] + 0.5 * QL(0) * (QL(4) * QL(4));
      uright[2] = uright[2] + 0.5 * QR(0) * (QR(4) * QR(4));
    }
    for (int n = 3; n < nvar; n++) { uleft[n] = QL(0) * QL(n); uright[n] = QR(0) * QR(n); }
    uleft[nvar] = QL(2) * entho; uright[nvar] = QR(2) * entho;
    fleft[0] = uleft[1]; fright[0] = uright[1];
    fleft[1] = QL(2) + uleft[1] * QL(1); fright[1] = QR(2) + uright[1] * QR(1);
    fleft[2] = QL(1) * (uleft[2] + QL(2)); fright[2] = QR(1) * (uright[2] + QR(2));
    for (int n = 3; n <= nvar; n++) { fleft[n] = QL(1) * uleft[n]; fright[n] = QR(1) * uright[n]; }
    for (int n = 0; n <= nvar; n++)
      FG(n) = (SR * fleft[n] - SL * fright[n] + SR * SL * (uright[n] - uleft[n])) / (SR - SL);
  }
}

/* godunov_utils.f90:988-1209 */
static void riemann_hllc(const ora_hydro_params *p, const double *qleft,
                         const double *qright, double *fgdnv, int ngrid,
                         int nvector) {
  const int ndim = p->ndim, nvar = p->nvar;
  const double smallp = p->smallc * p->smallc / p->gamma;
  const double entho = 1.0 / (p->gamma - 1.0);
  for (int i = 0; i < ngrid; i++) {
    double rl = dmax(QL(0), p->smallr);
    double Pl = dmax(QL(2), rl * smallp);
    double ul = QL(1);
    double el = Pl * entho;
    double ecinl = 0.5 * rl * ul * ul;
    if (ndim > 1) ecinl = ecinl + 0.5 * rl * (QL(3) * QL(3));
    if (ndim > 2) ecinl = ecinl + 0.5 * rl * (QL(4) * QL(4));
    double etotl = el + ecinl;
    double Ptotl = Pl;
    double rr = dmax(QR(0), p->smallr);
    double Pr = dmax(QR(2), rr * smallp);
    double ur = QR(1);
    double er = Pr * entho;
    double ecinr = 0.5 * rr * ur * ur;
    if (ndim > 1) ecinr = ecinr + 0.5 * rr * (QR(3) * QR(3));
    if (ndim > 2) ecinr = ecinr + 0.5 * rr * (QR(4) * QR(4));
    double etotr = er + ecinr;
    double Ptotr = Pr;
    double cfastl = p->gamma * Pl;
    cfastl = sqrt(dmax(cfastl / rl, p->smallc * p->smallc));
    double cfastr = p->gamma * Pr;
    cfastr = sqrt(dmax(cfastr / rr, p->smallc * p->smallc));
    double SL = dmin(ul, ur) - dmax(cfastl, cfastr);
    double SR = dmax(ul, ur) + dmax(cfastl, cfastr);
    double rcl = rl * (ul - SL);
    double rcr = rr * (SR - ur);
    double ustar = (rcr * ur + rcl * ul + (Ptotl - Ptotr)) / (rcr + rcl);
    double Ptotstar = (rcr * Ptotl + rcl * Ptotr + rcl * rcr * (ul - ur)) / (rcr + rcl);
    double rstarl = rl * (SL - ul) / (SL - ustar);
    double etotstarl = ((SL - ul) * etotl - Ptotl * ul + Ptotstar * ustar) / (SL - ustar);
    double estarl = el * (SL - ul) / (SL - ustar);
    double rstarr = rr * (SR - ur) / (SR - ustar);
    double etotstarr = ((SR - ur) * etotr - Ptotr * ur + Ptotstar * ustar) / (SR - ustar);
    double estarr = er * (SR - ur) / (SR - ustar);
    double ro, uo, Ptoto, etoto, eo;
    if (SL > 0.0) { ro = rl; uo = ul; Ptoto = Ptotl; etoto = etotl; eo = el; }
    else if (ustar > 0.0) { ro = rstarl; uo = ustar; Ptoto = Ptotstar; etoto = etotstarl; eo = estarl; }
    else if (SR > 0.0) { ro = rstarr; uo = ustar; Ptoto = Ptotstar; etoto = etotstarr; eo = estarr; }
    else { ro = rr; uo = ur; Ptoto = Ptotr; etoto = etotr; eo = er; }
    FG(0) = ro * uo;
    FG(1) = ro * uo * uo + Ptoto;
    FG(2) = (etoto + Ptoto) * uo;
    for (int n = 3; n < nvar; n++) {
      if (ustar > 0) FG(n) = ro * uo * QL(n);
      else FG(n) = ro * uo * QR(n);
    }
    FG(nvar) = uo * eo;
  }
}

/* shared tail of riemann_approx / riemann_acoustic:
 * godunov_utils.f90:465-493 and :627-653 */
static inline void gdnv_to_flux(const ora_hydro_params *p, const double *qg,
                                double *f) {
  const int ndim = p->ndim, nvar = p->nvar;
  const double entho = 1.0 / (p->gamma - 1.0);
  f[0] = qg[0] * qg[1];
  f[1] = qg[2] + qg[0] * (qg[1] * qg[1]);
  double etot = qg[2] * entho + 0.5 * qg[0] * (qg[1] * qg[1]);
  if (ndim > 1) etot = etot + 0.5 * qg[0] * (qg[3] * qg[3]);
  if (ndim > 2) etot = etot + 0.5 * qg[0] * (qg[4] * qg[4]);
  f[2] = qg[1] * (etot + qg[2]);
  for (int n = 3; n <= nvar; n++) f[n] = f[0] * qg[n];
}

/* godunov_utils.f90:500-655 */
static void riemann_acoustic(const ora_hydro_params *p, const double *qleft,
                             const double *qright, double *fgdnv, int ngrid,
                             int nvector) {
  const int nvar = p->nvar;
  const double smallp = p->smallc * p->smallc / p->gamma;
  const double entho = 1.0 / (p->gamma - 1.0);
  for (int i = 0; i < ngrid; i++) {
    double rl = dmax(QL(0), p->smallr), ul = QL(1), pl = dmax(QL(2), rl * smallp);
    double rr = dmax(QR(0), p->smallr), ur = QR(1), pr = dmax(QR(2), rr * smallp);
    double cl = sqrt(p->gamma * pl / rl);
    double cr = sqrt(p->gamma * pr / rr);
    double wl = cl * rl, wr = cr * rr;
    double pstar = ((wr * pl + wl * pr) + wl * wr * (ul - ur)) / (wl + wr);
    double ustar = ((wr * ur + wl * ul) + (pl - pr)) / (wl + wr);
    double sgnm = fsign(1.0, ustar);
    double ro, uo, po, co;
    if (sgnm == 1.0) { ro = rl; uo = ul; po = pl; co = cl; }
    else { ro = rr; uo = ur; po = pr; co = cr; }
    double rstar = ro + (pstar - po) / (co * co);
    rstar = dmax(rstar, p->smallr);
    double cstar = sqrt(fabs(p->gamma * pstar / rstar));
    cstar = dmax(cstar, p->smallc);
    double spout = co - sgnm * uo;
    double spin = cstar - sgnm * ustar;
    double ushock = 0.5 * (spin + spout);
    ushock = dmax(ushock, -sgnm * ustar);
    if (pstar >= po) { spout = ushock; spin = spout; }
    double qg[MAXVAR + 1], f[MAXVAR + 1];
    if (spout < 0.0) { qg[0] = ro; qg[1] = uo; qg[2] = po; }
    else if (spin >= 0.0) { qg[0] = rstar; qg[1] = ustar; qg[2] = pstar; }
    else {
      double frac = spout / (spout - spin);
      qg[0] = frac * rstar + (1.0 - frac) * ro;
      qg[1] = frac * ustar + (1.0 - frac) * uo;
      qg[2] = frac * pstar + (1.0 - frac) * po;
    }
    for (int n = 3; n < nvar; n++) qg[n] = (sgnm == 1.0) ? QL(n) : QR(n);
    qg[nvar] = po / ro * entho;
    gdnv_to_flux(p, qg, f);
    for (int n = 0; n <= nvar; n++) FG(n) = f[n];
  }
}

/* godunov_utils.f90:268-495.  The reference iterates a compacted list of
 * not-yet-converged lanes for niter_riemann Newton steps; per lane this is:
 * iterate while the lane has not converged, at most niter_riemann times. */
static void riemann_approx(const ora_hydro_params *p, const double *qleft,
                           const double *qright, double *fgdnv, int ngrid,
                           int nvector) {
  const int nvar = p->nvar;
  const double gamma = p->gamma;
  const double smallp = p->smallc * p->smallc / gamma;
  const double smallpp = p->smallr * smallp;
  const double gamma6 = (gamma + 1.0) / (2.0 * gamma);
  const double entho = 1.0 / (gamma - 1.0);
  for (int i = 0; i < ngrid; i++) {
    double rl = dmax(QL(0), p->smallr), ul = QL(1), pl = dmax(QL(2), rl * smallp);
    double rr = dmax(QR(0), p->smallr), ur = QR(1), pr = dmax(QR(2), rr * smallp);
    double cl = gamma * pl * rl, cr = gamma * pr * rr;
    double wl = sqrt(cl), wr = sqrt(cr);
    double pstar = ((wr * pl + wl * pr) + wl * wr * (ul - ur)) / (wl + wr);
    pstar = dmax(pstar, 0.0);
    double pold = pstar;
    for (int iter = 0; iter < p->niter_riemann; iter++) {
      double wwl = sqrt(cl * (1.0 + gamma6 * (pold - pl) / pl));
      double wwr = sqrt(cr * (1.0 + gamma6 * (pold - pr) / pr));
      double ql = 2.0 * (wwl * wwl * wwl) / (wwl * wwl + cl);
      double qr = 2.0 * (wwr * wwr * wwr) / (wwr * wwr + cr);
      double usl = ul - (pold - pl) / wwl;
      double usr = ur + (pold - pr) / wwr;
      double delp = dmax(qr * ql / (qr + ql) * (usl - usr), -pold);
      pold = pold + delp;
      double uo = fabs(delp / (pold + smallpp));
      if (!(uo > 1e-06)) break;
    }
    pstar = pold;
    wl = sqrt(cl * (1.0 + gamma6 * (pstar - pl) / pl));
    wr = sqrt(cr * (1.0 + gamma6 * (pstar - pr) / pr));
    double ustar = 0.5 * (ul + (pl - pstar) / wl + ur - (pr - pstar) / wr);
    double sgnm = fsign(1.0, ustar);
    double ro, uo, po, wo;
    if (sgnm == 1.0) { ro = rl; uo = ul; po = pl; wo = wl; }
    else { ro = rr; uo = ur; po = pr; wo = wr; }
    double co = dmax(p->smallc, sqrt(fabs(gamma * po / ro)));
    double rstar;
    if (pstar >= po) rstar = ro / (1.0 + ro * (po - pstar) / (wo * wo));
    else rstar = ro * pow(pstar / po, 1.0 / gamma);
    rstar = dmax(rstar, p->smallr);
    double cstar = sqrt(fabs(gamma * pstar / rstar));
    cstar = dmax(cstar, p->smallc);
    double spout = co - sgnm * uo;
    double spin = cstar - sgnm * ustar;
    double ushock = wo / ro - sgnm * uo;
    if (pstar >= po) { spout = ushock; spin = spout; }
    double qg[MAXVAR + 1], f[MAXVAR + 1];
    if (spout <= 0.0) { qg[0] = ro; qg[1] = uo; qg[2] = po; }
    else if (spin >= 0.0) { qg[0] = rstar; qg[1] = ustar; qg[2] = pstar; }
    else {
      double frac = spout / (spout - spin);
      qg[1] = frac * ustar + (1.0 - frac) * uo;
      qg[2] = frac * pstar + (1.0 - frac) * po;
      qg[0] = ro * pow(qg[2] / po, 1.0 / gamma);
    }
    for (int n = 3; n < nvar; n++) qg[n] = (sgnm == 1.0) ? QL(n) : QR(n);
    qg[nvar] = po / ro * entho;
    gdnv_to_flux(p, qg, f);
    for (int n = 0; n <= nvar; n++) FG(n) = f[n];
  }
}

void ora_riemann(const ora_hydro_params *p, const double *qleft,
                 const double *qright, double *fgdnv, int ngrid, int nvector) {
  switch (p->riemann) {
    case ORA_RIEMANN_LLF: riemann_llf(p, qleft, qright, fgdnv, ngrid, nvector); break;
    case ORA_RIEMANN_HLLC: riemann_hllc(p, qleft, qright, fgdnv, ngrid, nvector); break;
    case ORA_RIEMANN_HLL: riemann_hll(p, qleft, qright, fgdnv, ngrid, nvector); break;
    case ORA_RIEMANN_ACOUSTIC: riemann_acoustic(p, qleft, qright, fgdnv, ngrid, nvector); break;
    case ORA_RIEMANN_EXACT: riemann_approx(p, qleft, qright, fgdnv, ngrid, nvector); break;
    default: fprintf(stderr, "unknown Riemann solver\n"); abort();
  }
}

/* ------------------------------------------------------------------------
 * cmpflxm -- hydro/umuscl.f90:714-856.  dir = 0,1,2; (sx,sy,sz) is the shift
 * that makes qm(i) refer to the cell on the LEFT of interface i (the
 * reference passes qm with lower bounds shifted by +1 along dir).
 * flx: (nv, NI,NJ,NK, nvar) ; tx: (nv,NI,NJ,NK,2), cell-array shaped.
 * ---------------------------------------------------------------------- */
static void ora_cmpflxm(const ora_hydro_params *p, const patch_t *g,
                        const double *qm, const double *qp, int dir, int ilo,
                        int ihi, int jlo, int jhi, int klo, int khi, double *flx,
                        double *tx, int ngrid, double *ql, double *qr, double *fg) {
  const int ndim = g->ndim, nvar = g->nvar, nv = g->nv;
  int ln, lt1, lt2; /* 0-based variable indices */
  if (dir == 0) { ln = 1; lt1 = 2; lt2 = 3; }
  else if (dir == 1) { ln = 2; lt1 = 1; lt2 = 3; }
  else { ln = 3; lt1 = 1; lt2 = 2; }
  const int si = dir == 0, sj = dir == 1, sk = dir == 2;
  for (int k = klo; k <= khi; k++)
    for (int j = jlo; j <= jhi; j++)
      for (int i = ilo; i <= ihi; i++) {
        for (int l = 0; l < ngrid; l++) {
          ql[l + nv * 0] = qm[CID(g, l, i - si, j - sj, k - sk, 0, dir)];
          qr[l + nv * 0] = qp[CID(g, l, i, j, k, 0, dir)];
          ql[l + nv * 1] = qm[CID(g, l, i - si, j - sj, k - sk, ln, dir)];
          qr[l + nv * 1] = qp[CID(g, l, i, j, k, ln, dir)];
          ql[l + nv * 2] = qm[CID(g, l, i - si, j - sj, k - sk, ndim + 1, dir)];
          qr[l + nv * 2] = qp[CID(g, l, i, j, k, ndim + 1, dir)];
          if (ndim > 1) {
            ql[l + nv * 3] = qm[CID(g, l, i - si, j - sj, k - sk, lt1, dir)];
            qr[l + nv * 3] = qp[CID(g, l, i, j, k, lt1, dir)];
          }
          if (ndim > 2) {
            ql[l + nv * 4] = qm[CID(g, l, i - si, j - sj, k - sk, lt2, dir)];
            qr[l + nv * 4] = qp[CID(g, l, i, j, k, lt2, dir)];
          }
          for (int n = ndim + 2; n < nvar; n++) {
            ql[l + nv * n] = qm[CID(g, l, i - si, j - sj, k - sk, n, dir)];
            qr[l + nv * n] = qp[CID(g, l, i, j, k, n, dir)];
          }
        }
        ora_riemann(p, ql, qr, fg, ngrid, nv);
        for (int l = 0; l < ngrid; l++) {
          flx[CI(g, l, i, j, k, 0)] = fg[l + nv * 0];
          flx[CI(g, l, i, j, k, ln)] = fg[l + nv * 1];
          if (ndim > 1) flx[CI(g, l, i, j, k, lt1)] = fg[l + nv * 3];
          if (ndim > 2) flx[CI(g, l, i, j, k, lt2)] = fg[l + nv * 4];
          flx[CI(g, l, i, j, k, ndim + 1)] = fg[l + nv * 2];
          for (int n = ndim + 2; n < nvar; n++) flx[CI(g, l, i, j, k, n)] = fg[l + nv * n];
          tx[CI(g, l, i, j, k, 0)] = 0.5 * (ql[l + nv * 1] + qr[l + nv * 1]);
          tx[CI(g, l, i, j, k, 1)] = fg[l + nv * nvar];
        }
      }
}

/* ------------------------------------------------------------------------
 * cmpdivu / consup -- hydro/uplmde.f90:702-764, 769-866 (difmag>0 only)
 * ---------------------------------------------------------------------- */
static inline size_t DIVI(const patch_t *g, int l, int i, int j, int k) {
  return (size_t)l + (size_t)g->nv * ((size_t)(i - 1) + (size_t)g->FI * ((size_t)(j - 1) + (size_t)g->FJ * (size_t)(k - 1)));
}
static void ora_cmpdivu(const patch_t *g, const double *q, double *div,
                        const double dxs[3], int ngrid) {
  const int ndim = g->ndim;
  double hp = 1.0;
  for (int d = 1; d < ndim; d++) hp = hp * 0.5; /* half**(ndim-1), exact */
  double fx = hp / dxs[0], fy = hp / dxs[1], fz = hp / dxs[2];
  for (int k = g->kf1; k <= g->kf2; k++)
    for (int j = g->jf1; j <= g->jf2; j++)
      for (int i = g->if1; i <= g->if2; i++)
        for (int l = 0; l < ngrid; l++) {
          double ux = 0.0, vy = 0.0, wz = 0.0;
#define Q(ii, jj, kk, n) q[CI(g, l, ii, jj, kk, n)]
          ux = ux + fx * (Q(i, j, k, 1) - Q(i - 1, j, k, 1));
          if (ndim > 1) {
            ux = ux + fx * (Q(i, j - 1, k, 1) - Q(i - 1, j - 1, k, 1));
            vy = vy + fy * (Q(i, j, k, 2) - Q(i, j - 1, k, 2) + Q(i - 1, j, k, 2) - Q(i - 1, j - 1, k, 2));
          }
          if (ndim > 2) {
            ux = ux + fx * (Q(i, j, k - 1, 1) - Q(i - 1, j, k - 1, 1) + Q(i, j - 1, k - 1, 1) - Q(i - 1, j - 1, k - 1, 1));
            vy = vy + fy * (Q(i, j, k - 1, 2) - Q(i, j - 1, k - 1, 2) + Q(i - 1, j, k - 1, 2) - Q(i - 1, j - 1, k - 1, 2));
            wz = wz + fz * (Q(i, j, k, 3) - Q(i, j, k - 1, 3) + Q(i, j - 1, k, 3) - Q(i, j - 1, k - 1, 3) +
                            Q(i - 1, j, k, 3) - Q(i - 1, j, k - 1, 3) + Q(i - 1, j - 1, k, 3) - Q(i - 1, j - 1, k - 1, 3));
          }
#undef Q
          div[DIVI(g, l, i, j, k)] = ux + vy + wz;
        }
}
static void ora_consup(const ora_hydro_params *p, const patch_t *g,
                       const double *uin, double *flux, const double *div,
                       double dt, int ngrid) {
  const int ndim = g->ndim, nvar = g->nvar;
  double factor = 1.0;
  for (int d = 1; d < ndim; d++) factor = factor * 0.5;
  for (int n = 0; n < nvar; n++) {
    for (int k = g->kf1; k <= imax(g->kf1, g->ku2 - 2); k++)
      for (int j = g->jf1; j <= imax(g->jf1, g->ju2 - 2); j++)
        for (int i = g->if1; i <= g->if2; i++)
          for (int l = 0; l < ngrid; l++) {
            double div1 = factor * div[DIVI(g, l, i, j, k)];
            if (ndim > 1) div1 = div1 + factor * div[DIVI(g, l, i, j + 1, k)];
            if (ndim > 2) div1 = div1 + factor * (div[DIVI(g, l, i, j, k + 1)] + div[DIVI(g, l, i, j + 1, k + 1)]);
            div1 = p->difmag * dmin(0.0, div1);
            size_t f = FIX(g, l, i, j, k, n, 0, nvar);
            flux[f] = flux[f] + dt * div1 * (uin[CI(g, l, i, j, k, n)] - uin[CI(g, l, i - 1, j, k, n)]);
          }
    if (ndim > 1)
      for (int k = g->kf1; k <= imax(g->kf1, g->ku2 - 2); k++)
        for (int j = g->jf1; j <= g->jf2; j++)
          for (int i = g->iu1 + 2; i <= g->iu2 - 2; i++)
            for (int l = 0; l < ngrid; l++) {
              double div1 = 0.0;
              div1 = div1 + factor * (div[DIVI(g, l, i, j, k)] + div[DIVI(g, l, i + 1, j, k)]);
              if (ndim > 2) div1 = div1 + factor * (div[DIVI(g, l, i, j, k + 1)] + div[DIVI(g, l, i + 1, j, k + 1)]);
              div1 = p->difmag * dmin(0.0, div1);
              size_t f = FIX(g, l, i, j, k, n, 1, nvar);
              flux[f] = flux[f] + dt * div1 * (uin[CI(g, l, i, j, k, n)] - uin[CI(g, l, i, j - 1, k, n)]);
            }
    if (ndim > 2)
      for (int k = g->kf1; k <= g->kf2; k++)
        for (int j = g->ju1 + 2; j <= g->ju2 - 2; j++)
          for (int i = g->iu1 + 2; i <= g->iu2 - 2; i++)
            for (int l = 0; l < ngrid; l++) {
              double div1 = factor * (div[DIVI(g, l, i, j, k)] + div[DIVI(g, l, i + 1, j, k)] +
                                      div[DIVI(g, l, i, j + 1, k)] + div[DIVI(g, l, i + 1, j + 1, k)]);
              div1 = p->difmag * dmin(0.0, div1);
              size_t f = FIX(g, l, i, j, k, n, 2, nvar);
              flux[f] = flux[f] + dt * div1 * (uin[CI(g, l, i, j, k, n)] - uin[CI(g, l, i, j, k - 1, n)]);
            }
  }
}

/* PLMDE tracing lives in hydro_oracle_plmde.c */
void ora_trace_plmde(const ora_hydro_params *p, int nv, const double *q,
                     const double *dq, const double *c, double *qm, double *qp,
                     const double dxs[3], double dt, int ngrid);

/* ------------------------------------------------------------------------
 * unsplit -- hydro/umuscl.f90:22-171
 * ---------------------------------------------------------------------- */
void ora_unsplit(const ora_hydro_params *p, const double *uin,
                 const double *gravin, double *flux, double *tmp, double dx,
                 double dy, double dz, double dt, int ngrid, int nvector) {
  patch_t gg = make_patch(p->ndim, p->nvar, nvector);
  const patch_t *g = &gg;
  const int ndim = g->ndim, nvar = g->nvar, nv = g->nv;
  const size_t ncell = (size_t)nv * g->NI * g->NJ * g->NK;
  double *q = (double *)calloc(ncell * nvar, sizeof(double));
  double *c = (double *)calloc(ncell, sizeof(double));
  double *dq = (double *)calloc(ncell * nvar * ndim, sizeof(double));
  double *qm = (double *)calloc(ncell * nvar * ndim, sizeof(double));
  double *qp = (double *)calloc(ncell * nvar * ndim, sizeof(double));
  double *fx = (double *)calloc(ncell * nvar, sizeof(double));
  double *tx = (double *)calloc(ncell * 2, sizeof(double));
  double *ql = (double *)calloc((size_t)nv * (nvar + 1), sizeof(double));
  double *qr = (double *)calloc((size_t)nv * (nvar + 1), sizeof(double));
  double *fg = (double *)calloc((size_t)nv * (nvar + 1), sizeof(double));
  const double dxs[3] = {dx, dy, dz};

  const int ilo = imin(1, g->iu1 + 2), ihi = imax(1, g->iu2 - 2);
  const int jlo = imin(1, g->ju1 + 2), jhi = imax(1, g->ju2 - 2);
  const int klo = imin(1, g->ku1 + 2), khi = imax(1, g->ku2 - 2);

  ora_ctoprim(p, g, uin, q, c, gravin, dt, ngrid);
  ora_uslope(p, g, q, dq, dx, dt, ngrid);
  if (p->scheme == ORA_SCHEME_MUSCL) ora_trace(p, g, q, dq, qm, qp, dxs, dt, ngrid);
  else ora_trace_plmde(p, nv, q, dq, c, qm, qp, dxs, dt, ngrid);

  for (int dir = 0; dir < ndim; dir++) {
    int a0 = ilo, a1 = ihi, b0 = jlo, b1 = jhi, c0 = klo, c1 = khi;
    if (dir == 0) { a0 = g->if1; a1 = g->if2; }
    if (dir == 1) { b0 = g->jf1; b1 = g->jf2; }
    if (dir == 2) { c0 = g->kf1; c1 = g->kf2; }
    ora_cmpflxm(p, g, qm, qp, dir, a0, a1, b0, b1, c0, c1, fx, tx, ngrid, ql, qr, fg);
    for (int i = a0; i <= a1; i++)
      for (int j = b0; j <= b1; j++)
        for (int k = c0; k <= c1; k++) {
          for (int n = 0; n < nvar; n++)
            for (int l = 0; l < ngrid; l++)
              flux[FIX(g, l, i, j, k, n, dir, nvar)] = fx[CI(g, l, i, j, k, n)] * dt / dxs[dir];
          for (int n = 0; n < 2; n++)
            for (int l = 0; l < ngrid; l++)
              tmp[FIX(g, l, i, j, k, n, dir, 2)] = tx[CI(g, l, i, j, k, n)] * dt / dxs[dir];
        }
  }
  if (p->difmag > 0.0) {
    double *div = (double *)calloc((size_t)nv * 27, sizeof(double));
    ora_cmpdivu(g, q, div, dxs, ngrid);
    ora_consup(p, g, uin, flux, div, dt, ngrid);
    free(div);
  }
  free(q); free(c); free(dq); free(qm); free(qp); free(fx); free(tx);
  free(ql); free(qr); free(fg);
}

/* ------------------------------------------------------------------------
 * godunov_fine + godfine1 on a fully refined periodic level --
 * hydro/godunov_fine.f90:5-35, 486-911 with every son()>0 neighbour present
 * (nbuffer=0, ok=.false.), so the interpolation and coarse-level branches
 * are not taken.
 * ---------------------------------------------------------------------- */
void ora_godunov_uniform(const ora_hydro_params *p, const double *uold,
                         const double *grav, double *unew, int nx, int ny,
                         int nz, double dx, double dt) {
  const int ndim = p->ndim, nvar = p->nvar;
  const int NV = 32; /* nvector */
  patch_t gg = make_patch(ndim, nvar, NV);
  const patch_t *g = &gg;
  const size_t ncell = (size_t)NV * g->NI * g->NJ * g->NK;
  const size_t nface = (size_t)NV * g->FI * g->FJ * g->FK;
  double *uloc = (double *)calloc(ncell * nvar, sizeof(double));
  double *gloc = (double *)calloc(ncell * ndim, sizeof(double));
  double *flux = (double *)calloc(nface * nvar * ndim, sizeof(double));
  double *tmp = (double *)calloc(nface * 2 * ndim, sizeof(double));
  const size_t N = (size_t)nx * ny * nz;
  const int ox = nx / 2, oy = ndim > 1 ? ny / 2 : 1, oz = ndim > 2 ? nz / 2 : 1;
  const long noct = (long)ox * oy * oz;
  int *octs = (int *)malloc(sizeof(int) * 3 * NV);

  for (long o0 = 0; o0 < noct; o0 += NV) {
    int ngrid = (int)((noct - o0) < NV ? (noct - o0) : NV);
    for (int l = 0; l < ngrid; l++) {
      long o = o0 + l;
      octs[3 * l + 0] = (int)(o % ox);
      octs[3 * l + 1] = (int)((o / ox) % oy);
      octs[3 * l + 2] = (int)(o / ((long)ox * oy));
    }
    /* gather the 6^ndim stencil: godunov_fine.f90:562-675 */
    for (int l = 0; l < ngrid; l++) {
      int bx = 2 * octs[3 * l + 0], by = 2 * octs[3 * l + 1], bz = 2 * octs[3 * l + 2];
      for (int k3 = g->ku1; k3 <= g->ku2; k3++)
        for (int j3 = g->ju1; j3 <= g->ju2; j3++)
          for (int i3 = g->iu1; i3 <= g->iu2; i3++) {
            int ix = ((bx + i3 - 1) % nx + nx) % nx;
            int iy = ndim > 1 ? ((by + j3 - 1) % ny + ny) % ny : 0;
            int iz = ndim > 2 ? ((bz + k3 - 1) % nz + nz) % nz : 0;
            size_t cell = (size_t)ix + (size_t)nx * ((size_t)iy + (size_t)ny * (size_t)iz);
            for (int n = 0; n < nvar; n++) uloc[CI(g, l, i3, j3, k3, n)] = uold[cell + N * n];
            for (int d = 0; d < ndim; d++) gloc[CI(g, l, i3, j3, k3, d)] = grav ? grav[cell + N * d] : 0.0;
          }
    }
    ora_unsplit(p, uloc, gloc, flux, tmp, dx, dx, dx, dt, ngrid, NV);
    /* conservative update: godunov_fine.f90:751-792 (idim outer, octant, ivar) */
    for (int d = 0; d < ndim; d++) {
      int i0 = d == 0, j0 = d == 1, k0 = d == 2;
      for (int k2 = 0; k2 <= (ndim > 2); k2++)
        for (int j2 = 0; j2 <= (ndim > 1); j2++)
          for (int i2 = 0; i2 <= 1; i2++) {
            int i3 = 1 + i2, j3 = 1 + j2, k3 = 1 + k2;
            for (int n = 0; n < nvar; n++)
              for (int l = 0; l < ngrid; l++) {
                int ix = 2 * octs[3 * l + 0] + i2;
                int iy = ndim > 1 ? 2 * octs[3 * l + 1] + j2 : 0;
                int iz = ndim > 2 ? 2 * octs[3 * l + 2] + k2 : 0;
                size_t cell = (size_t)ix + (size_t)nx * ((size_t)iy + (size_t)ny * (size_t)iz);
                unew[cell + N * n] = unew[cell + N * n] +
                    (flux[FIX(g, l, i3, j3, k3, n, d, nvar)] - flux[FIX(g, l, i3 + i0, j3 + j0, k3 + k0, n, d, nvar)]);
              }
          }
    }
  }
  free(uloc); free(gloc); free(flux); free(tmp); free(octs);
}

/* ------------------------------------------------------------------------
 * cmpdt -- hydro/godunov_utils.f90:5-120, looped as in courant_fine
 * (hydro/courant_fine.f90:44-131); min() is order independent.
 * ---------------------------------------------------------------------- */
double ora_courant_uniform(const ora_hydro_params *p, const double *uold,
                           const double *grav, int nx, int ny, int nz,
                           double dx, double courant_factor) {
  const int ndim = p->ndim;
  const size_t N = (size_t)nx * ny * nz;
  const double smallp = p->smallc * p->smallc / p->gamma;
  double dt = courant_factor * dx / p->smallc;
  for (size_t cidx = 0; cidx < N; cidx++) {
    double uu[MAXVAR];
    for (int n = 0; n < ndim + 2; n++) uu[n] = uold[cidx + N * n];
    uu[0] = dmax(uu[0], p->smallr);
    for (int d = 0; d < ndim; d++) uu[d + 1] = uu[d + 1] / uu[0];
    for (int d = 0; d < ndim; d++) uu[ndim + 1] = uu[ndim + 1] - 0.5 * uu[0] * (uu[d + 1] * uu[d + 1]);
    uu[ndim + 1] = dmax((p->gamma - 1.0) * uu[ndim + 1], uu[0] * smallp);
    uu[ndim + 1] = p->gamma * uu[ndim + 1];
    uu[ndim + 1] = sqrt(uu[ndim + 1] / uu[0]);
    uu[ndim + 1] = (double)ndim * uu[ndim + 1];
    for (int d = 0; d < ndim; d++) uu[ndim + 1] = uu[ndim + 1] + fabs(uu[d + 1]);
    uu[0] = 0.0;
    for (int d = 0; d < ndim; d++) uu[0] = uu[0] + fabs(grav ? grav[cidx + N * d] : 0.0);
    uu[0] = uu[0] * dx / (uu[ndim + 1] * uu[ndim + 1]);
    uu[0] = dmax(uu[0], 0.0001);
    double dtcell = dx / uu[ndim + 1] * (sqrt(1.0 + 2.0 * courant_factor * uu[0]) - 1.0) / uu[0];
    dt = dmin(dt, dtcell);
  }
  return dt;
}
