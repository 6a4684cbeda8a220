/*
 * amr_oracle.c -- TEST INFRASTRUCTURE ONLY (see hydro_oracle.h for the rules).
 *
 * CPU restatement of the coarse<->fine hydro operators of the reference:
 *   interpol_hydro + compute_limiter_minmod/_central, compute_central
 *                         hydro/interpol_hydro.f90:268-444, 449-637
 *   upl (restriction of upload_fine)        hydro/interpol_hydro.f90:73-263
 * 3-D, NENER=0.  Arrays use the reference's Fortran layouts with leading
 * dimension nvector:  u1(nvector,0:6,nvar), u2(nvector,1:8,nvar).
 * Parity status: PINNED bit-for-bit against the reference's own routines
 * (oracle/_ref/libref_kernels3d.so, tests/test_amr_ops.py).
 */
#include <math.h>
#include <stdlib.h>

#define U1(i, j, v) u1[(size_t)(i) + (size_t)nv * ((size_t)(j) + 7 * (size_t)(v))]
#define U2(i, c, v) u2[(size_t)(i) + (size_t)nv * ((size_t)(c) + 8 * (size_t)(v))]

static inline double dmax(double a, double b) { return a > b ? a : b; }
static inline double dmin(double a, double b) { return a < b ? a : b; }

static void limiter_minmod(const double a[7], double w[3]) {
  for (int d = 0; d < 3; d++) {
    double dl = 0.5 * (a[2 * d + 2] - a[0]);
    double dr = 0.5 * (a[0] - a[2 * d + 1]);
    double mm;
    if (dl * dr <= 0.0) mm = 0;
    else mm = dmin(fabs(dl), fabs(dr)) * dl / fabs(dl);
    w[d] = mm;
  }
}
static void central(const double a[7], double w[3]) {
  for (int d = 0; d < 3; d++) w[d] = 0.25 * (a[2 * d + 2] - a[2 * d + 1]);
}
static void limiter_central(const double a[7], double w[3]) {
  double xc[8][3], ac[8];
  for (int ind = 0; ind < 8; ind++) {
    xc[ind][0] = (double)(ind & 1) - 0.5;
    xc[ind][1] = (double)((ind >> 1) & 1) - 0.5;
    xc[ind][2] = (double)((ind >> 2) & 1) - 0.5;
  }
  central(a, w);
  for (int ind = 0; ind < 8; ind++) ac[ind] = a[0];
  for (int d = 0; d < 3; d++)
    for (int ind = 0; ind < 8; ind++) ac[ind] = ac[ind] + 2.0 * w[d] * xc[ind][d];
  double corner = ac[0], kernel = a[1];
  for (int j = 1; j < 8; j++) corner = dmax(corner, ac[j]);
  for (int j = 2; j <= 6; j++) kernel = dmax(kernel, a[j]);
  double dk = a[0] - kernel, dc = a[0] - corner;
  double max_lim = 0.0;
  if (dk * dc > 0.0) max_lim = dmin(1.0, dk / dc);
  corner = ac[0]; kernel = a[1];
  for (int j = 1; j < 8; j++) corner = dmin(corner, ac[j]);
  for (int j = 2; j <= 6; j++) kernel = dmin(kernel, a[j]);
  dk = a[0] - kernel; dc = a[0] - corner;
  double min_lim = 0.0;
  if (dk * dc > 0.0) min_lim = dmin(1.0, dk / dc);
  double lim = dmin(min_lim, max_lim);
  for (int d = 0; d < 3; d++) w[d] = w[d] * lim;
}

/* u1 is modified in place exactly like the reference does (internal energy /
 * velocity conversion of the stencil when interpol_var = 1 or 2). */
void ora_interpol_hydro(double *u1, double *u2, int nn, int nv, int nvar, int interpol_var,
                        int interpol_type, double smallr) {
  const int ndim = 3;
  double xc[8][3];
  for (int ind = 0; ind < 8; ind++) {
    xc[ind][0] = (double)(ind & 1) - 0.5;
    xc[ind][1] = (double)((ind >> 1) & 1) - 0.5;
    xc[ind][2] = (double)((ind >> 2) & 1) - 0.5;
  }
  if (interpol_var == 1 || interpol_var == 2) {
    for (int j = 0; j <= 6; j++)
      for (int i = 0; i < nn; i++) {
        double ekin = 0.0;
        for (int d = 0; d < ndim; d++) ekin = ekin + 0.5 * (U1(i, j, d + 1) * U1(i, j, d + 1)) / dmax(U1(i, j, 0), smallr);
        U1(i, j, ndim + 1) = U1(i, j, ndim + 1) - ekin - 0.0;
        if (interpol_var == 2)
          for (int d = 0; d < ndim; d++) U1(i, j, d + 1) = U1(i, j, d + 1) / dmax(U1(i, j, 0), smallr);
      }
  }
  for (int v = 0; v < nvar; v++)
    for (int i = 0; i < nn; i++) {
      double a[7], w[3] = {0, 0, 0};
      for (int j = 0; j <= 6; j++) a[j] = U1(i, j, v);
      if (interpol_type == 1) limiter_minmod(a, w);
      if (interpol_type == 2) limiter_central(a, w);
      if (interpol_type == 3) central(a, w);
      if (interpol_type == 4) {
        if (v >= 1 && v <= ndim) central(a, w);
        else limiter_central(a, w);
      }
      for (int ind = 0; ind < 8; ind++) {
        double val = a[0];
        for (int d = 0; d < ndim; d++) val = val + w[d] * xc[ind][d];
        U2(i, ind, v) = val;
      }
    }
  if (interpol_var == 1 || interpol_var == 2) {
    if (interpol_var == 2) {
      for (int i = 0; i < nn; i++) {
        for (int ind = 0; ind < 8; ind++)
          for (int d = 0; d < ndim; d++) U2(i, ind, d + 1) = U2(i, ind, d + 1) * U2(i, ind, 0);
        for (int d = 0; d < ndim; d++) {
          double mom = 0;
          for (int ind = 0; ind < 8; ind++) mom = mom + U2(i, ind, d + 1) * 0.125;
          mom = mom - U1(i, 0, d + 1) * U1(i, 0, 0);
          for (int ind = 0; ind < 8; ind++) U2(i, ind, d + 1) = U2(i, ind, d + 1) - mom;
        }
      }
    }
    for (int ind = 0; ind < 8; ind++)
      for (int i = 0; i < nn; i++) {
        double ekin = 0.0;
        for (int d = 0; d < ndim; d++) ekin = ekin + 0.5 * (U2(i, ind, d + 1) * U2(i, ind, d + 1)) / dmax(U2(i, ind, 0), smallr);
        U2(i, ind, ndim + 1) = U2(i, ind, ndim + 1) + ekin + 0.0;
      }
  }
}

/* upl on gathered data: child(nv, 8, nvar) = the 8 son cells, parent_in/out(nv, nvar)
 * (parent_in supplies the coarse momenta/density used for the kinetic energy when
 * interpol_var is 1 or 2 -- the reference reads them AFTER overwriting them with
 * the restricted values, so parent_in is ignored there). */
void ora_upl(const double *child, double *parent, int nn, int nv, int nvar, int interpol_var,
             double smallr) {
  const int ndim = 3;
#define CH(i, c, v) child[(size_t)(i) + (size_t)nv * ((size_t)(c) + 8 * (size_t)(v))]
#define PA(i, v) parent[(size_t)(i) + (size_t)nv * (size_t)(v)]
  for (int i = 0; i < nn; i++) {
    double getx = 0.0;
    for (int c = 0; c < 8; c++) getx = getx + dmax(CH(i, c, 0), smallr);
    PA(i, 0) = getx / 8.0;
    for (int v = 1; v < nvar; v++) {
      getx = 0.0;
      for (int c = 0; c < 8; c++) getx = getx + CH(i, c, v);
      PA(i, v) = getx / 8.0;
    }
    if (interpol_var == 1 || interpol_var == 2) {
      getx = 0.0;
      for (int c = 0; c < 8; c++) {
        double ekin = 0.0;
        for (int d = 0; d < ndim; d++) ekin = ekin + 0.5 * (CH(i, c, 1 + d) * CH(i, c, 1 + d)) / dmax(CH(i, c, 0), smallr);
        getx = getx + CH(i, c, ndim + 1) - ekin - 0.0;
      }
      double ekin = 0.0;
      for (int d = 0; d < ndim; d++) ekin = ekin + 0.5 * (PA(i, 1 + d) * PA(i, 1 + d)) / dmax(PA(i, 0), smallr);
      PA(i, ndim + 1) = getx / 8.0 + ekin + 0.0;
    }
  }
}
