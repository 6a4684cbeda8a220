/*
 * amr_godfine_oracle.c -- TEST INFRASTRUCTURE ONLY (see hydro_oracle.h).
 *
 * CPU restatement of godunov_fine/godfine1 on an AMR level, 3-D, directly on the
 * reference's tree arrays, in the reference's loop order:
 *   godunov_fine        hydro/godunov_fine.f90:5-35     (batches of nvector octs)
 *   godfine1            hydro/godunov_fine.f90:486-911
 *   get3cubefather      amr/nbors_utils.f90:5-194       (as x,y,z steps through son(nbor))
 *   getnborfather       amr/nbors_utils.f90:404-525
 * interpol_hydro and unsplit are the restatements of amr_oracle.c / hydro_oracle.c.
 *
 * Parity status: PINNED against dumps of the unmodified reference
 * (oracle/dump_patch -> tests/golden/amr_godunov_ref.npz, tests/test_amr_oracle.py).
 */
#include <stdlib.h>
#include <string.h>

#include "hydro_oracle.h"

void ora_interpol_hydro(double *u1, double *u2, int nn, int nv, int nvar, int interpol_var, int interpol_type,
                        double smallr);

typedef struct {
  const int *son, *nbor, *father;
  long ncoarse, ngridmax, ncell;
} tree_t;

/* same-level neighbour of cell c (1-based) in direction dir (0:-x 1:+x 2:-y ...); if its oct does not
 * exist: minus the neighbouring father cell of c's oct (getnborfather's fallback) */
static int nbor_cell(const tree_t *T, int c, int dir) {
  const int pos = (int)((c - T->ncoarse - 1) / T->ngridmax);
  const int g = (int)(c - T->ncoarse - (long)pos * T->ngridmax);
  const int axis = dir >> 1, up = dir & 1;
  const int bit = (pos >> axis) & 1;
  if (bit != up) return c + (up ? 1 : -1) * (int)((1 << axis) * T->ngridmax);
  const int nb = T->nbor[(long)dir * T->ngridmax + g - 1];
  const int g2 = T->son[nb - 1];
  if (g2 == 0) return -nb;
  return (int)(T->ncoarse + (long)(pos ^ (1 << axis)) * T->ngridmax + g2);
}

/* uloc/flux index helpers: the reference's Fortran layout with leading dimension nv */
#define UL(l, i, j, k, v) uloc[(size_t)(l) + (size_t)nv * ((size_t)((i) + 1) + 6 * ((size_t)((j) + 1) + 6 * ((size_t)((k) + 1) + 6 * (size_t)(v))))]
#define GL(l, i, j, k, d) gloc[(size_t)(l) + (size_t)nv * ((size_t)((i) + 1) + 6 * ((size_t)((j) + 1) + 6 * ((size_t)((k) + 1) + 6 * (size_t)(d))))]
#define OK(l, i, j, k) ok[(size_t)(l) + (size_t)nv * ((size_t)((i) + 1) + 6 * ((size_t)((j) + 1) + 6 * (size_t)((k) + 1)))]
#define FX(l, i, j, k, v, d) flux[(size_t)(l) + (size_t)nv * ((size_t)((i) - 1) + 3 * ((size_t)((j) - 1) + 3 * ((size_t)((k) - 1) + 3 * ((size_t)(v) + (size_t)nvar * (size_t)(d)))))]
#define TP(l, i, j, k, t, d) tmp[(size_t)(l) + (size_t)nv * ((size_t)((i) - 1) + 3 * ((size_t)((j) - 1) + 3 * ((size_t)((k) - 1) + 3 * ((size_t)(t) + 2 * (size_t)(d)))))]

/* uold, unew: [nvar][ncell]; f: [3][ncell] or NULL; divu, enew: [ncell] or NULL (pressure_fix) */
void ora_godunov_fine_amr(const ora_hydro_params *p, int ngrid_tot, const int *igrid, const int *son, const int *nbor,
                          const int *father, long ngridmax, long ncoarse, const double *uold, double *unew,
                          const double *f, double *divu, double *enew, double dx, double dt, int nvector,
                          int interpol_var, int interpol_type) {
  const int nvar = p->nvar, nv = nvector;
  tree_t TT = {son, nbor, father, ncoarse, ngridmax, ncoarse + 8 * ngridmax};
  const tree_t *T = &TT;
  const long ncell = T->ncell;
  const size_t ncl = (size_t)nv * 216, nfc = (size_t)nv * 27;
  double *uloc = (double *)calloc(ncl * nvar, sizeof(double));
  double *gloc = (double *)calloc(ncl * 3, sizeof(double));
  unsigned char *ok = (unsigned char *)calloc(ncl, 1);
  double *flux = (double *)calloc(nfc * nvar * 3, sizeof(double));
  double *tmp = (double *)calloc(nfc * 2 * 3, sizeof(double));
  double *u1 = (double *)calloc((size_t)nv * 7 * nvar, sizeof(double));
  double *u2 = (double *)calloc((size_t)nv * 8 * nvar, sizeof(double));
  int *fc = (int *)malloc(sizeof(int) * (size_t)nv * 27);
  int *ibuf = (int *)malloc(sizeof(int) * (size_t)nv * 7);
  int *ind_nexist = (int *)malloc(sizeof(int) * nv);
  const double oneontwotondim = 1.0 / 8.0;

  for (int i0 = 0; i0 < ngrid_tot; i0 += nv) {
    const int ncache = (ngrid_tot - i0) < nv ? (ngrid_tot - i0) : nv;
    const int *ind_grid = igrid + i0;
    /* the 3^3 neighbouring father cells */
    for (int l = 0; l < ncache; l++)
      for (int t = 0; t < 27; t++) {
        const int d3[3] = {t % 3 - 1, (t / 3) % 3 - 1, t / 9 - 1};
        int c = father[ind_grid[l] - 1];
        for (int axis = 0; axis < 3; axis++)
          if (d3[axis] != 0) c = nbor_cell(T, c, 2 * axis + (d3[axis] > 0 ? 1 : 0));
        fc[l * 27 + t] = c;
      }
    /* gather the 6^3 stencil, father cell by father cell (godunov_fine.f90:562-676) */
    for (int k1 = 0; k1 < 3; k1++)
      for (int j1 = 0; j1 < 3; j1++)
        for (int i1 = 0; i1 < 3; i1++) {
          const int t = i1 + 3 * j1 + 9 * k1;
          int nbuffer = 0;
          for (int l = 0; l < ncache; l++)
            if (son[fc[l * 27 + t] - 1] == 0) {
              ind_nexist[nbuffer] = l;
              /* father cell and its 2*ndim neighbours, coarser cell where the neighbour does not exist */
              const int c0 = fc[l * 27 + t];
              ibuf[nbuffer * 7] = c0;
              for (int j = 1; j <= 6; j++) {
                int c = nbor_cell(T, c0, j - 1);
                ibuf[nbuffer * 7 + j] = c < 0 ? -c : c;
              }
              nbuffer++;
            }
          if (nbuffer > 0) {
            for (int j = 0; j < 7; j++)
              for (int v = 0; v < nvar; v++)
                for (int b = 0; b < nbuffer; b++)
                  u1[(size_t)b + (size_t)nv * ((size_t)j + 7 * (size_t)v)] = uold[(size_t)v * ncell + ibuf[b * 7 + j] - 1];
            ora_interpol_hydro(u1, u2, nbuffer, nv, nvar, interpol_var, interpol_type, p->smallr);
          }
          for (int k2 = 0; k2 < 2; k2++)
            for (int j2 = 0; j2 < 2; j2++)
              for (int i2 = 0; i2 < 2; i2++) {
                const int ind = i2 + 2 * j2 + 4 * k2;
                const int i3 = 1 + 2 * (i1 - 1) + i2, j3 = 1 + 2 * (j1 - 1) + j2, k3 = 1 + 2 * (k1 - 1) + k2;
                for (int l = 0; l < ncache; l++) {
                  const int og = son[fc[l * 27 + t] - 1];
                  if (og > 0) {
                    const long cell = ncoarse + (long)ind * ngridmax + og;
                    for (int v = 0; v < nvar; v++) UL(l, i3, j3, k3, v) = uold[(size_t)v * ncell + cell - 1];
                    if (f) for (int d = 0; d < 3; d++) GL(l, i3, j3, k3, d) = f[(size_t)d * ncell + cell - 1];
                    OK(l, i3, j3, k3) = son[cell - 1] > 0;
                  }
                }
                for (int b = 0; b < nbuffer; b++) {
                  const int l = ind_nexist[b];
                  for (int v = 0; v < nvar; v++) UL(l, i3, j3, k3, v) = u2[(size_t)b + (size_t)nv * ((size_t)ind + 8 * (size_t)v)];
                  if (f) for (int d = 0; d < 3; d++) GL(l, i3, j3, k3, d) = f[(size_t)d * ncell + ibuf[b * 7] - 1];
                  OK(l, i3, j3, k3) = 0;
                }
              }
        }
    ora_unsplit(p, uloc, gloc, flux, tmp, dx, dx, dx, dt, ncache, nv);
    /* reset the fluxes at refined interfaces (:720-747) */
    for (int d = 0; d < 3; d++) {
      const int i0_ = d == 0, j0_ = d == 1, k0_ = d == 2;
      for (int k3 = 1; k3 <= 2 + k0_; k3++)
        for (int j3 = 1; j3 <= 2 + j0_; j3++)
          for (int i3 = 1; i3 <= 2 + i0_; i3++)
            for (int l = 0; l < ncache; l++)
              if (OK(l, i3 - i0_, j3 - j0_, k3 - k0_) || OK(l, i3, j3, k3)) {
                for (int v = 0; v < nvar; v++) FX(l, i3, j3, k3, v, d) = 0.0;
                if (divu) { TP(l, i3, j3, k3, 0, d) = 0.0; TP(l, i3, j3, k3, 1, d) = 0.0; }
              }
    }
    /* conservative update at level ilevel (:751-792) */
    for (int d = 0; d < 3; d++) {
      const int i0_ = d == 0, j0_ = d == 1, k0_ = d == 2;
      for (int k2 = 0; k2 < 2; k2++)
        for (int j2 = 0; j2 < 2; j2++)
          for (int i2 = 0; i2 < 2; i2++) {
            const int ind = i2 + 2 * j2 + 4 * k2;
            const int i3 = 1 + i2, j3 = 1 + j2, k3 = 1 + k2;
            for (int v = 0; v < nvar; v++)
              for (int l = 0; l < ncache; l++) {
                const long cell = ncoarse + (long)ind * ngridmax + ind_grid[l];
                double *u = &unew[(size_t)v * ncell + cell - 1];
                *u = *u + (FX(l, i3, j3, k3, v, d) - FX(l, i3 + i0_, j3 + j0_, k3 + k0_, v, d));
              }
            if (divu)
              for (int l = 0; l < ncache; l++) {
                const long cell = ncoarse + (long)ind * ngridmax + ind_grid[l];
                divu[cell - 1] = divu[cell - 1] + (TP(l, i3, j3, k3, 0, d) - TP(l, i3 + i0_, j3 + j0_, k3 + k0_, 0, d));
                enew[cell - 1] = enew[cell - 1] + (TP(l, i3, j3, k3, 1, d) - TP(l, i3 + i0_, j3 + j0_, k3 + k0_, 1, d));
              }
          }
    }
    /* conservative update at level ilevel-1 (:798-908) */
    for (int d = 0; d < 3; d++) {
      const int i0_ = d == 0, j0_ = d == 1, k0_ = d == 2;
      for (int side = 0; side < 2; side++) {
        for (int v = 0; v < nvar + (divu ? 2 : 0); v++) {
          const int klo = side ? 1 + k0_ : 1, khi = side ? 2 : 2 - k0_;
          const int jlo = side ? 1 + j0_ : 1, jhi = side ? 2 : 2 - j0_;
          const int ilo = side ? 1 + i0_ : 1, ihi = side ? 2 : 2 - i0_;
          for (int k3 = klo; k3 <= khi; k3++)
            for (int j3 = jlo; j3 <= jhi; j3++)
              for (int i3 = ilo; i3 <= ihi; i3++)
                for (int l = 0; l < ncache; l++) {
                  const int nb = nbor[(long)(2 * d + side) * ngridmax + ind_grid[l] - 1];
                  if (son[nb - 1] != 0) continue;
                  double *tgt = v < nvar ? &unew[(size_t)v * ncell + nb - 1] : (v == nvar ? &divu[nb - 1] : &enew[nb - 1]);
                  const int fi = i3 + (side ? i0_ : 0), fj = j3 + (side ? j0_ : 0), fk = k3 + (side ? k0_ : 0);
                  const double q = v < nvar ? FX(l, fi, fj, fk, v, d) : TP(l, fi, fj, fk, v - nvar, d);
                  if (side == 0) *tgt = *tgt - q * oneontwotondim;
                  else *tgt = *tgt + q * oneontwotondim;
                }
        }
      }
    }
  }
  free(uloc); free(gloc); free(ok); free(flux); free(tmp); free(u1); free(u2); free(fc); free(ibuf); free(ind_nexist);
}
