/* oracle/cg_oracle.c -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * CPU restatement of the conjugate-gradient Poisson solver's iteration loop
 * (poisson/phi_fine_cg.f90:88-187) and of cmp_Ap_cg (:344-447) on one AMR level, in the
 * reference's own loop order (sums run over ind = 1..8 outermost, then the level's octs in list
 * order), so the result is bit-identical to the reference's.  The state it starts from (phi after
 * the initial guess, r = p = f(:,1:2) after cmp_residual_cg) is the reference's own.
 * Pinned by tests/test_cg_oracle.py against dumps of the unmodified reference
 * (oracle/dump_patch/phi_fine_cg.f90 -> tests/golden/cg_ref.npz).
 *
 * Arrays are the reference's: cell = ncoarse + ind*ngridmax + igrid (1-based igrid, ind 0..7),
 * f = (3, ncell) variable-major, nbor = (6, ngridmax). */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

/* same-level neighbour tables (poisson/phi_fine_cg.f90:370-375): oct (0 = own, k = son(nbor(k)))
 * and octant of the left/right neighbour of octant ind in direction idim */
static const int III[3][2][8] = {{{1, 0, 1, 0, 1, 0, 1, 0}, {0, 2, 0, 2, 0, 2, 0, 2}},
                                 {{3, 3, 0, 0, 3, 3, 0, 0}, {0, 0, 4, 4, 0, 0, 4, 4}},
                                 {{5, 5, 5, 5, 0, 0, 0, 0}, {0, 0, 0, 0, 6, 6, 6, 6}}};
static const int JJJ[3][2][8] = {{{2, 1, 4, 3, 6, 5, 8, 7}, {2, 1, 4, 3, 6, 5, 8, 7}},
                                 {{3, 4, 1, 2, 7, 8, 5, 6}, {3, 4, 1, 2, 7, 8, 5, 6}},
                                 {{5, 6, 7, 8, 1, 2, 3, 4}, {5, 6, 7, 8, 1, 2, 3, 4}}};

/* cmp_Ap_cg (:344-447): f(:,3) = -p + (1/6) sum of the six neighbouring p (0 where the
 * neighbouring oct does not exist) */
static void cmp_ap(int ngrid, const int *igrid, const int *son, const int *nbor, int64_t ngridmax,
                   int64_t ncoarse, const double *p, double *ap) {
  const double oneoversix = 1.0 / 6.0;
  for (int i = 0; i < ngrid; i++) {
    const int g = igrid[i];
    int64_t gn[7];
    gn[0] = g;
    for (int k = 1; k <= 6; k++) {
      const int c = nbor[(int64_t)(k - 1) * ngridmax + g - 1];
      gn[k] = son[c - 1];
    }
    for (int ind = 0; ind < 8; ind++) {
      double r = -p[ncoarse + (int64_t)ind * ngridmax + g - 1];
      for (int idim = 0; idim < 3; idim++) {
        const int64_t g1 = gn[III[idim][0][ind]], g2 = gn[III[idim][1][ind]];
        const double pg = g1 > 0 ? p[ncoarse + (int64_t)(JJJ[idim][0][ind] - 1) * ngridmax + g1 - 1] : 0.0;
        const double pd = g2 > 0 ? p[ncoarse + (int64_t)(JJJ[idim][1][ind] - 1) * ngridmax + g2 - 1] : 0.0;
        r = r + oneoversix * (pg + pd);
      }
      ap[ncoarse + (int64_t)ind * ngridmax + g - 1] = r;
    }
  }
}

/* the iteration loop (:88-187), serial (one rank: the level's octs are all the rank's).
 * Returns the number of iterations; err[0] = last error (rms residual), err[1] = first. */
int ora_cg_solve(int ngrid, const int *igrid, const int *son, const int *nbor, int64_t ngridmax,
                 int64_t ncoarse, double *phi, double *f, double epsilon, int itermax, double *err) {
  const int64_t ncell = ncoarse + 8 * ngridmax;
  double *r = f, *p = f + ncell, *ap = f + 2 * ncell;
  int iter = 0;
  double error = 1.0, error_ini = 1.0, r2_old = 0.0;
  while (error > epsilon * error_ini && iter < itermax) {
    iter++;
    double r2 = 0.0;
    for (int ind = 0; ind < 8; ind++)
      for (int i = 0; i < ngrid; i++) {
        const int64_t c = ncoarse + (int64_t)ind * ngridmax + igrid[i] - 1;
        r2 = r2 + r[c] * r[c];
      }
    const double beta = iter == 1 ? 0.0 : r2 / r2_old;
    r2_old = r2;
    for (int ind = 0; ind < 8; ind++)
      for (int i = 0; i < ngrid; i++) {
        const int64_t c = ncoarse + (int64_t)ind * ngridmax + igrid[i] - 1;
        p[c] = r[c] + beta * p[c];
      }
    cmp_ap(ngrid, igrid, son, nbor, ngridmax, ncoarse, p, ap);
    double pap = 0.0;
    for (int ind = 0; ind < 8; ind++)
      for (int i = 0; i < ngrid; i++) {
        const int64_t c = ncoarse + (int64_t)ind * ngridmax + igrid[i] - 1;
        pap = pap + p[c] * ap[c];
      }
    const double alpha = r2 / pap;
    for (int ind = 0; ind < 8; ind++)
      for (int i = 0; i < ngrid; i++) {
        const int64_t c = ncoarse + (int64_t)ind * ngridmax + igrid[i] - 1;
        phi[c] = phi[c] + alpha * p[c];
      }
    for (int ind = 0; ind < 8; ind++)
      for (int i = 0; i < ngrid; i++) {
        const int64_t c = ncoarse + (int64_t)ind * ngridmax + igrid[i] - 1;
        r[c] = r[c] - alpha * ap[c];
      }
    error = sqrt(r2 / (double)(8 * (int64_t)ngrid));
    if (iter == 1) error_ini = error;
  }
  if (err) { err[0] = error; err[1] = error_ini; }
  return iter;
}
