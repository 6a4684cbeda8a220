/* oracle/mg_amr_oracle.c -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * CPU restatement of the fine-level multigrid routines on an AMR level, in the reference's own
 * arrays and loop order:
 *   ora_gauss_seidel_mg_fine   poisson/multigrid_fine_fine.f90:332-451  (one colour; fast path for
 *                              cells with scan flag 0, masked "solve" path otherwise, safe mode)
 *   ora_cmp_residual_mg_fine   :147-249 (f(:,1) = -(sum_nb - 6 phi)/dx^2 + f(:,2), masked cells 0)
 * Pinned by tests/test_amr_mg_oracle.py against dumps of the unmodified reference
 * (oracle/dump_patch/multigrid_fine_fine.f90 -> tests/golden/amr_mg_ref.npz): a partially refined
 * level with 1880 cells on the masked branch.
 *
 * Arrays are the reference's: cell = ncoarse + ind*ngridmax + igrid (1-based igrid, ind 0..7),
 * f = (3, ncell) variable-major, nbor = (6, ngridmax), flag2 = (ncell) [flag2(1:ncell)]. */
#include <stdint.h>

static const int III[3][2][8] = {{{1, 0, 1, 0, 1, 0, 1, 0}, {0, 2, 0, 2, 0, 2, 0, 2}},
                                 {{3, 3, 0, 0, 3, 3, 0, 0}, {0, 0, 4, 4, 0, 0, 4, 4}},
                                 {{5, 5, 5, 5, 0, 0, 0, 0}, {0, 0, 0, 0, 6, 6, 6, 6}}};
static const int JJJ[3][2][8] = {{{2, 1, 4, 3, 6, 5, 8, 7}, {2, 1, 4, 3, 6, 5, 8, 7}},
                                 {{3, 4, 1, 2, 7, 8, 5, 6}, {3, 4, 1, 2, 7, 8, 5, 6}},
                                 {{5, 6, 7, 8, 1, 2, 3, 4}, {5, 6, 7, 8, 1, 2, 3, 4}}};

/* neighbouring oct of oct g in shift direction k (1..6), 0 = the oct itself */
static int nb_oct(int g, int k, const int *son, const int *nbor, int64_t ngridmax) {
  if (k == 0) return g;
  return son[nbor[(int64_t)(k - 1) * ngridmax + g - 1] - 1];
}

void ora_gauss_seidel_mg_fine(int ilevel, int redstep, int safe, int ngrid, const int *igrid, const int *son,
                              const int *nbor, const int *flag2, int64_t ngridmax, int64_t ncoarse,
                              double *phi, const double *f) {
  const int64_t ncell = ncoarse + 8 * ngridmax;
  const double *f2 = f + ncell, *f3 = f + 2 * ncell;
  double dx2 = 1.0;
  for (int l = 0; l < ilevel; l++) dx2 *= 0.5;
  dx2 = dx2 * dx2;
  static const int ired[4] = {1, 4, 6, 7}, iblack[4] = {2, 3, 5, 8};
  for (int ind0 = 0; ind0 < 4; ind0++) {
    const int ind = (redstep ? ired[ind0] : iblack[ind0]) - 1;
    const int64_t iskip = ncoarse + (int64_t)ind * ngridmax;
    for (int i = 0; i < ngrid; i++) {
      const int g = igrid[i];
      const int64_t c = iskip + g - 1;
      double nb_sum = 0.0;
      if (flag2[c] / ngridmax == 0) {
        for (int inbor = 0; inbor < 2; inbor++)
          for (int idim = 0; idim < 3; idim++) {
            const int gn = nb_oct(g, III[idim][inbor][ind], son, nbor, ngridmax);
            nb_sum = nb_sum + phi[ncoarse + (int64_t)(JJJ[idim][inbor][ind] - 1) * ngridmax + gn - 1];
          }
        phi[c] = (nb_sum - dx2 * f2[c]) / 6.0;
      } else {
        if (f3[c] <= 0.0) continue;
        if (safe && f3[c] < 1.0) continue;
        double weight = 0.0;
        for (int inbor = 0; inbor < 2; inbor++)
          for (int idim = 0; idim < 3; idim++) {
            const int gn = nb_oct(g, III[idim][inbor][ind], son, nbor, ngridmax);
            if (gn == 0) {
              weight = weight - 1.0 / f3[c];
            } else {
              const int64_t cn = ncoarse + (int64_t)(JJJ[idim][inbor][ind] - 1) * ngridmax + gn - 1;
              if (f3[cn] <= 0.0) weight = weight + f3[cn] / f3[c];
              else nb_sum = nb_sum + phi[cn];
            }
          }
        phi[c] = (nb_sum - dx2 * f2[c]) / (6.0 - weight);
      }
    }
  }
}

void ora_cmp_residual_mg_fine(int ilevel, int ngrid, const int *igrid, const int *son, const int *nbor,
                              const int *flag2, int64_t ngridmax, int64_t ncoarse, const double *phi, double *f) {
  const int64_t ncell = ncoarse + 8 * ngridmax;
  double *f1 = f;
  const double *f2 = f + ncell, *f3 = f + 2 * ncell;
  double dx = 1.0;
  for (int l = 0; l < ilevel; l++) dx *= 0.5;
  const double oneoverdx2 = 1.0 / (dx * dx);
  for (int ind = 0; ind < 8; ind++) {
    const int64_t iskip = ncoarse + (int64_t)ind * ngridmax;
    for (int i = 0; i < ngrid; i++) {
      const int g = igrid[i];
      const int64_t c = iskip + g - 1;
      const double phi_c = phi[c];
      double nb_sum = 0.0;
      if (flag2[c] / ngridmax == 0) {
        for (int inbor = 0; inbor < 2; inbor++)
          for (int idim = 0; idim < 3; idim++) {
            const int gn = nb_oct(g, III[idim][inbor][ind], son, nbor, ngridmax);
            nb_sum = nb_sum + phi[ncoarse + (int64_t)(JJJ[idim][inbor][ind] - 1) * ngridmax + gn - 1];
          }
      } else {
        if (f3[c] <= 0.0) { f1[c] = 0.0; continue; }
        for (int idim = 0; idim < 3; idim++)
          for (int inbor = 0; inbor < 2; inbor++) {
            const int gn = nb_oct(g, III[idim][inbor][ind], son, nbor, ngridmax);
            if (gn == 0) {
              nb_sum = nb_sum - phi_c / f3[c];
            } else {
              const int64_t cn = ncoarse + (int64_t)(JJJ[idim][inbor][ind] - 1) * ngridmax + gn - 1;
              if (f3[cn] <= 0.0) nb_sum = nb_sum + phi_c * (f3[cn] / f3[c]);
              else nb_sum = nb_sum + phi[cn];
            }
          }
      }
      f1[c] = -oneoverdx2 * (nb_sum - 6.0 * phi_c) + f2[c];
    }
  }
}
