/*
 * mg_oracle.c -- TEST INFRASTRUCTURE ONLY (see hydro_oracle.h for the rules).
 *
 * CPU restatement of the reference's fine-level multigrid Poisson solver
 * restricted to a fully refined periodic level (every mask = 1, every scan
 * flag = 0, levelmin_mg = 1), in dense brick order phi[k][j][i]:
 *
 *   multigrid_fine, recursive_multigrid_coarse  poisson/multigrid_fine_commons.f90:25-390
 *   gauss_seidel_mg_fine/_coarse                poisson/multigrid_fine_fine.f90:332-451,
 *                                               poisson/multigrid_fine_coarse.f90:411-560
 *   cmp_residual_mg_fine/_coarse                multigrid_fine_fine.f90:147-249, _coarse.f90:167-300
 *   restrict_residual_*_reverse                 multigrid_fine_fine.f90:528-590, _coarse.f90:692-760
 *   interpolate_and_correct_*                   multigrid_fine_fine.f90:596-698, _coarse.f90:769-870
 *   make_fine_bc_rhs (unmasked branch)          multigrid_fine_commons.f90:1058-1159
 *   gradient_phi                                poisson/force_fine.f90:199-324
 *
 * Floating-point order follows the reference (neighbour sum: inbor outer,
 * idim inner; restriction: octant order; prolongation: weights a,b,b,c,b,c,c,d).
 * Parity status: PINNED against end-to-end runs of the reference program
 * (tests/golden/poisson_ref_runs.npz, tests/test_mg_oracle.py).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define IDX(i, j, k, n) ((size_t)(i) + (size_t)(n) * ((size_t)(j) + (size_t)(n) * (size_t)(k)))

static inline int wrap(int i, int n) { return i < 0 ? i + n : (i >= n ? i - n : i); }

/* neighbour sum in the reference's order: x-, y-, z-, x+, y+, z+ */
static inline double nb_sum6(const double *phi, int i, int j, int k, int n) {
  double s = 0.0;
  s = s + phi[IDX(wrap(i - 1, n), j, k, n)];
  s = s + phi[IDX(i, wrap(j - 1, n), k, n)];
  s = s + phi[IDX(i, j, wrap(k - 1, n), n)];
  s = s + phi[IDX(wrap(i + 1, n), j, k, n)];
  s = s + phi[IDX(i, wrap(j + 1, n), k, n)];
  s = s + phi[IDX(i, j, wrap(k + 1, n), n)];
  return s;
}

/* one colour of red-black Gauss-Seidel; red = octants 1,4,6,7 = even parity */
void ora_mg_gauss_seidel(double *phi, const double *rhs, int n, double dx2, int redstep) {
  const int par = redstep ? 0 : 1;
  for (int k = 0; k < n; k++)
    for (int j = 0; j < n; j++)
      for (int i = 0; i < n; i++) {
        if (((i + j + k) & 1) != par) continue;
        double nb = nb_sum6(phi, i, j, k, n);
        phi[IDX(i, j, k, n)] = (nb - dx2 * rhs[IDX(i, j, k, n)]) / 6.0;
      }
}

/* res = -(nb - 6 phi)/dx^2 + rhs   ("minus the residual" of the reference) */
void ora_mg_residual(const double *phi, const double *rhs, double *res, int n, double dx) {
  const double oneoverdx2 = 1.0 / (dx * dx);
  for (int k = 0; k < n; k++)
    for (int j = 0; j < n; j++)
      for (int i = 0; i < n; i++) {
        double phi_c = phi[IDX(i, j, k, n)];
        double nb = nb_sum6(phi, i, j, k, n);
        res[IDX(i, j, k, n)] = -oneoverdx2 * (nb - 6.0 * phi_c) + rhs[IDX(i, j, k, n)];
      }
}

/* dx^3 * sum res^2 in brick order (the reference sums octant-major over its
 * linked-list oct order; the norm only gates the iteration count) */
double ora_mg_norm2(const double *res, int n, double dx) {
  double s = 0.0;
  const size_t N = (size_t)n * n * n;
  for (size_t c = 0; c < N; c++) s = s + res[c] * res[c];
  return (dx * dx * dx) * s;
}

/* coarse rhs = sum over the 8 children (octant order) of res/8 */
void ora_mg_restrict(const double *res_f, double *rhs_c, int nf) {
  const int nc = nf / 2;
  for (int K = 0; K < nc; K++)
    for (int J = 0; J < nc; J++)
      for (int I = 0; I < nc; I++) {
        double acc = 0.0;
        for (int ind = 0; ind < 8; ind++) {
          int ix = ind & 1, iy = (ind >> 1) & 1, iz = (ind >> 2) & 1;
          acc = acc + res_f[IDX(2 * I + ix, 2 * J + iy, 2 * K + iz, nf)] / 8.0;
        }
        rhs_c[IDX(I, J, K, nc)] = acc;
      }
}

/* phi_f += sum_{8 of 27 parents} w * corr_c, weights (a,b,b,c,b,c,c,d)/4^3 */
void ora_mg_interp_correct(double *phi_f, const double *corr_c, int nf) {
  const int nc = nf / 2;
  const double a = 1.0 / 64.0, b = 3 * a, c = 9 * a, d = 27 * a;
  const double bbb[8] = {a, b, b, c, b, c, c, d};
  for (int k = 0; k < nf; k++)
    for (int j = 0; j < nf; j++)
      for (int i = 0; i < nf; i++) {
        const int I = i >> 1, J = j >> 1, K = k >> 1;
        const int sx = (i & 1) ? 1 : -1, sy = (j & 1) ? 1 : -1, sz = (k & 1) ? 1 : -1;
        double corr = 0.0;
        for (int t = 0; t < 8; t++) {
          int pi = (t & 1) ? I : wrap(I + sx, nc);
          int pj = (t & 2) ? J : wrap(J + sy, nc);
          int pk = (t & 4) ? K : wrap(K + sz, nc);
          corr = corr + bbb[t] * corr_c[IDX(pi, pj, pk, nc)];
        }
        phi_f[IDX(i, j, k, nf)] = phi_f[IDX(i, j, k, nf)] + corr;
      }
}

typedef struct {
  int nlev;          /* fine level index L: n = 2^L */
  double **u1, **u2, **u3; /* per MG level 1..L-1 : correction, rhs, residual */
} mg_hier;

static void coarse_cycle(mg_hier *h, int l, int safe, int ngs_coarse, int ncycles_safe) {
  const int n = 1 << l;
  const double dx = pow(0.5, l), dx2 = dx * dx;
  double *u1 = h->u1[l], *u2 = h->u2[l], *u3 = h->u3[l];
  if (l <= 1) { /* levelmin_mg = 1: "direct" solve by 2*ngs_coarse sweeps */
    for (int s = 0; s < 2 * ngs_coarse; s++) {
      ora_mg_gauss_seidel(u1, u2, n, dx2, 1);
      ora_mg_gauss_seidel(u1, u2, n, dx2, 0);
    }
    return;
  }
  const int ncycle = safe ? ncycles_safe : 1;
  for (int cyc = 0; cyc < ncycle; cyc++) {
    for (int s = 0; s < ngs_coarse; s++) {
      ora_mg_gauss_seidel(u1, u2, n, dx2, 1);
      ora_mg_gauss_seidel(u1, u2, n, dx2, 0);
    }
    ora_mg_residual(u1, u2, u3, n, dx);
    ora_mg_restrict(u3, h->u2[l - 1], n);
    memset(h->u1[l - 1], 0, sizeof(double) * (size_t)(n / 2) * (n / 2) * (n / 2));
    coarse_cycle(h, l - 1, safe, ngs_coarse, ncycles_safe);
    ora_mg_interp_correct(u1, h->u1[l - 1], n);
    for (int s = 0; s < ngs_coarse; s++) {
      ora_mg_gauss_seidel(u1, u2, n, dx2, 1);
      ora_mg_gauss_seidel(u1, u2, n, dx2, 0);
    }
  }
}

/* multigrid_fine on a fully refined periodic level `level` (n = 2^level).
 * rho: density brick; rho_tot: box mean; fourpi = 2*twopi*scale as the
 * reference computes it (twopi = 6.2831853d0, scale = boxlen/nx_loc).
 * phi: in = first guess (0 at levelmin: make_multipole_phi), out = solution.
 * f2 (n^3) and f1 (n^3) are work arrays returned for inspection (BC-modified
 * RHS and minus-residual).  Returns the number of V-cycles; *err_out = last
 * error; *safe_mode is read and updated like safe_mode(ilevel). */
int ora_mg_solve_uniform(const double *rho, double rho_tot, int level, double fourpi,
                         double epsilon, int *safe_mode, double *phi, double *f1, double *f2,
                         double *err_out) {
  const int MAXITER = 10, ngs_fine = 2, ngs_coarse = 2, ncycles_coarse_safe = 1;
  const double SAFE_FACTOR = 0.5;
  const int n = 1 << level;
  const size_t N = (size_t)n * n * n;
  const double dx = pow(0.5, level), dx2 = dx * dx;
  mg_hier h;
  h.nlev = level;
  h.u1 = (double **)calloc(level + 1, sizeof(double *));
  h.u2 = (double **)calloc(level + 1, sizeof(double *));
  h.u3 = (double **)calloc(level + 1, sizeof(double *));
  for (int l = 1; l < level; l++) {
    size_t nl = (size_t)1 << (3 * l);
    h.u1[l] = (double *)calloc(nl, sizeof(double));
    h.u2[l] = (double *)calloc(nl, sizeof(double));
    h.u3[l] = (double *)calloc(nl, sizeof(double));
  }
  for (size_t c = 0; c < N; c++) f2[c] = fourpi * (rho[c] - rho_tot);

  int iter = 0;
  double err = 1.0, last_err, i_res_norm2 = 0.0, res_norm2;
  for (;;) {
    iter++;
    for (int s = 0; s < ngs_fine; s++) {
      ora_mg_gauss_seidel(phi, f2, n, dx2, 1);
      ora_mg_gauss_seidel(phi, f2, n, dx2, 0);
    }
    ora_mg_residual(phi, f2, f1, n, dx);
    if (iter == 1) i_res_norm2 = ora_mg_norm2(f1, n, dx);
    if (level > 1) {
      ora_mg_restrict(f1, h.u2[level - 1], n);
      memset(h.u1[level - 1], 0, sizeof(double) * (N / 8));
      coarse_cycle(&h, level - 1, *safe_mode, ngs_coarse, ncycles_coarse_safe);
      ora_mg_interp_correct(phi, h.u1[level - 1], n);
    }
    for (int s = 0; s < ngs_fine; s++) {
      ora_mg_gauss_seidel(phi, f2, n, dx2, 1);
      ora_mg_gauss_seidel(phi, f2, n, dx2, 0);
    }
    ora_mg_residual(phi, f2, f1, n, dx);
    res_norm2 = ora_mg_norm2(f1, n, dx);
    last_err = err;
    err = sqrt(res_norm2 / (i_res_norm2 + 1e-20 * (rho_tot * rho_tot)));
    if (err < epsilon || iter >= MAXITER) break;
    if (err > last_err * SAFE_FACTOR && !*safe_mode) *safe_mode = 1;
  }
  for (int l = 1; l < level; l++) { free(h.u1[l]); free(h.u2[l]); free(h.u3[l]); }
  free(h.u1); free(h.u2); free(h.u3);
  if (err_out) *err_out = err;
  return iter;
}

/* gradient_phi on a fully refined periodic level: f[d] = a(phi(-1)-phi(+1)) - b(phi(-2)-phi(+2)) */
void ora_gradient_phi_uniform(const double *phi, int level, double *f) {
  const int n = 1 << level;
  const size_t N = (size_t)n * n * n;
  const double dx = pow(0.5, level);
  const double a = 0.50 * 4.0 / 3.0 / dx;
  const double b = 0.25 * 1.0 / 3.0 / dx;
  for (int k = 0; k < n; k++)
    for (int j = 0; j < n; j++)
      for (int i = 0; i < n; i++) {
        for (int d = 0; d < 3; d++) {
          int di = d == 0, dj = d == 1, dk = d == 2;
          double p1 = phi[IDX(wrap(i - di, n), wrap(j - dj, n), wrap(k - dk, n), n)];
          double p2 = phi[IDX(wrap(i + di, n), wrap(j + dj, n), wrap(k + dk, n), n)];
          double p3 = phi[IDX(wrap(i - 2 * di, n), wrap(j - 2 * dj, n), wrap(k - 2 * dk, n), n)];
          double p4 = phi[IDX(wrap(i + 2 * di, n), wrap(j + 2 * dj, n), wrap(k + 2 * dk, n), n)];
          f[IDX(i, j, k, n) + N * d] = a * (p1 - p2) - b * (p3 - p4);
        }
      }
}
