/* oracle/gravity_oracle.c -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * CPU restatement of the gravity source terms the hydro state receives in amr_step's gravity branch,
 * on a dense level brick u[nvar][N] (rho, rho*u, rho*v, rho*w, E, ...) with the acceleration f[3][N]:
 *   ora_synchro_hydro        synchro_hydro_fine / synchydrofine1, which_force = 1
 *                            (hydro/synchro_hydro_fine.f90:5-136)
 *   ora_add_gravity_source   add_gravity_source_terms (hydro/godunov_fine.f90:237-289)
 * Pinned by tests/test_gravity_step_oracle.py: together with the sweep, Courant and multigrid
 * oracles they reproduce end-to-end runs of the reference program with self-gravity
 * (tests/golden/poisson_ref_runs.npz: state after one and after three coarse steps) bit for bit. */
#include <stdint.h>

static double dmax(double a, double b) { return a > b ? a : b; }

void ora_synchro_hydro(double *u, const double *f, int64_t N, double dteff, double smallr) {
  for (int64_t c = 0; c < N; c++) {
    const double d = dmax(u[c], smallr);
    /* remove the kinetic energy from the total energy (:65-73) */
    double pp = u[4 * N + c];
    for (int k = 0; k < 3; k++) pp = pp - 0.5 * (u[(k + 1) * N + c] * u[(k + 1) * N + c]) / d;
    u[4 * N + c] = pp;
    /* kick (:76-100): momentum + max(rho,smallr)*f*dteff */
    for (int k = 0; k < 3; k++) u[(k + 1) * N + c] = u[(k + 1) * N + c] + d * f[k * N + c] * dteff;
    /* put the kinetic energy of the new momenta back (:103-113) */
    pp = u[4 * N + c];
    for (int k = 0; k < 3; k++) pp = pp + 0.5 * (u[(k + 1) * N + c] * u[(k + 1) * N + c]) / d;
    u[4 * N + c] = pp;
  }
}

void ora_add_gravity_source(double *unew, const double *uold, const double *f, int64_t N, double dt, double smallr) {
  for (int64_t c = 0; c < N; c++) {
    const double d = dmax(unew[c], smallr);
    double u = unew[N + c] / d, v = unew[2 * N + c] / d, w = unew[3 * N + c] / d;
    double e_kin = 0.5 * d * (u * u + v * v + w * w);
    const double e_prim = unew[4 * N + c] - e_kin;
    const double d_old = dmax(uold[c], smallr);
    const double req = 0.0;                       /* strict_equilibrium = 0 */
    const double fact = (d_old - req) / d * 0.5 * dt;
    u = u + f[c] * fact;
    unew[N + c] = d * u;
    v = v + f[N + c] * fact;
    unew[2 * N + c] = d * v;
    w = w + f[2 * N + c] * fact;
    unew[3 * N + c] = d * w;
    e_kin = 0.5 * d * (u * u + v * v + w * w);
    unew[4 * N + c] = e_prim + e_kin;
  }
}
