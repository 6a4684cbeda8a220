/*
 * hydro_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C, no FMA contraction) of the RAMSES Godunov hydro
 * path.  Every function cites the reference file:line it restates.  Only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use
 * anything under oracle/ -- and only as the checker.  The product path
 * (ramses_amd/) never links, imports or calls this code.
 *
 * Parity status: PINNED.  tests/test_oracle_vs_ref.py checks these functions
 * bit-for-bit against the reference's own unsplit()/riemann_*() compiled from
 * /root/reference by oracle/build_ref.sh (-> oracle/_ref/), and against the
 * golden vectors under tests/golden/ generated from that build.
 */
#ifndef RAMSES_HYDRO_ORACLE_H
#define RAMSES_HYDRO_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

enum { ORA_RIEMANN_LLF = 0, ORA_RIEMANN_HLLC = 1, ORA_RIEMANN_HLL = 2,
       ORA_RIEMANN_ACOUSTIC = 3, ORA_RIEMANN_EXACT = 4 };
enum { ORA_SCHEME_MUSCL = 0, ORA_SCHEME_PLMDE = 1 };

/* Mirrors the solver knobs of hydro/hydro_parameters.f90:75-89 */
typedef struct {
  int ndim;            /* compile-time NDIM of the reference build            */
  int nvar;            /* ndim+2 (+ passive scalars); NENER=0                 */
  double gamma;
  double smallr;
  double smallc;
  int slope_type;
  double slope_theta;
  int riemann;         /* ORA_RIEMANN_*                                       */
  int scheme;          /* ORA_SCHEME_*                                        */
  int niter_riemann;
  double difmag;
} ora_hydro_params;

/* hydro/umuscl.f90:22-171.  Arrays are in the reference's Fortran layout with
 * leading dimension nvector:
 *   uin   (nvector, 6^ndim cells [-1:4 per active dim], nvar)
 *   gravin(nvector, 6^ndim, ndim)
 *   flux  (nvector, 3^ndim faces [1:3 per active dim], nvar, ndim)
 *   tmp   (nvector, 3^ndim, 2, ndim)
 */
void ora_unsplit(const ora_hydro_params *p, const double *uin,
                 const double *gravin, double *flux, double *tmp, double dx,
                 double dy, double dz, double dt, int ngrid, int nvector);

/* The Riemann solvers of hydro/godunov_utils.f90 on (nvector,nvar) arrays;
 * fgdnv is (nvector,nvar+1). */
void ora_riemann(const ora_hydro_params *p, const double *qleft,
                 const double *qright, double *fgdnv, int ngrid, int nvector);

/* hydro/godunov_fine.f90:486-911 restricted to a fully refined periodic
 * level (every neighbour oct exists, ok(:)=.false.): for every oct gather the
 * 6^ndim stencil from uold, call unsplit, update the oct's 2^ndim cells of
 * unew.  Arrays are dense bricks u[ivar][k][j][i] (i fastest), n cells per
 * active dimension, periodic.  grav may be NULL (poisson=.false.).
 * unew must be pre-set by the caller (set_unew: unew=uold). */
void ora_godunov_uniform(const ora_hydro_params *p, const double *uold,
                         const double *grav, double *unew, int nx, int ny,
                         int nz, double dx, double dt);

/* hydro/godunov_utils.f90:5-120 (cmpdt) + hydro/courant_fine.f90:1-159
 * restricted to a uniform periodic level: returns the CFL time step
 * courant_factor*dx/max(...) minimised over the brick. grav may be NULL. */
double ora_courant_uniform(const ora_hydro_params *p, const double *uold,
                           const double *grav, int nx, int ny, int nz,
                           double dx, double courant_factor);

#ifdef __cplusplus
}
#endif
#endif
