// rho_args.hpp -- argument block of rho_fine's hydro deposit on a level brick (rho_fine.hip)
#pragma once
#include <hip/hip_runtime.h>

namespace ramses_amd {

struct RhoArgs {
  const double *dens;     // uold(:,1) of the level as a dense brick [n][n][n]
  double *rho;            // out: the deposit, same layout
  const long *octorg;     // [ngrid] brick index of each oct's (0,0,0) cell, in list order
  const int *octidx;      // [(n/2)^3] list index of the oct at each oct position
  int n, ngrid, nvector;
  double dx;              // cell size in units of the coarse box (0.5^level)
  double scale;           // boxlen/nx_loc
  double vol_loc;         // (dx*scale)^3
  double smallr;
};

hipError_t launch_oct_index(const long *octorg, int ngrid, int n, int *octidx, hipStream_t s);
hipError_t launch_rho_deposit(const RhoArgs &A, hipStream_t s);
// the four strictly sequential sums multipole(1:4), bit for bit, in parallel; scratch: multipole_scratch_bytes(8*ngrid) bytes;
// the last 4 ints of the scratch area count the segments that took the slow path (tests, tuning)
size_t multipole_scratch_bytes(long ncells);
hipError_t launch_multipole(const RhoArgs &A, double *out4, void *scratch, hipStream_t s);
// the same sums over the multipoles mp(4, ncell) of the cells of an AMR level (cic_from_multipole at levelmin, pm/rho_fine.f90:931-938)
hipError_t launch_multipole_vec(const double *mp, const int *igrid, int ngrid, int nvector, long ncell, long ncoarse, long ngridmax,
                                double *out4, void *scratch, hipStream_t s);

// rho_fine's hydro deposit on one level of an AMR run (multipole_fine + cic_from_multipole on the cell vectors and the tree);
// levels must be visited from the finest down (a split cell sums its children's multipoles).  posof: ngridmax ints, all -1.
hipError_t launch_amr_rho_level(const double *dens, double *mp, double *rho, const double *xg, const int *son, const int *nbor,
                                const int *father, const int *igrid, int *posof, int ngrid, int nvector, long ncoarse, long ngridmax,
                                int ilevel, double boxlen_over_nx, double smallr, hipStream_t s);

// the same with several ranks (own octs deposit into own and reception cells; the exchanges between the two steps and after the
// second belong to the caller): step 1 multipole_fine on the own octs, step 2 the gather over own + reception target cells with the
// own octs found by position (hkeys / hvals: hcap slots, a power of two >= 2 n_own)
hipError_t launch_amr_multipole_level(const double *dens, double *mp, const double *xg, const int *son, const int *igrid, int n_own,
                                      long ncoarse, long ngridmax, int ilevel, double boxlen_over_nx, double smallr, hipStream_t s);
hipError_t launch_amr_deposit_level(double *mp, double *rho, const double *xg, const int *igrid, int n_own, int n_all, int nvector,
                                    long ncoarse, long ngridmax, int ilevel, double boxlen_over_nx, unsigned long long *hkeys, int *hvals,
                                    unsigned hcap, hipStream_t s);

}  // namespace ramses_amd
