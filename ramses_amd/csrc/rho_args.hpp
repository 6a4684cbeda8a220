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
hipError_t launch_multipole(const RhoArgs &A, double *out4, hipStream_t s);

}  // namespace ramses_amd
