// pack_args.hpp -- argument block of the octree <-> brick copy kernels.
#pragma once
#include <hip/hip_runtime.h>

namespace ramses_amd {

struct PackArgs {
  double *brick;          // [nvar][n][n][n]
  double *cellvec;        // RAMSES cell vectors, (1:ncell,1:nvar) column-major
  const int *igrid;       // active(ilevel)%igrid(1:ngrid), 1-based oct slots
  const long *octorg;     // brick index of each oct's (0,0,0) cell
  int ngrid, n, nvar;
  long ncoarse, ngridmax, ncell, pitch_var;
  long pitch_y = 0, pitch_z = 0;   // 0: dense n^3 brick (pitch_y = n, pitch_z = n*n)
};

hipError_t launch_oct_origin(const int *igrid, const double *xg, long ngridmax, int ngrid, int n,
                             const double skip[3], long *octorg, int *bad, hipStream_t s);
hipError_t launch_oct_copy(const PackArgs &A, bool gather, hipStream_t s);
hipError_t launch_oct_leaf(const PackArgs &A, const int *son, int *leaf, hipStream_t s);

}  // namespace ramses_amd
