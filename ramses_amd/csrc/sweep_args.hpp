// sweep_args.hpp -- kernel argument block of the Godunov sweep.
#pragma once
#include "hydro_core.hpp"

namespace ramses_amd {

struct SweepArgs {
  const double *uold;
  double *unew;
  const double *grav;   // may be null
  int nx, ny, nz;       // interior cells
  int ng;               // ghost width (0 = periodic wrap in-kernel)
  long pitch_y, pitch_z, pitch_var;
  int zchunk;           // planes marched per workgroup
  int ntx, nty, ntz;    // tiles per direction (filled by the launcher)
  double dt, dx, rdx;   // rdx = 1/dx (exact when dx is a power of two)
  int pow2;             // dx is a power of two: (f*dt)/dx == (f*dt)*rdx bit for bit
  HydroConst P;
};

namespace strictmode {
hipError_t launch_godunov_sweep(SweepArgs &A, int slope_type, int riemann, int by, int scheme, int nvar,
                                bool grav, hipStream_t s);
}
namespace fastmode {
hipError_t launch_godunov_sweep(SweepArgs &A, int slope_type, int riemann, int by, int scheme, int nvar,
                                bool grav, hipStream_t s);
}

}  // namespace ramses_amd
