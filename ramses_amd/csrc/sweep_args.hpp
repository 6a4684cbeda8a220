// sweep_args.hpp -- kernel argument block of the Godunov sweep.
#pragma once
#include "hydro_core.hpp"

namespace ramses_amd {

// sub-box selectors of one sweep launch
enum { SWEEP_ALL = 0, SWEEP_INTERIOR = 1, SWEEP_SHELL_ZLO = 2, SWEEP_SHELL_ZHI = 3, SWEEP_SHELL_YLO = 4,
       SWEEP_SHELL_YHI = 5, SWEEP_SHELL_XLO = 6, SWEEP_SHELL_XHI = 7 };

struct SweepArgs {
  const double *uold;
  double *unew;
  const double *grav;   // may be null
  int nx, ny, nz;       // interior cells
  int ng;               // ghost width (0 = periodic wrap in-kernel)
  long pitch_y, pitch_z, pitch_var;
  int zchunk;           // planes marched per workgroup
  int ntx, nty, ntz;    // tiles per direction of this launch (filled by the launcher)
  int tx0, ty0, zlo, zhi;  // first tile / plane range of this launch
  int region;           // SWEEP_*
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
