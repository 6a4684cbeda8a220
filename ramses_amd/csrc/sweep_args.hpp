// sweep_args.hpp -- kernel argument block of the Godunov sweep.
#pragma once
#include "hydro_core.hpp"

namespace ramses_amd {

// what one sweep launch covers
enum { SWEEP_ALL = 0, SWEEP_INTERIOR = 1, SWEEP_SHELL = 2 };

// a box of tiles x planes inside the brick, cut into z-chunks; `first` = index
// of its first workgroup in the launch
struct SweepBox {
  int tx0, ntx, ty0, nty, zlo, zhi, zchunk, first;
};

struct SweepArgs {
  const double *uold;
  double *unew;
  const double *grav;   // may be null
  // A fully covered level of an AMR run (hydro/godunov_fine.f90:661-666,720-747), swept IN PLACE on the reference's cell vectors:
  //   mask    1 byte per cell of the level's brick, [nz][ny][nx]: non-zero where the cell is refined (son > 0) -- the fluxes through
  //           the faces of such cells are reset to zero;
  //   cellidx the 0-based index of every brick cell in the cell vectors, [nz][ny][nx] ints: uold / grav / unew are then the cell
  //           vectors themselves (pitch_var = ncell; pitch_y, pitch_z unused), every lane addresses them through the index of its
  //           (plane, column), and the update starts from unew -- which already holds what the finer level owes to this one --
  //           and lands there.
  // Both null: the plain brick sweep (unew = uold + updates).
  const unsigned char *mask = nullptr;
  const int *cellidx = nullptr;
  int nx, ny, nz;       // interior cells
  int ng;               // ghost width (0 = periodic wrap in-kernel)
  long pitch_y, pitch_z, pitch_var;
  int zchunk;           // planes marched per workgroup
  int region;           // SWEEP_* (in); the launcher fills the boxes
  int nbox, nblocks;
  SweepBox box[6];
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
