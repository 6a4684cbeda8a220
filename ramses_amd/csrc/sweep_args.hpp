// sweep_args.hpp -- kernel argument block of the Godunov sweep.
#pragma once
#include "hydro_core.hpp"

namespace ramses_amd {

// what one sweep launch covers
enum { SWEEP_ALL = 0, SWEEP_INTERIOR = 1, SWEEP_SHELL = 2 };

// status bits of a device cell (SweepArgs::stat)
enum { CELL_REFINED = 1, CELL_OWNED = 2, CELL_GHOST = 4 };
// storage tiles of a level: 32 x 4 x 4 octs
constexpr int TILE_OX = 32, TILE_OY = 4, TILE_OZ = 4, TILE_OCTS = TILE_OX * TILE_OY * TILE_OZ;

// a box of tiles x planes inside the brick, cut into z-chunks; `first` = index
// of its first workgroup in the launch
struct SweepBox {
  int tx0, ntx, ty0, nty, zlo, zhi, zchunk, first;
};

struct SweepArgs {
  const double *uold;
  double *unew;
  const double *grav;   // may be null
  // A level of a resident AMR run, swept IN PLACE on the device's cell vectors (csrc/amr_layout.hpp: the octs of a level are
  // numbered on the device so that they fill TILES of 32 x 4 x 4 octs = 64 x 8 x 8 cells; inside a tile the cell vectors of
  // one octant position are a dense little brick, 256-byte runs along x).  stat != null selects this mode:
  //   stat    one byte per device cell, indexed like a cell vector: bit 0 the cell is refined (son > 0: the fluxes through its
  //           faces are reset, hydro/godunov_fine.f90:661-666,720-747), bit 1 the cell belongs to an oct of the call's list (it
  //           is updated), bit 2 the cell belongs to a GHOST oct (a missing neighbour oct, interpolated from the coarser level
  //           by a pre-pass; the marching kernel treats it like any other cell it does not update);
  //   dir     the level's tile directory [ntz][nty][ntx]: 0-based index of the tile's first cell in a cell vector
  //           (= ncoarse + first oct - 1), or -1 where no oct of the level and no ghost oct falls into the tile;
  //   work    the launch's work items, one per workgroup: (x0, y0, z0, z1) = first interior column / row of the 60 x (BY-4)
  //           tile and the planes [z0, z1) it marches;
  // (the fluxes owed to the coarser level are the surface pass's business: SurfArgs below)
  // uold / grav / unew are then the cell vectors themselves (pitch_var = ncell of the device; pitch_y, pitch_z unused), the
  // update starts from unew -- which already holds what the finer level owes to this one (:752-790) -- and lands there.
  const unsigned char *stat = nullptr;
  const int *dir = nullptr;
  const int *work = nullptr;          // int4 per workgroup
  int ntx = 0, nty = 0, ntz = 0, nwork = 0;
  int base_uold = 0;                  // the level has no finer octs: unew == uold on entry (set_unew), the update may start from uold
  long ngd = 0, ncoarse = 0;
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

// The surface pass of the sweep of a level in tiles (hydro/godunov_fine.f90:798-908): the fluxes an updated cell exchanges with
// a GHOST cell (an oct the level does not have, interpolated by the pre-pass) are owed to the leaf cell of the coarser level
// behind that oct face.  One thread per (event = (oct of the list, face) with such a neighbour, fine face q): it rebuilds the
// traced states of the two cells from their own seven-cell stencils with the functions of the marching kernel -- the same
// operations on the same values, hence the same flux -- and files it in rec[event][q][nvar+2].  The marching loop itself knows
// nothing about the surfaces of the level.
struct SurfArgs {
  const double *uold = nullptr, *grav = nullptr;     // the device's cell vectors
  const unsigned char *stat = nullptr;
  const int *dir = nullptr, *tileid = nullptr;       // the level's tile directory / the tile of every 512-oct slab of its index range
  const int *events = nullptr;                       // list position * 6 + face, sorted by (face, device oct)
  const int *ig = nullptr;                           // the call's list, device octs (1-based)
  double *rec = nullptr;                             // [nevent][4][nvar + 2]
  int nevent = 0;
  long base = 0, ncoarse = 0, ngd = 0, ncell = 0;
  int no = 0, ntx = 0, nty = 0, ntz = 0;
  double dt = 0, dx = 0, rdx = 0;
  int pow2 = 0;
  int qminor = 0;               // thread t -> (event, fine face): 0: (t % nevent, t / nevent), 1: (t / 4, t % 4)
  HydroConst P;
};

namespace strictmode {
hipError_t launch_surface_flux(const SurfArgs &A, int slope_type, int riemann, int nvar, int scheme, bool grav, hipStream_t s);
hipError_t launch_godunov_sweep(SweepArgs &A, int slope_type, int riemann, int by, int scheme, int nvar,
                                bool grav, hipStream_t s);
int tile_sweep_rows(int riemann, int nvar, int slope_type, int scheme);
}
namespace fastmode {
hipError_t launch_surface_flux(const SurfArgs &A, int slope_type, int riemann, int nvar, int scheme, bool grav, hipStream_t s);
hipError_t launch_godunov_sweep(SweepArgs &A, int slope_type, int riemann, int by, int scheme, int nvar,
                                bool grav, hipStream_t s);
int tile_sweep_rows(int riemann, int nvar, int slope_type, int scheme);
}

}  // namespace ramses_amd
