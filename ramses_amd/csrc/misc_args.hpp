// misc_args.hpp -- argument blocks of the streaming kernels.
#pragma once
#include "hydro_core.hpp"

namespace ramses_amd {

struct CourantArgs {
  const double *uold;
  const double *grav;
  double *out;          // {dt, mass, etot, eint}
  int nx, ny, nz, ng;
  long pitch_y, pitch_z, pitch_var;
  double dx, vol, courant_factor, dt_init;
  double ndimf;         // NDIM of the problem (1, 2: embedded in the brick with ny and/or nz = 1)
  HydroConst P;
};

// copy an (ex,ey,ez) box of nvar variables between two strided layouts
struct BoxCopyArgs {
  const double *src;
  double *dst;
  int ex, ey, ez, nvar;
  long s_off, s_py, s_pz, s_pv;
  long d_off, d_py, d_pz, d_pv;
};

hipError_t launch_courant_init(double *out, double dt_init, hipStream_t s);
hipError_t launch_courant(const CourantArgs &A, bool grav, hipStream_t s);
hipError_t launch_box_copy(const BoxCopyArgs &A, hipStream_t s);

// all 26 neighbour regions of a ghost-layer brick in ONE launch: region r is the box
// (org,ext) of the brick, packed contiguously ([var][k][j][i]) at buf + off[r]
struct MultiBoxArgs {
  double *brick;
  double *buf;
  int nbox, nvar, pack;     // pack: brick -> buf, else buf -> brick
  long pitch_y, pitch_z, pitch_var;
  int org[26][3], ext[26][3];
  long off[26];
  long rows_before[27];     // prefix sum of ext_y*ext_z*nvar per box
};
hipError_t launch_multi_box(const MultiBoxArgs &A, hipStream_t s);

// physical boundary of one face of a ghost-layer brick (make_boundary_hydro)
struct BoundaryArgs {
  double *u;
  int nx, ny, nz, ng, nvar;
  long pitch_y, pitch_z, pitch_var;
  int face;             // 0:-x 1:+x 2:-y 3:+y 4:-z 5:+z
  int type;             // 1 reflexive, 2 outflow (zero gradient), 3 imposed
  int no_inflow;
  double smallr;
  double value[8];      // imposed conserved state
};
hipError_t launch_boundary(const BoundaryArgs &A, hipStream_t s);

// gravity source terms on a dense level brick u[nvar][N] with the acceleration f[3][N]
// synchydrofine1 (hydro/synchro_hydro_fine.f90:45-136, which_force = 1): momentum kick by dteff,
// total energy carried along through the internal energy
hipError_t launch_synchro_hydro(double *u, const double *f, long N, double dteff, double smallr, hipStream_t s);
// add_gravity_source_terms (hydro/godunov_fine.f90:237-289): half a time step on unew, the density
// ratio taken against uold
hipError_t launch_add_gravity_source(double *unew, const double *uold, const double *f, long N, double dt, double smallr,
                                     hipStream_t s);

// force_fine's diagnostics (poisson/force_fine.f90:158-190) on a level brick: out = {sum fact*f^2 over leaf
// cells, max |rho|}; partial = FORCE_DIAG_SCRATCH doubles; leaf[N] (0/1) or nullptr = every cell is a leaf
constexpr int FORCE_DIAG_SCRATCH = 1024;
hipError_t launch_force_diag(const double *f, const double *rho, const int *leaf, long N, double fact, double *partial,
                             double *out, hipStream_t s);

}  // namespace ramses_amd
