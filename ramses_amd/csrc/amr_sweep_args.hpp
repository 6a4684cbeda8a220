// amr_sweep_args.hpp -- argument block of the AMR godunov_fine kernels.
#pragma once
#include "hydro_core.hpp"

namespace ramses_amd {

struct AmrSweepArgs {
  const double *uold;   // [nvar][ncell]   (uold(1:ncell,1:nvar), column major)
  double *unew;         // [nvar][ncell]
  const double *grav;   // f(1:ncell,1:3) or null
  double *divu, *enew;  // pressure_fix: velocity divergence and internal energy (1:ncell), or null
  const int *son;       // [ncell]
  const int *nbor;      // [6][ngridmax]   (nbor(1:ngridmax,1:twondim))
  const int *father;    // [ngridmax]
  const int *igrid;     // active(ilevel)%igrid, 1-based oct indices
  int ngrid;
  int nvar;             // 5 + passive scalars (<= 7)
  int scheme;           // 0 muscl (trace3d), 1 plmde (tracexyz)
  long ncell, ncoarse, ngridmax;
  double dt, dx, rdx;
  double difmag;        // artificial diffusion coefficient (cmpdivu + consup), 0: off
  int pow2;
  int interpol_var, interpol_type;
  double *corr;         // [ngrid][6][4][nvar+2] fluxes (+ the two pressure_fix quantities) owed to coarse neighbour cells
  int *corr_tgt;        // [ngrid][6] the coarse cell (1-based) or 0
  int *err;             // tree inconsistencies found
  // grouped kernel: the level's octs repacked as contiguous records (see amr_pack_kernel), or null
  const double *packed;
  int rec;              // doubles per record
  HydroConst P;
};

hipError_t launch_amr_godunov(const AmrSweepArgs &A, int slope_type, int riemann, int *posof, int nvector,
                              hipStream_t s, double *pack_area = nullptr, int *walk_area = nullptr);
hipError_t launch_amr_coarse_update(const AmrSweepArgs &A, const int *posof, int nvector, hipStream_t s);
// doubles per packed oct record for nvar variables: 8 primitive values per variable, 8 refinement flags, padded to 128 bytes
inline int amr_pack_rec(int nvar) { return ((8 * nvar + 4) + 15) / 16 * 16; }
constexpr int AMR_PACK_REC_MAX = 64;   // nvar = 7

}  // namespace ramses_amd
