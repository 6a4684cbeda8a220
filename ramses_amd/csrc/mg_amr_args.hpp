// mg_amr_args.hpp -- argument blocks of the AMR multigrid kernels.
#pragma once
#include <hip/hip_runtime.h>

namespace ramses_amd {

// one level of a solve in the reference's multigrid layout (cell (ind,i) at ind*ngrid+i)
struct MgAmrLevel {
  int ngrid;             // octs in the layout: the rank's own first, then (MPI) the reception octs of the other ranks
  int nact;              // the first nact octs are updated (the rank's own); the rest is read only
  const int *igrid;      // AMR index (1-based) of the i-th oct
  double *u1, *u2, *u3, *u4;   // phi/correction, rhs, residual, mask
  const int *scan;       // per cell: 0 = inner cell (fast path), 1 = perform scan
};

struct MgAmrTree {
  const int *son;        // [ncell]
  const int *nbor;       // [6][ngridmax]
  const int *father;     // [ngridmax]
  const int *lookup;     // [ngridmax] oct -> 1-based position in its level's list, <=0: not in the solve
  long ncoarse, ngridmax;
};

hipError_t mgamr_launch_gs(const MgAmrLevel &L, const MgAmrTree &T, int color, int safe, double dx2, hipStream_t s);
hipError_t mgamr_launch_residual(const MgAmrLevel &L, const MgAmrTree &T, double oneoverdx2, hipStream_t s);
hipError_t mgamr_launch_norm(const MgAmrLevel &L, double scale, double *partial, double *out, hipStream_t s);
hipError_t mgamr_launch_restrict(const MgAmrLevel &F, const MgAmrLevel &C, const MgAmrTree &T, hipStream_t s);
hipError_t mgamr_launch_interp(const MgAmrLevel &F, const MgAmrLevel &C, const MgAmrTree &T, hipStream_t s);
hipError_t mgamr_launch_gather(const double *vec, double *out, const int *igrid, int ngrid, long ncoarse, long ngridmax, hipStream_t s);
hipError_t mgamr_launch_scatter(double *vec, const double *in, const int *igrid, int nact, int ngrid, long ncoarse, long ngridmax, hipStream_t s);
hipError_t mgamr_launch_gather_scan(const int *flag2, int *out, const int *igrid, int ngrid, long ncoarse, long ngridmax, hipStream_t s);
hipError_t mgamr_launch_lookup(const int *igrid, int ngrid, int *lookup, hipStream_t s);

}  // namespace ramses_amd
