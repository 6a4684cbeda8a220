// mg_amr.hip -- multigrid operators on AMR levels (partially refined level,
// masked cells, Dirichlet boundaries captured by the mask), one kernel per
// reference routine:
//   gauss_seidel_mg_fine / _coarse          poisson/multigrid_fine_fine.f90:332-451, multigrid_fine_coarse.f90:411-593
//   cmp_residual_mg_fine / _coarse          :147-249, multigrid_fine_coarse.f90:167-329
//   cmp_residual_norm2_fine                 :254-287
//   restrict_residual_fine/coarse_reverse   :528-590, multigrid_fine_coarse.f90:692-764
//   interpolate_and_correct_fine/_coarse    :596-698, multigrid_fine_coarse.f90:769-886
//
// Every level of a solve (the AMR level itself and the multigrid levels the
// reference builds under it, build_parent_comms_mg) is held in the reference's
// multigrid layout: cell (ind, i) of the i-th oct of the level's list at
// ind*ngrid + i, with u1 = phi/correction, u2 = rhs, u3 = residual, u4 = mask
// and one scan flag per cell.  Neighbours are found through the tree
// (son(nbor(oct, dir))) and the lookup table oct -> position in the level's list.
// Arithmetic and operation order are the reference's (bit parity,
// -ffp-contract=off); red/black cells of one colour never read each other, so
// the sequential loops parallelise without changing a bit.
#include <hip/hip_runtime.h>

#include "mg_amr_args.hpp"

namespace ramses_amd {
namespace mgamr {

// neighbour of cell (ind, oct position i) in direction (axis, up): returns the cell index
// in the level's layout, -1 if the neighbour oct does not exist in the tree, -2 if it exists
// but is not part of this multigrid level
__device__ __forceinline__ long nbr_cell(const MgAmrLevel &L, const MgAmrTree &T, int ind, int i, int axis, int up) {
  const int bit = (ind >> axis) & 1;
  const int jnd = ind ^ (1 << axis);
  if (bit != up) return (long)jnd * L.ngrid + i;          // inside the same oct
  const int g = L.igrid[i];
  const int nb = T.nbor[(long)(2 * axis + up) * T.ngridmax + g - 1];
  const int g2 = T.son[nb - 1];
  if (g2 == 0) return -1;
  const int j = T.lookup[g2 - 1];
  // (membership is checked, not assumed: an oct outside the level -- a physical-boundary oct --
  //  may carry a stale entry of an earlier solve)
  if (j <= 0 || j > L.ngrid || L.igrid[j - 1] != g2) return -2;
  return (long)jnd * L.ngrid + (j - 1);
}

// one colour of red-black Gauss-Seidel.  color 0: octants 1,4,6,7 (red), 1: 2,3,5,8 (black)
__global__ __launch_bounds__(256) void gs_kernel(MgAmrLevel L, MgAmrTree T, int color, int safe, double dx2) {
  const long total = 4L * L.nact;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    const int i = (int)(t % L.nact);
    const int k = (int)(t / L.nact);
    const int red[4] = {0, 3, 5, 6}, black[4] = {1, 2, 4, 7};
    const int ind = color ? black[k] : red[k];
    const long c = (long)ind * L.ngrid + i;
    double nb_sum = 0.0;
    if (L.scan[c] == 0) {
      // inner cell: all neighbours exist and are active
#pragma unroll
      for (int up = 0; up < 2; up++)
#pragma unroll
        for (int axis = 0; axis < 3; axis++) nb_sum = nb_sum + L.u1[nbr_cell(L, T, ind, i, axis, up)];
      L.u1[c] = (nb_sum - dx2 * L.u2[c]) / 6.0;
    } else {
      const double m = L.u4[c];
      if (m <= 0.0) continue;
      if (safe && m < 1.0) continue;
      double weight = 0.0;
#pragma unroll
      for (int up = 0; up < 2; up++)
#pragma unroll
        for (int axis = 0; axis < 3; axis++) {
          const long n = nbr_cell(L, T, ind, i, axis, up);
          if (n < 0) {
            weight = weight - 1.0 / m;
          } else if (L.u4[n] <= 0.0) {
            weight = weight + L.u4[n] / m;
          } else {
            nb_sum = nb_sum + L.u1[n];
          }
        }
      L.u1[c] = (nb_sum - dx2 * L.u2[c]) / (6.0 - weight);
    }
  }
}

// u3 = -(sum_nb - 6 phi)/dx^2 + rhs, masked cells 0
__global__ __launch_bounds__(256) void residual_kernel(MgAmrLevel L, MgAmrTree T, double oneoverdx2) {
  const long total = 8L * L.nact;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    const int i = (int)(t % L.nact);
    const int ind = (int)(t / L.nact);
    const long c = (long)ind * L.ngrid + i;
    const double phi_c = L.u1[c];
    double nb_sum = 0.0;
    if (L.scan[c] == 0) {
#pragma unroll
      for (int up = 0; up < 2; up++)
#pragma unroll
        for (int axis = 0; axis < 3; axis++) nb_sum = nb_sum + L.u1[nbr_cell(L, T, ind, i, axis, up)];
    } else {
      const double m = L.u4[c];
      if (m <= 0.0) { L.u3[c] = 0.0; continue; }
      // the scan branch runs over the directions first, then left/right
#pragma unroll
      for (int axis = 0; axis < 3; axis++)
#pragma unroll
        for (int up = 0; up < 2; up++) {
          const long n = nbr_cell(L, T, ind, i, axis, up);
          if (n < 0) {
            nb_sum = nb_sum - phi_c / m;
          } else if (L.u4[n] <= 0.0) {
            nb_sum = nb_sum + phi_c * (L.u4[n] / m);
          } else {
            nb_sum = nb_sum + L.u1[n];
          }
        }
    }
    L.u3[c] = -oneoverdx2 * (nb_sum - 6.0 * phi_c) + L.u2[c];
  }
}

// partial sums of u3^2 over unmasked cells (fixed-order tree per block)
__global__ __launch_bounds__(256) void norm_kernel(MgAmrLevel L, double *partial) {
  const long total = 8L * L.nact;
  double acc = 0.0;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    const long c = (long)(t / L.nact) * L.ngrid + (t % L.nact);
    if (L.u4[c] > 0.0) acc = acc + L.u3[c] * L.u3[c];
  }
  __shared__ double sm[256];
  sm[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) sm[threadIdx.x] = sm[threadIdx.x] + sm[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = sm[0];
}
__global__ void norm_final_kernel(const double *partial, int m, double scale, double *out) {
  __shared__ double sm[256];
  double acc = 0.0;
  for (int t = threadIdx.x; t < m; t += 256) acc = acc + partial[t];
  sm[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) sm[threadIdx.x] = sm[threadIdx.x] + sm[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = scale * sm[0];
}

// restriction of the residual of level F into the rhs of level C (one thread per oct of F:
// the 8 children of a coarse cell are added in octant order, as the reference's loop does);
// the coarse rhs and correction were zeroed by the launcher
__global__ __launch_bounds__(256) void restrict_kernel(MgAmrLevel F, MgAmrLevel C, MgAmrTree T) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < F.nact; i += gridDim.x * blockDim.x) {
    const int g = F.igrid[i];
    const int fc = T.father[g - 1];                    // father cell (1-based), a cell of an oct of level C
    const int ind_c = (int)((fc - T.ncoarse - 1) / T.ngridmax);
    const int g_c = (int)(fc - T.ncoarse - (long)ind_c * T.ngridmax);
    const int j = T.lookup[g_c - 1];
    if (j <= 0 || j > C.ngrid || C.igrid[j - 1] != g_c) continue;
    const long cc = (long)ind_c * C.ngrid + (j - 1);
    if (C.u4[cc] <= 0.0) continue;
    double acc = 0.0;
#pragma unroll
    for (int ind = 0; ind < 8; ind++) {
      const long c = (long)ind * F.ngrid + i;
      if (F.u4[c] <= 0.0) continue;
      acc = acc + F.u3[c] / 8.0;
    }
    C.u2[cc] = acc;
  }
}

// same-level neighbour of an AMR cell (1-based index) in direction dir; 0 if its oct does not exist
__device__ __forceinline__ int amr_nbor_cell(int c, int dir, const MgAmrTree &T) {
  const int pos = (int)((c - T.ncoarse - 1) / T.ngridmax);
  const int g = (int)(c - T.ncoarse - (long)pos * T.ngridmax);
  const int axis = dir >> 1, up = dir & 1;
  const int bit = (pos >> axis) & 1;
  if (bit != up) return c + (up ? 1 : -1) * (int)((1 << axis) * T.ngridmax);
  const int nb = T.nbor[(long)dir * T.ngridmax + g - 1];
  const int g2 = T.son[nb - 1];
  if (g2 == 0) return 0;
  return (int)(T.ncoarse + (long)(pos ^ (1 << axis)) * T.ngridmax + g2);
}

// phi_F += trilinear interpolation of the correction of level C (8 of the 27 father cells
// around the oct, weights 1,3,3,9,3,9,9,27 /64 in the reference's order)
__global__ __launch_bounds__(256) void interp_kernel(MgAmrLevel F, MgAmrLevel C, MgAmrTree T) {
  const long total = 8L * F.nact;
  const double a = 1.0 / 64.0, b = 3 * a, cc = 9 * a, d = 27 * a;
  const double bbb[8] = {a, b, b, cc, b, cc, cc, d};
  // ccc(ind_average, ind_f): which of the 27 father cells (1-based, x fastest)
  const int ccc[8][8] = {{1, 2, 4, 5, 10, 11, 13, 14},   {3, 2, 6, 5, 12, 11, 15, 14},  {7, 8, 4, 5, 16, 17, 13, 14},
                         {9, 8, 6, 5, 18, 17, 15, 14},   {19, 20, 22, 23, 10, 11, 13, 14}, {21, 20, 24, 23, 12, 11, 15, 14},
                         {25, 26, 22, 23, 16, 17, 13, 14}, {27, 26, 24, 23, 18, 17, 15, 14}};
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    const int i = (int)(t % F.nact);
    const int ind_f = (int)(t / F.nact);
    const long c = (long)ind_f * F.ngrid + i;
    double corr = 0.0;
    if (F.u4[c] > 0.0) {
      const int g = F.igrid[i];
      const int f0 = T.father[g - 1];
#pragma unroll 1
      for (int av = 0; av < 8; av++) {
        const int t = ccc[ind_f][av] - 1;
        const int d3[3] = {t % 3 - 1, (t / 3) % 3 - 1, t / 9 - 1};
        int fc = f0;
        for (int axis = 0; axis < 3 && fc > 0; axis++)
          if (d3[axis] != 0) fc = amr_nbor_cell(fc, 2 * axis + (d3[axis] > 0 ? 1 : 0), T);
        if (fc <= 0) continue;   // cannot happen: the 3^3 father cells exist
        const int ind_c = (int)((fc - T.ncoarse - 1) / T.ngridmax);
        const int g_c = (int)(fc - T.ncoarse - (long)ind_c * T.ngridmax);
        const int j = T.lookup[g_c - 1];
        if (j <= 0 || j > C.ngrid || C.igrid[j - 1] != g_c) continue;
        corr = corr + bbb[av] * C.u1[(long)ind_c * C.ngrid + (j - 1)];
      }
    }
    F.u1[c] = F.u1[c] + corr;
  }
}

// AMR layout <-> multigrid layout of the fine level: cell vector v(1:ncell) at
// ncoarse + ind*ngridmax + igrid(i)
__global__ void gather_kernel(const double *vec, double *out, const int *igrid, int ngrid, long ncoarse, long ngridmax) {
  const long total = 8L * ngrid;
  for (long c = (long)blockIdx.x * blockDim.x + threadIdx.x; c < total; c += (long)gridDim.x * blockDim.x) {
    const int i = (int)(c % ngrid), ind = (int)(c / ngrid);
    out[c] = vec[ncoarse + (long)ind * ngridmax + igrid[i] - 1];
  }
}
// (only the first nact octs of a layout of ngrid octs are written back)
__global__ void scatter_kernel(double *vec, const double *in, const int *igrid, int nact, int ngrid, long ncoarse, long ngridmax) {
  const long total = 8L * nact;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    const int i = (int)(t % nact), ind = (int)(t / nact);
    vec[ncoarse + (long)ind * ngridmax + igrid[i] - 1] = in[(long)ind * ngrid + i];
  }
}
// scan flag of the fine level: flag2(cell)/ngridmax
__global__ void gather_scan_kernel(const int *flag2, int *out, const int *igrid, int ngrid, long ncoarse, long ngridmax) {
  const long total = 8L * ngrid;
  for (long c = (long)blockIdx.x * blockDim.x + threadIdx.x; c < total; c += (long)gridDim.x * blockDim.x) {
    const int i = (int)(c % ngrid), ind = (int)(c / ngrid);
    out[c] = (int)(flag2[ncoarse + (long)ind * ngridmax + igrid[i] - 1] / ngridmax);
  }
}
// lookup[oct-1] = position (1-based) in the level's list
__global__ void lookup_kernel(const int *igrid, int ngrid, int *lookup) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < ngrid) lookup[igrid[i] - 1] = i + 1;
}

static inline int grid_for(long work, int cap = 4096) {
  long g = (work + 255) / 256;
  if (g < 1) g = 1;
  if (g > cap) g = cap;
  return (int)g;
}

}  // namespace mgamr

using namespace mgamr;

hipError_t mgamr_launch_gs(const MgAmrLevel &L, const MgAmrTree &T, int color, int safe, double dx2, hipStream_t s) {
  if (L.ngrid <= 0) return hipSuccess;
  hipLaunchKernelGGL(gs_kernel, dim3(grid_for(4L * L.nact)), dim3(256), 0, s, L, T, color, safe, dx2);
  return hipGetLastError();
}
hipError_t mgamr_launch_residual(const MgAmrLevel &L, const MgAmrTree &T, double oneoverdx2, hipStream_t s) {
  if (L.ngrid <= 0) return hipSuccess;
  hipLaunchKernelGGL(residual_kernel, dim3(grid_for(8L * L.nact)), dim3(256), 0, s, L, T, oneoverdx2);
  return hipGetLastError();
}
hipError_t mgamr_launch_norm(const MgAmrLevel &L, double scale, double *partial, double *out, hipStream_t s) {
  const int blocks = L.ngrid > 0 ? grid_for(8L * L.nact, 1024) : 1;
  if (L.ngrid > 0) hipLaunchKernelGGL(norm_kernel, dim3(blocks), dim3(256), 0, s, L, partial);
  else hipMemsetAsync(partial, 0, sizeof(double), s);
  hipLaunchKernelGGL(norm_final_kernel, dim3(1), dim3(256), 0, s, partial, blocks, scale, out);
  return hipGetLastError();
}
hipError_t mgamr_launch_restrict(const MgAmrLevel &F, const MgAmrLevel &C, const MgAmrTree &T, hipStream_t s) {
  if (C.ngrid > 0) {
    hipMemsetAsync(C.u2, 0, sizeof(double) * 8 * C.ngrid, s);
    hipMemsetAsync(C.u1, 0, sizeof(double) * 8 * C.ngrid, s);
  }
  if (F.ngrid <= 0 || C.ngrid <= 0) return hipGetLastError();
  hipLaunchKernelGGL(restrict_kernel, dim3(grid_for(F.nact)), dim3(256), 0, s, F, C, T);
  return hipGetLastError();
}
hipError_t mgamr_launch_interp(const MgAmrLevel &F, const MgAmrLevel &C, const MgAmrTree &T, hipStream_t s) {
  if (F.ngrid <= 0) return hipSuccess;
  hipLaunchKernelGGL(interp_kernel, dim3(grid_for(8L * F.nact)), dim3(256), 0, s, F, C, T);
  return hipGetLastError();
}
hipError_t mgamr_launch_gather(const double *vec, double *out, const int *igrid, int ngrid, long ncoarse, long ngridmax,
                               hipStream_t s) {
  if (ngrid <= 0) return hipSuccess;
  hipLaunchKernelGGL(gather_kernel, dim3(grid_for(8L * ngrid)), dim3(256), 0, s, vec, out, igrid, ngrid, ncoarse, ngridmax);
  return hipGetLastError();
}
hipError_t mgamr_launch_scatter(double *vec, const double *in, const int *igrid, int nact, int ngrid, long ncoarse, long ngridmax,
                                hipStream_t s) {
  if (nact <= 0) return hipSuccess;
  hipLaunchKernelGGL(scatter_kernel, dim3(grid_for(8L * nact)), dim3(256), 0, s, vec, in, igrid, nact, ngrid, ncoarse, ngridmax);
  return hipGetLastError();
}
hipError_t mgamr_launch_gather_scan(const int *flag2, int *out, const int *igrid, int ngrid, long ncoarse, long ngridmax,
                                    hipStream_t s) {
  if (ngrid <= 0) return hipSuccess;
  hipLaunchKernelGGL(gather_scan_kernel, dim3(grid_for(8L * ngrid)), dim3(256), 0, s, flag2, out, igrid, ngrid, ncoarse,
                     ngridmax);
  return hipGetLastError();
}
hipError_t mgamr_launch_lookup(const int *igrid, int ngrid, int *lookup, hipStream_t s) {
  if (ngrid <= 0) return hipSuccess;
  hipLaunchKernelGGL(lookup_kernel, dim3((ngrid + 255) / 256), dim3(256), 0, s, igrid, ngrid, lookup);
  return hipGetLastError();
}

}  // namespace ramses_amd

#include "warm.hpp"
RAMSES_AMD_TU_WARM(mg_amr)
