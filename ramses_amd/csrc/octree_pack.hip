// octree_pack.hip -- gather/scatter between RAMSES's octree cell vectors and the
// level brick.
//
// RAMSES addresses cell `ind` (1..8, ind-1 = ix+2*iy+4*iz) of oct `igrid` as
//   icell = ncoarse + (ind-1)*ngridmax + igrid          (1-based)
// in every per-cell array (uold(1:ncell,1:nvar), phi(1:ncell), f(1:ncell,1:3)),
// hydro/godunov_fine.f90:600-601, amr/refine_utils.f90:663-673.  Octs sit in
// linked-list / Hilbert order, so the brick position of an oct comes from its
// centre xg(igrid,1:3) (amr/amr_commons.f90:67-75).  The permutation
// oct -> brick origin is built once per (level, oct list) on the device and
// reused by every gather/scatter until the mesh changes.
#include <hip/hip_runtime.h>

#include "pack_args.hpp"

namespace ramses_amd {

// octorg[g] = brick cell index (i + n*(j + n*k)) of the oct's (0,0,0) cell
__global__ __launch_bounds__(256) void oct_origin_kernel(const int *__restrict__ igrid,
                                                          const double *__restrict__ xg, long ngridmax,
                                                          int ngrid, int n, double skipx, double skipy,
                                                          double skipz, long *__restrict__ octorg,
                                                          int *__restrict__ bad) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= ngrid) return;
  const long ig = igrid[g] - 1;  // 0-based oct slot
  // oct centre in units of the coarse cell, minus the coarse-grid offset;
  // the oct spans 2 cells of this level: origin = centre*n - 1
  const double cx = (xg[ig] - skipx) * n;
  const double cy = (xg[ig + ngridmax] - skipy) * n;
  const double cz = (xg[ig + 2 * ngridmax] - skipz) * n;
  const long i = (long)__builtin_floor(cx + 0.5) - 1;
  const long j = (long)__builtin_floor(cy + 0.5) - 1;
  const long k = (long)__builtin_floor(cz + 0.5) - 1;
  if (i < 0 || j < 0 || k < 0 || i + 1 >= n || j + 1 >= n || k + 1 >= n || (i & 1) || (j & 1) || (k & 1)) {
    atomicAdd(bad, 1);
    octorg[g] = 0;
    return;
  }
  octorg[g] = i + (long)n * (j + (long)n * k);
}

// brick[v][cell] <- cellvec[v*ncell + icell]   (gather) or the reverse (scatter)
template <bool GATHER>
__global__ __launch_bounds__(256) void oct_copy_kernel(PackArgs A) {
  const long total = (long)A.ngrid * 8;
  const long py = A.pitch_y ? A.pitch_y : (long)A.n, pz = A.pitch_z ? A.pitch_z : (long)A.n * A.n;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    // consecutive threads walk consecutive octs of one octant: the cell-vector
    // side is then contiguous whenever the oct list is (it is after defrag)
    const int ind = (int)(t / A.ngrid);
    const int g = (int)(t % A.ngrid);
    const long icell = A.ncoarse + (long)ind * A.ngridmax + (A.igrid[g] - 1);  // 0-based
    const long b = A.octorg[g] + (ind & 1) + py * ((ind >> 1) & 1) + pz * ((ind >> 2) & 1);
    for (int v = 0; v < A.nvar; v++) {
      if (GATHER) A.brick[b + (long)v * A.pitch_var] = A.cellvec[icell + (long)v * A.ncell];
      else A.cellvec[icell + (long)v * A.ncell] = A.brick[b + (long)v * A.pitch_var];
    }
  }
}

// leaf[b] = (son(icell) == 0) on the level brick (A.n, A.octorg, A.igrid as for the copies)
__global__ __launch_bounds__(256) void oct_leaf_kernel(PackArgs A, const int *__restrict__ son, int *__restrict__ leaf) {
  const long total = (long)A.ngrid * 8;
  const long py = A.pitch_y ? A.pitch_y : (long)A.n, pz = A.pitch_z ? A.pitch_z : (long)A.n * A.n;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    const int ind = (int)(t / A.ngrid);
    const int g = (int)(t % A.ngrid);
    const long icell = A.ncoarse + (long)ind * A.ngridmax + (A.igrid[g] - 1);
    const long b = A.octorg[g] + (ind & 1) + py * ((ind >> 1) & 1) + pz * ((ind >> 2) & 1);
    leaf[b] = son[icell] == 0 ? 1 : 0;
  }
}
hipError_t launch_oct_leaf(const PackArgs &A, const int *son, int *leaf, hipStream_t s) {
  long grid = ((long)A.ngrid * 8 + 255) / 256;
  if (grid > 8192) grid = 8192;
  if (grid < 1) grid = 1;
  hipLaunchKernelGGL(oct_leaf_kernel, dim3((int)grid), dim3(256), 0, s, A, son, leaf);
  return hipGetLastError();
}

hipError_t launch_oct_origin(const int *igrid, const double *xg, long ngridmax, int ngrid, int n,
                             const double skip[3], long *octorg, int *bad, hipStream_t s) {
  hipLaunchKernelGGL(oct_origin_kernel, dim3((ngrid + 255) / 256), dim3(256), 0, s, igrid, xg, ngridmax, ngrid, n,
                     skip[0], skip[1], skip[2], octorg, bad);
  return hipGetLastError();
}

hipError_t launch_oct_copy(const PackArgs &A, bool gather, hipStream_t s) {
  long total = (long)A.ngrid * 8;
  long grid = (total + 255) / 256;
  if (grid > 8192) grid = 8192;
  if (grid < 1) grid = 1;
  if (gather) hipLaunchKernelGGL(oct_copy_kernel<true>, dim3((int)grid), dim3(256), 0, s, A);
  else hipLaunchKernelGGL(oct_copy_kernel<false>, dim3((int)grid), dim3(256), 0, s, A);
  return hipGetLastError();
}

}  // namespace ramses_amd

#include "warm.hpp"
RAMSES_AMD_TU_WARM(octree_pack)
