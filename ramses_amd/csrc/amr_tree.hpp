// amr_tree.hpp -- steps through the reference's tree arrays (son, nbor, father) shared by the tree-walking sweeps
// (amr_sweep.hip: one father oct or one oct per workgroup / wavefront; amr_block.hip: one grandfather oct per workgroup):
//   getnborfather       amr/nbors_utils.f90:404-525
//   get3cubefather      amr/nbors_utils.f90:5-194
#pragma once
#include <hip/hip_runtime.h>

#include "amr_sweep_args.hpp"

namespace ramses_amd {
namespace amrsweep {

// cell index (1-based, level >= 2) -> octant position and oct
__device__ __forceinline__ void cell_split(int c, long ncoarse, long ngridmax, int &pos, int &g) {
  pos = (int)((c - ncoarse - 1) / ngridmax);
  g = (int)(c - ncoarse - (long)pos * ngridmax);
}

// same-level neighbour of cell c in direction dir (0:-x 1:+x 2:-y 3:+y 4:-z 5:+z).
// Returns the neighbour cell; if its oct does not exist returns -(coarser cell)
// (the neighbouring father cell of c's oct: getnborfather's fallback).
__device__ __forceinline__ int nbor_cell(int c, int dir, const AmrSweepArgs &A) {
  int pos, g;
  cell_split(c, A.ncoarse, A.ngridmax, pos, g);
  const int axis = dir >> 1, up = dir & 1;
  const int bit = (pos >> axis) & 1;
  if (bit != up) return c + (up ? 1 : -1) * (int)((1 << axis) * A.ngridmax);   // sibling in the same oct
  const int nb = A.nbor[(long)dir * A.ngridmax + g - 1];
  const int g2 = A.son[nb - 1];
  if (g2 == 0) return -nb;
  return (int)(A.ncoarse + (long)(pos ^ (1 << axis)) * A.ngridmax + g2);
}

// father cell t (of the 4^3 around father oct gF): x, then y, then z steps through son(nbor(...)), the arithmetic
// of getnborfather; 0 when an oct on that path does not exist
__device__ __forceinline__ int group_father_cell(const AmrSweepArgs &A, int gF, int t) {
  const int i = t & 3, j = (t >> 2) & 3, k = t >> 4;
  const int bi = i == 0 ? 0 : (i == 3 ? 1 : i - 1), bj = j == 0 ? 0 : (j == 3 ? 1 : j - 1), bk = k == 0 ? 0 : (k == 3 ? 1 : k - 1);
  int c = (int)(A.ncoarse + (long)(bi + 2 * bj + 4 * bk) * A.ngridmax + gF);
  const int step[3] = {i == 0 ? -1 : (i == 3 ? 1 : 0), j == 0 ? -1 : (j == 3 ? 1 : 0), k == 0 ? -1 : (k == 3 ? 1 : 0)};
#pragma unroll
  for (int axis = 0; axis < 3; axis++) {
    if (step[axis] != 0 && c > 0) {
      c = nbor_cell(c, 2 * axis + (step[axis] > 0 ? 1 : 0), A);
      if (c < 0) c = 0;                      // no oct there: only octs that do not exist would need it
    }
  }
  return c;
}

}  // namespace amrsweep
}  // namespace ramses_amd
