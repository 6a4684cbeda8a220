// amr_args.hpp -- argument block of the coarse<->fine operators.
#pragma once
#include <hip/hip_runtime.h>

namespace ramses_amd {

struct AmrOpArgs {
  double *coarse;   // [nvar][nc][nc][nc]
  double *fine;     // [nvar][2nc][2nc][2nc]
  int nc, nvar;
  int interpol_var, interpol_type;
  double smallr;
};

hipError_t launch_amr_op(const AmrOpArgs &A, bool prolong, hipStream_t s);

}  // namespace ramses_amd
