// warm.hpp -- one empty kernel per translation unit.  HIP loads the device code of a translation unit when its
// first kernel is launched (tens of milliseconds each); ramses_amd_warmup() (capi.hip) launches all of them once,
// from the first shim the program reaches -- during the reference's initialisation, before its timed loop.
#pragma once
#include <hip/hip_runtime.h>

#define RAMSES_AMD_TU_WARM(tag)                                                                    \
  namespace { __global__ void warm_kernel_##tag() {} }                                             \
  extern "C" int ramses_amd_warm_##tag(void) {                                                     \
    hipLaunchKernelGGL(warm_kernel_##tag, dim3(1), dim3(1), 0, nullptr);                           \
    return hipGetLastError() == hipSuccess ? 0 : 1;                                                \
  }
