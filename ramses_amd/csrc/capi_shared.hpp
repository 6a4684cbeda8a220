// capi_shared.hpp -- what the translation units of the C ABI share (capi.hip: bricks, halos, dense multigrid;
// capi_host.hip: the staged and the resident entry points on the reference's host arrays; capi_tree_poisson.hip: the
// multigrid and conjugate-gradient solves on AMR levels): the error path, a growable device buffer, the constants of the
// hydro kernels.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>

#include "../../include/ramses_amd.h"
#include "hydro_core.hpp"

extern "C" int ramses_amd_set_error(int code, const char *msg);   // capi.hip: the thread's last error text

namespace ramses_amd {

static inline int fail(int code, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  return ramses_amd_set_error(code, buf);
}
static inline int hipfail(hipError_t e, const char *what) {
  return fail(RAMSES_AMD_EHIP, "%s: %s", what, hipGetErrorString(e));
}

struct DevBuf {
  void *p = nullptr;
  size_t cap = 0;
  hipError_t ensure(size_t bytes) {
    if (bytes <= cap) return hipSuccess;
    if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
    hipError_t e = hipMalloc(&p, bytes);
    if (e == hipSuccess) cap = bytes;
    return e;
  }
  template <class T> T *as() { return reinterpret_cast<T *>(p); }
};

static inline HydroConst make_const(const ramses_amd_hydro_params *p) {
  HydroConst P;
  P.gamma = p->gamma;
  P.smallr = p->smallr;
  P.smallc = p->smallc;
  P.smallc2 = p->smallc * p->smallc;
  P.smallp = P.smallc2 / p->gamma;                       // smallc**2/gamma
  P.smalle = P.smallc2 / p->gamma / (p->gamma - 1.0);    // smallc**2/gamma/(gamma-one)
  P.entho = 1.0 / (p->gamma - 1.0);
  P.gm1 = p->gamma - 1.0;
  P.gamma6 = (p->gamma + 1.0) / (2.0 * p->gamma);
  P.smallpp = p->smallr * P.smallp;
  P.oneovergamma = 1.0 / p->gamma;
  P.slope_theta = p->slope_theta;
  P.niter_riemann = p->niter_riemann;
  return P;
}

// capi_host.hip: the staged entry points reuse the staging buffers of the resident level; refuses while that level holds
// the only current copy of the hydro state
int capi_resident_release(const char *who);

}  // namespace ramses_amd
