// cg_amr_args.hpp -- argument block and launchers of the conjugate-gradient Poisson solver kernels.
#pragma once
#include <hip/hip_runtime.h>

namespace ramses_amd {

// One AMR level in the reference's own cell vectors (cell = ncoarse + ind*ngridmax + igrid - 1):
// x = phi, r = f(:,1), p = f(:,2), z = A p = f(:,3)   (poisson/phi_fine_cg.f90:18-23)
struct CgLevel {
  int ngrid;
  const int *igrid;     // [ngrid] 1-based oct indices of the level, list order
  const int *nb;        // [6][ngrid] son(nbor(igrid,k)): neighbouring oct or 0
  long ncoarse, ngridmax;
  double *x, *r, *p, *z;
  double *scal;         // device scalars: [0] r2, [1] r2 of the previous iteration, [2] pAp, [3] rhs norm^2,
                        // [6] (as unsigned) count of finished blocks of the running reduction
  double *host_r2;      // pinned ring of 4 (device-visible): r2 of iteration k goes to slot k & 3
  double *partial;      // [CG_MAX_BLOCKS] per-block partial sums
  double *prod;         // [8*ngrid] products in the reference's summation order (ordered modes) or nullptr
  void *scan;           // ordered sums by the parallel parity scan (parity_scan.hpp): cg_scan_bytes(ngrid) bytes of scratch;
                        // nullptr with prod set: one lane adds the products one after the other (verification of the scan)
};
size_t cg_scan_bytes(int ngrid);
// *out = the sequential sum of x[0..n-1] (parity scan); scratch: ordered_sum_bytes(n)
size_t ordered_sum_bytes(long n);
hipError_t ordered_sum_launch(const double *x, long n, double *out, void *scratch, hipStream_t s);
constexpr int CG_MAX_BLOCKS = 1024;
enum { CG_R2 = 0, CG_R2_OLD = 1, CG_PAP = 2, CG_RHS = 3 };

hipError_t cg_launch_setup(const int *igrid, int ngrid, const int *son, const int *nbor, long ngridmax, int *nb, hipStream_t s);
// scal[CG_RHS] = sum fact2*(rho-rho_tot)^2 over the level
hipError_t cg_launch_rhs_norm(const CgLevel &L, const double *rho, double rho_tot, double fact2, hipStream_t s);
// scal[CG_R2] = r.r   (start of the first iteration), also stored to host_r2[slot]
hipError_t cg_launch_dot_rr(const CgLevel &L, int slot, hipStream_t s);
// one iteration (:96-183): p = r + beta p; z = A p; pAp; x += alpha p; r -= alpha z; r2 of the new r
// (stored to host_r2[slot]).  Three launches (parallel sums), eleven (ordered sums by the scan) or five (one-lane chain).
hipError_t cg_launch_iteration(const CgLevel &L, int iter, int slot, hipStream_t s);
// its three routines one by one (several MPI ranks: the caller reduces scal[CG_PAP] / scal[CG_R2] over the ranks in between)
hipError_t cg_launch_update_p(const CgLevel &L, int iter, hipStream_t s);
hipError_t cg_launch_ap(const CgLevel &L, hipStream_t s);
hipError_t cg_launch_update_xr(const CgLevel &L, hipStream_t s);

}  // namespace ramses_amd
