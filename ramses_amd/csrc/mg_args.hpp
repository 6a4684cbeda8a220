// mg_args.hpp -- launch helpers of the multigrid kernels.
#pragma once
#include <hip/hip_runtime.h>

namespace ramses_amd {

constexpr int MG_MAX_PARTIALS = 8192;      // workgroups of one fused-smoother launch (one partial sum each)
constexpr int MG_RESIDUAL_BLOCKS = 4096;   // grid cap of the per-colour residual kernel (fixes the order of its partial sums)

// levels ltop .. 1 of a dense periodic hierarchy for the single-workgroup coarse tail: w + off[l][0..2] = correction,
// right-hand side, residual of level l (2^l cells per direction)
constexpr int MG_TAIL_LTOP = 5;
struct MgTailArgs {
  double *w;
  int ltop;
  long off[MG_TAIL_LTOP + 1][3];
  double dx2[MG_TAIL_LTOP + 1], oneoverdx2[MG_TAIL_LTOP + 1];
};
hipError_t mg_launch_coarse_tail(const MgTailArgs &T, hipStream_t s);

hipError_t mg_launch_rhs(const double *rho, double *f2, long N, double fourpi, double rho_tot, hipStream_t s);
hipError_t mg_launch_gs(double *phi, const double *rhs, int n, double dx2, int color, hipStream_t s);
int mg_residual_blocks(int n);
hipError_t mg_launch_residual(const double *phi, const double *rhs, double *res, int n, double dx,
                              double *partial, double *norm_out, hipStream_t s);
hipError_t mg_launch_restrict(const double *res_f, double *rhs_c, double *u1_c, int nf, hipStream_t s);
hipError_t mg_launch_interp(double *phi_f, const double *corr_c, int nf, hipStream_t s);
hipError_t mg_launch_smooth_fused(const double *phi_in, double *phi_out, const double *rhs, double *res,
                                  double *partial, double *norm_out, int n, double dx, int npass,
                                  hipStream_t s, int ng = 0, double *rhs_c = nullptr, double *u1_c = nullptr,
                                  const double *corr_c = nullptr, int ny = 0, int nz = 0);
// ny, nz (with ng > 0 only): the extents of a brick that is not a cube, n being the extent along x; 0 = n
// rhs_c / u1_c given (dense periodic level, 2 colour passes on 32-row tiles: mg_smooth_can_restrict): the kernel restricts its
// residual into the coarse right-hand side and zeroes the coarse correction itself; res may then be NULL.
// corr_c given (same condition, no residual): the planes the kernel reads are phi_in + the prolongation of corr_c
bool mg_smooth_can_restrict(int n, int npass);
// one rank's brick of a distributed level (ng ghost layers)
hipError_t mg_launch_restrict_ghost(const double *res_f, double *rhs_c, int nfx, int nfy, int nfz, int ngf, int ngc, hipStream_t s);
hipError_t mg_launch_interp_ghost(double *phi_f, int nfx, int nfy, int nfz, int ngf, const double *corr_c, int ngc, int cglob,
                                  int cox, int coy, int coz, hipStream_t s);
hipError_t mg_launch_gradient_ghost(const double *phi, double *f, int nx, int ny, int nz, int ng, double a, double b, hipStream_t s);
void mg_set_smooth_rows(int ly);
hipError_t mg_launch_gradient(const double *phi, double *f, int n, double a, double b, hipStream_t s);

}  // namespace ramses_amd
