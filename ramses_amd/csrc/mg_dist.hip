// mg_dist.hip -- multigrid_fine of a periodic, fully refined level cut into one brick per rank (one rank per GPU):
// the V-cycle driver WITH its halo exchanges behind the C ABI (ramses_amd_mgdist_*), so that the Fortran shim
// (ramses_amd/patch/multigrid_fine_commons.f90) and the Python mirror (ramses_amd/poisson_parallel.py) bind the same
// entry points.  Reference:
//     multigrid_fine(ilevel,icount)      poisson/multigrid_fine_commons.f90:25-296
//     recursive_multigrid_coarse         :307-390
//     make_virtual_mg_dp / _fine_dp      :1172-1290, amr/virtual_boundaries.f90:373-528   (the exchanges replaced here)
//     force_fine / gradient_phi          poisson/force_fine.f90:5-324
//
// The reference's box is a cube; 2^k ranks own bricks of power-of-two extents (2 ranks: half boxes, 4: quarter columns,
// 8: octants -- the Hilbert decomposition of a uniform level).  MI355X-first choices:
//  * every level of a rank is a brick inside NG = 5 ghost layers; the fused smoother recomputes the neighbours' updates
//    inside them, so ONE 5-cell exchange per smoother launch replaces the reference's exchange after every colour pass;
//    the 26 neighbour regions travel as one message per peer (7 peers on a 2x2x2 node: every xGMI link at once);
//  * levels whose brick would fall below the smoother's 64-cell tile in any direction are REPLICATED: one all-gather of
//    the restricted residual, then every rank runs the rest of the V-cycle on the whole coarse level (the single-GPU
//    code) and reads its part of the correction;
//  * transport: RCCL inside this library (ramses_amd_rccl_*), or -- ranks sharing a GPU, CPU protocol tests -- the
//    caller's own message layer through three callbacks on pinned host buffers.
// All arithmetic is that of the dense single-brick kernels in the reference's operation order: phi equals the
// single-rank solve bit for bit whenever the iteration counts agree.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/ramses_amd.h"
#include "mg_args.hpp"
#include "pack_args.hpp"

using namespace ramses_amd;

extern "C" int ramses_amd_set_error(int code, const char *msg);   // capi.hip

namespace {

constexpr int NG = 5;            // ghost layers: 4 colour passes + the residual's stencil
constexpr int MIN_FUSED = 64;    // the fused smoother's tile width
constexpr int MAXITER = 10;      // multigrid_fine_commons.f90:34
constexpr double SAFE_FACTOR = 0.5;   // :35

int failf(int code, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  return ramses_amd_set_error(code, buf);
}
#define HCHK(call, what) do { hipError_t e_ = (call); if (e_ != hipSuccess) return failf(RAMSES_AMD_EHIP, "%s: %s", what, hipGetErrorString(e_)); } while (0)
#define RCHK(call) do { int rc_ = (call); if (rc_) return rc_; } while (0)

struct Seg {
  int peer;
  int64_t off, cnt;
};

// the 26 neighbour regions of a brick, ordered by peer so that every peer gets ONE message.  Region g (an offset in
// {-1,0,1}^3) of MY ghost layers is filled from the neighbour at coords+g, which sends its interior region next to its
// side -g; both sides order a peer's regions by the sender's offset index.
struct HaloPlan {
  int boxes_s[26 * 6], boxes_r[26 * 6];
  int64_t offs_s[26], offs_r[26];
  std::vector<Seg> segs_s, segs_r;   // one per peer, same peers in the same order on both sides
  int64_t total = 0;
  double *d_send = nullptr, *d_recv = nullptr;
  double *h_send = nullptr, *h_recv = nullptr;   // pinned, callback transport only
};

struct Level {
  int l = 0;
  int n[3] = {0, 0, 0};
  ramses_amd_brick brick;
  size_t cells = 0;        // allocated cells (with ghosts)
  double *u[4] = {nullptr, nullptr, nullptr, nullptr};
  HaloPlan plan;
  bool built = false;
};

// whole replicated level from the all-gathered parts: parts[r] = the [nz][ny][nx] brick of rank r
__global__ __launch_bounds__(256) void assemble_kernel(const double *__restrict__ parts, double *__restrict__ cube,
                                                       const int *__restrict__ rank_of_brick, int px, int py, int nx,
                                                       int ny, int nz, int N) {
  const long total = (long)N * N * N;
  const long part = (long)nx * ny * nz;
  for (long c = (long)blockIdx.x * blockDim.x + threadIdx.x; c < total; c += (long)gridDim.x * blockDim.x) {
    const int i = (int)(c % N), j = (int)((c / N) % N), k = (int)(c / ((long)N * N));
    const int bx = i / nx, by = j / ny, bz = k / nz;
    const int r = rank_of_brick[bx + px * (by + py * bz)];
    cube[c] = parts[(long)r * part + (i - bx * nx) + (long)nx * ((j - by * ny) + (long)ny * (k - bz * nz))];
  }
}

}  // namespace

struct ramses_amd_mgdist {
  int level = 0, pgrid[3] = {1, 1, 1}, coords[3] = {0, 0, 0}, rank = 0, world = 1;
  int dims[3] = {0, 0, 0};
  std::vector<int> rank_of_brick;
  int *d_rank_of_brick = nullptr;
  bool use_rccl = false;
  bool rccl_self = false;    // RAMSES_AMD_MGDIST_RCCL_SELF=1 (tests): with the library's communicator up, a rank's messages to itself
                             // (periodic wrap), the all-reduce and the all-gather go through RCCL even on ONE rank -- what N ranks execute
  ramses_amd_mg_transport tr;
  std::vector<Level> lev;     // indexed by level; built = distributed
  int lrep = 0, rep_dims[3] = {0, 0, 0};
  Level rep_local;
  double *rep_rhs = nullptr, *rep_u1 = nullptr, *rep_work = nullptr, *rep_parts = nullptr, *rep_mine = nullptr;
  double *h_parts = nullptr, *h_mine = nullptr;
  double *work = nullptr, *norm = nullptr, *dense = nullptr;   // dense: one brick without ghosts (rhs staging)
  int safe_mode = 0;
  bool phi_fresh = false;      // u[0] of the finest level holds the potential of the last solve (force_fine may use it)
  int64_t exchanges = 0;
  int last_iters = 0;
  double last_err = 0.0;
  // the convergence test in the REFERENCE's summation order (cmp_residual_norm2_fine, poisson/multigrid_fine_fine.f90:254-287:
  // octant by octant over the rank's oct list, one double after the other): order[k] = dense-brick index of the k-th cell of
  // that loop (ramses_amd_mgdist_set_order); without it the norm is the smoother's own reduction tree
  int *d_order = nullptr;
  long order_n = 0;
  double *ord_x = nullptr;
  void *ord_scratch = nullptr;
};

namespace {

// host only: the regions and messages of one rank's brick of n[0] x n[1] x n[2] cells inside ng ghost layers
int plan_regions(const int *pgrid, const int *coords, const int *rank_of_brick, const int *n, int ng, HaloPlan &P) {
  int offs[26][3], nof = 0;
  for (int oz = -1; oz <= 1; oz++)
    for (int oy = -1; oy <= 1; oy++)
      for (int ox = -1; ox <= 1; ox++)
        if (ox || oy || oz) { offs[nof][0] = ox; offs[nof][1] = oy; offs[nof][2] = oz; nof++; }
  auto index_of = [&](int ox, int oy, int oz) {
    for (int i = 0; i < 26; i++) if (offs[i][0] == ox && offs[i][1] == oy && offs[i][2] == oz) return i;
    return -1;
  };
  auto peer = [&](const int *o) {
    int c[3];
    for (int d = 0; d < 3; d++) c[d] = (((coords[d] + o[d]) % pgrid[d]) + pgrid[d]) % pgrid[d];
    return rank_of_brick[c[0] + pgrid[0] * (c[1] + pgrid[1] * c[2])];
  };
  int order_s[26], order_r[26];
  for (int i = 0; i < 26; i++) order_s[i] = order_r[i] = i;
  std::sort(order_s, order_s + 26, [&](int a, int b) {
    const int pa = peer(offs[a]), pb = peer(offs[b]);
    return pa != pb ? pa < pb : a < b;
  });
  std::sort(order_r, order_r + 26, [&](int a, int b) {
    const int pa = peer(offs[a]), pb = peer(offs[b]);
    if (pa != pb) return pa < pb;
    return index_of(-offs[a][0], -offs[a][1], -offs[a][2]) < index_of(-offs[b][0], -offs[b][1], -offs[b][2]);
  });
  P.segs_s.clear(); P.segs_r.clear();
  for (int side = 0; side < 2; side++) {
    const int *order = side == 0 ? order_s : order_r;
    int *boxes = side == 0 ? P.boxes_s : P.boxes_r;
    int64_t *offsets = side == 0 ? P.offs_s : P.offs_r;
    std::vector<Seg> &segs = side == 0 ? P.segs_s : P.segs_r;
    int64_t pos = 0;
    for (int r = 0; r < 26; r++) {
      const int *o = offs[order[r]];
      int64_t size = 1;
      for (int d = 0; d < 3; d++) {
        int org, ext = o[d] == 0 ? n[d] : ng;
        if (side == 0) org = o[d] <= 0 ? ng : n[d];                           // interior cells next to side o
        else org = o[d] < 0 ? 0 : (o[d] == 0 ? ng : ng + n[d]);              // ghost cells on side g
        boxes[6 * r + d] = org;
        boxes[6 * r + 3 + d] = ext;
        size *= ext;
      }
      offsets[r] = pos;
      const int q = peer(o);
      if (!segs.empty() && segs.back().peer == q) segs.back().cnt += size;
      else segs.push_back(Seg{q, pos, size});
      pos += size;
    }
    P.total = pos;
  }
  if (P.segs_s.size() != P.segs_r.size()) return failf(RAMSES_AMD_EINVAL, "halo plan: send and receive peers differ");
  for (size_t i = 0; i < P.segs_s.size(); i++)
    if (P.segs_s[i].peer != P.segs_r[i].peer || P.segs_s[i].cnt != P.segs_r[i].cnt)
      return failf(RAMSES_AMD_EINVAL, "halo plan: message sizes of peer %d differ", P.segs_s[i].peer);
  return 0;
}

int build_plan(ramses_amd_mgdist *M, Level &L) {
  HaloPlan &P = L.plan;
  RCHK(plan_regions(M->pgrid, M->coords, M->rank_of_brick.data(), L.n, NG, P));
  HCHK(hipMalloc(&P.d_send, sizeof(double) * P.total), "hipMalloc");
  HCHK(hipMalloc(&P.d_recv, sizeof(double) * P.total), "hipMalloc");
  if (!M->use_rccl) {
    HCHK(hipHostMalloc(&P.h_send, sizeof(double) * P.total), "hipHostMalloc");
    HCHK(hipHostMalloc(&P.h_recv, sizeof(double) * P.total), "hipHostMalloc");
  }
  return 0;
}

int build_level(ramses_amd_mgdist *M, Level &L, int l, const int *dims, bool with_plan) {
  L.l = l;
  for (int d = 0; d < 3; d++) L.n[d] = dims[d];
  ramses_amd_brick_dense(&L.brick, dims[0], dims[1], dims[2], NG);
  L.cells = (size_t)(dims[0] + 2 * NG) * (dims[1] + 2 * NG) * (dims[2] + 2 * NG);
  for (int a = 0; a < 4; a++) {
    HCHK(hipMalloc(&L.u[a], sizeof(double) * L.cells), "hipMalloc (multigrid level)");
    HCHK(hipMemset(L.u[a], 0, sizeof(double) * L.cells), "hipMemset");
  }
  L.built = true;
  if (with_plan) RCHK(build_plan(M, L));
  return 0;
}

void free_level(Level &L) {
  for (int a = 0; a < 4; a++) if (L.u[a]) { (void)hipFree(L.u[a]); L.u[a] = nullptr; }
  HaloPlan &P = L.plan;
  if (P.d_send) (void)hipFree(P.d_send);
  if (P.d_recv) (void)hipFree(P.d_recv);
  if (P.h_send) (void)hipHostFree(P.h_send);
  if (P.h_recv) (void)hipHostFree(P.h_recv);
  P.d_send = P.d_recv = P.h_send = P.h_recv = nullptr;
  L.built = false;
}

// forward halo of the brick tensor t (NG layers, faces + edges + corners) in ONE round
int exchange(ramses_amd_mgdist *M, Level &L, double *t, hipStream_t s) {
  HaloPlan &P = L.plan;
  RCHK(ramses_amd_halo_multi(&L.brick, t, 1, 26, P.boxes_s, P.offs_s, P.d_send, 1, s));
  const int np = (int)P.segs_s.size();
  std::vector<int> peers;
  std::vector<int64_t> so, sc, ro, rc;
  for (int i = 0; i < np; i++) {
    const Seg &a = P.segs_s[i], &b = P.segs_r[i];
    if (a.peer == M->rank && !M->rccl_self) {
      if (M->use_rccl)   // periodic wrap onto myself
        HCHK(hipMemcpyAsync(P.d_recv + b.off, P.d_send + a.off, sizeof(double) * a.cnt, hipMemcpyDeviceToDevice, s), "self copy");
      continue;
    }
    peers.push_back(a.peer); so.push_back(a.off); sc.push_back(a.cnt); ro.push_back(b.off); rc.push_back(b.cnt);
  }
  if (M->use_rccl) {
    if (!peers.empty())
      RCHK(ramses_amd_rccl_exchange((int)peers.size(), peers.data(), P.d_send, so.data(), sc.data(), P.d_recv, ro.data(), rc.data(), s));
  } else {
    HCHK(hipMemcpyAsync(P.h_send, P.d_send, sizeof(double) * P.total, hipMemcpyDeviceToHost, s), "halo D2H");
    HCHK(hipStreamSynchronize(s), "stream sync");
    for (int i = 0; i < np; i++)
      if (P.segs_s[i].peer == M->rank)
        std::memcpy(P.h_recv + P.segs_r[i].off, P.h_send + P.segs_s[i].off, sizeof(double) * P.segs_s[i].cnt);
    if (!peers.empty()) {
      if (!M->tr.exchange) return failf(RAMSES_AMD_EINVAL, "distributed multigrid: no transport (exchange callback missing)");
      const int rc_ = M->tr.exchange(M->tr.user, (int)peers.size(), peers.data(), P.h_send, so.data(), sc.data(), P.h_recv, ro.data(), rc.data());
      if (rc_) return failf(RAMSES_AMD_EHIP, "distributed multigrid: the transport's exchange failed (%d)", rc_);
    }
    HCHK(hipMemcpyAsync(P.d_recv, P.h_recv, sizeof(double) * P.total, hipMemcpyHostToDevice, s), "halo H2D");
  }
  RCHK(ramses_amd_halo_multi(&L.brick, t, 1, 26, P.boxes_r, P.offs_r, P.d_recv, 0, s));
  M->exchanges++;
  return 0;
}

// interior of a ghost brick <-> a dense [nz][ny][nx] array
int interior_copy(Level &L, double *t, double *dense, int pack, hipStream_t s) {
  const int box[6] = {NG, NG, NG, L.n[0], L.n[1], L.n[2]};
  const int64_t off = 0;
  return ramses_amd_halo_multi(&L.brick, t, 1, 1, box, &off, dense, pack, s);
}

int allreduce_sum(ramses_amd_mgdist *M, double *d_value, double *out, hipStream_t s) {
  if (M->use_rccl && (M->world > 1 || M->rccl_self)) RCHK(ramses_amd_rccl_allreduce(d_value, 1, 0, s));
  HCHK(hipMemcpyAsync(out, d_value, sizeof(double), hipMemcpyDeviceToHost, s), "norm copy");
  HCHK(hipStreamSynchronize(s), "stream sync");
  if (!M->use_rccl && M->world > 1) {
    if (!M->tr.allreduce_sum) return failf(RAMSES_AMD_EINVAL, "distributed multigrid: no transport (allreduce callback missing)");
    const int rc_ = M->tr.allreduce_sum(M->tr.user, out);
    if (rc_) return failf(RAMSES_AMD_EHIP, "distributed multigrid: the transport's allreduce failed (%d)", rc_);
  }
  return 0;
}

hipError_t fused(ramses_amd_mgdist *M, Level &L, const double *src, double *dst, const double *rhs, double *res, double *norm, hipStream_t s) {
  const double dx = std::ldexp(1.0, -L.l);
  return mg_launch_smooth_fused(src, dst, rhs, res, (res || norm) ? M->work : nullptr, norm, L.n[0], dx, 4, s, NG, nullptr, nullptr, nullptr,
                                L.n[1], L.n[2]);
}

// ---- the local residual norm in the reference's order (see ramses_amd_mgdist::d_order) -------------------------------------
__global__ __launch_bounds__(256) void mgdist_gather_sq_kernel(const double *__restrict__ res, const int *__restrict__ order, long n, double *__restrict__ x) {
  const long k = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (k < n) { const double r = res[order[k]]; x[k] = r * r; }
}
__global__ void mgdist_scale_kernel(double *v, double a) { v[0] = a * v[0]; }   // norm2 = dx2*norm2 (:285)
// res: the residual of level L with its ghost layers; *slot = dx^3 * (((r_1^2 + r_2^2) + r_3^2) + ...) over the rank's own cells
int ordered_norm(ramses_amd_mgdist *M, Level &L, double *res, double *slot, hipStream_t s) {
  const long N = M->order_n;
  RCHK(interior_copy(L, res, M->dense, 1, s));
  hipLaunchKernelGGL(mgdist_gather_sq_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s, M->dense, M->d_order, N, M->ord_x);
  HCHK(hipGetLastError(), "residual norm (gather)");
  RCHK(ramses_amd_ordered_sum_device(M->ord_x, N, slot, M->ord_scratch, s));
  const double dx = std::ldexp(1.0, -L.l);
  hipLaunchKernelGGL(mgdist_scale_kernel, dim3(1), dim3(1), 0, s, slot, dx * dx * dx);
  HCHK(hipGetLastError(), "residual norm (scale)");
  return 0;
}

bool distributed(const ramses_amd_mgdist *M, int l) { return l >= 1 && l < (int)M->lev.size() && M->lev[l].built; }

// restrict the residual of level l into level l-1 (its u2) and zero that level's correction
int restrict_to(ramses_amd_mgdist *M, int l, const double *res, hipStream_t s) {
  if (l - 1 < 1) return 0;
  Level &Lf = M->lev[l];
  Level *Lc = distributed(M, l - 1) ? &M->lev[l - 1] : &M->rep_local;
  if (distributed(M, l - 1)) HCHK(hipMemsetAsync(Lc->u[0], 0, sizeof(double) * Lc->cells, s), "memset");
  HCHK(mg_launch_restrict_ghost(res, Lc->u[1], Lf.n[0], Lf.n[1], Lf.n[2], NG, NG, s), "mg restrict launch");
  return 0;
}

// phi_f += prolongation of the correction of level lc
int interp_from(ramses_amd_mgdist *M, Level &Lf, double *phi_f, int lc, hipStream_t s) {
  if (distributed(M, lc)) {
    Level &Lc = M->lev[lc];
    RCHK(exchange(M, Lc, Lc.u[0], s));     // one ghost layer is needed; the region mover sends all NG
    HCHK(mg_launch_interp_ghost(phi_f, Lf.n[0], Lf.n[1], Lf.n[2], NG, Lc.u[0], NG, 0, 0, 0, 0, s), "mg interp launch");
  } else {
    HCHK(mg_launch_interp_ghost(phi_f, Lf.n[0], Lf.n[1], Lf.n[2], NG, M->rep_u1, 0, 1 << M->lrep, M->coords[0] * M->rep_dims[0],
                                M->coords[1] * M->rep_dims[1], M->coords[2] * M->rep_dims[2], s), "mg interp launch");
  }
  return 0;
}

// recursive_multigrid_coarse (multigrid_fine_commons.f90:307-390); on entry the restricted residual is in the level's
// u2 interior and u1 is zero
int coarse_cycle(ramses_amd_mgdist *M, int l, int safe, hipStream_t s) {
  if (l < 1) return 0;
  if (!distributed(M, l)) {
    // replicated levels: gather the right-hand side, solve everywhere
    const size_t part = (size_t)M->rep_dims[0] * M->rep_dims[1] * M->rep_dims[2];
    RCHK(interior_copy(M->rep_local, M->rep_local.u[1], M->rep_mine, 1, s));
    if (M->world == 1 && !M->rccl_self) {
      HCHK(hipMemcpyAsync(M->rep_parts, M->rep_mine, sizeof(double) * part, hipMemcpyDeviceToDevice, s), "gather copy");
    } else if (M->use_rccl) {
      RCHK(ramses_amd_rccl_allgather(M->rep_mine, (int64_t)part, M->rep_parts, s));
    } else {
      HCHK(hipMemcpyAsync(M->h_mine, M->rep_mine, sizeof(double) * part, hipMemcpyDeviceToHost, s), "gather D2H");
      HCHK(hipStreamSynchronize(s), "stream sync");
      if (!M->tr.allgather) return failf(RAMSES_AMD_EINVAL, "distributed multigrid: no transport (allgather callback missing)");
      const int rc_ = M->tr.allgather(M->tr.user, M->h_mine, (int64_t)part, M->h_parts);
      if (rc_) return failf(RAMSES_AMD_EHIP, "distributed multigrid: the transport's allgather failed (%d)", rc_);
      HCHK(hipMemcpyAsync(M->rep_parts, M->h_parts, sizeof(double) * part * M->world, hipMemcpyHostToDevice, s), "gather H2D");
    }
    const int N = 1 << l;
    const long total = (long)N * N * N;
    const int grid = (int)std::min<long>((total + 255) / 256, 4096);
    hipLaunchKernelGGL(assemble_kernel, dim3(grid), dim3(256), 0, s, M->rep_parts, M->rep_rhs, M->d_rank_of_brick, M->pgrid[0], M->pgrid[1],
                       M->rep_dims[0], M->rep_dims[1], M->rep_dims[2], N);
    HCHK(hipGetLastError(), "assemble launch");
    return ramses_amd_mg_coarse_solve_dense(l, M->rep_rhs, M->rep_u1, M->rep_work, safe, s);
  }
  Level &L = M->lev[l];
  RCHK(exchange(M, L, L.u[1], s));
  HCHK(fused(M, L, L.u[0], L.u[3], L.u[1], L.u[2], nullptr, s), "mg fused smoother launch");   // pre-smoothing + residual
  RCHK(restrict_to(M, l, L.u[2], s));
  RCHK(coarse_cycle(M, l - 1, safe, s));
  if (l - 1 >= 1) RCHK(interp_from(M, L, L.u[3], l - 1, s));
  RCHK(exchange(M, L, L.u[3], s));
  HCHK(fused(M, L, L.u[3], L.u[0], L.u[1], nullptr, nullptr, s), "mg fused smoother launch");  // post-smoothing
  return 0;
}

}  // namespace

extern "C" {

int ramses_amd_mgdist_create(int level, const int *pgrid, int rank, const int *rank_of_brick,
                             const ramses_amd_mg_transport *transport, ramses_amd_mgdist **out) {
  if (!pgrid || !out) return failf(RAMSES_AMD_EINVAL, "NULL argument");
  *out = nullptr;
  if (level < 1 || level > 11) return failf(RAMSES_AMD_EINVAL, "multigrid level must be in [1,11] (got %d)", level);
  int world = 1;
  for (int d = 0; d < 3; d++) {
    if (pgrid[d] < 1 || (pgrid[d] & (pgrid[d] - 1))) return failf(RAMSES_AMD_EINVAL, "distributed multigrid needs a power-of-two rank grid (got %d x %d x %d)", pgrid[0], pgrid[1], pgrid[2]);
    world *= pgrid[d];
  }
  if (rank < 0 || rank >= world) return failf(RAMSES_AMD_EINVAL, "rank %d outside the %d ranks of the grid", rank, world);
  ramses_amd_mgdist *M = new ramses_amd_mgdist();
  M->level = level; M->rank = rank; M->world = world;
  M->rank_of_brick.resize(world);
  std::vector<int> seen(world, 0);
  int mine = -1;
  for (int b = 0; b < world; b++) {
    const int r = rank_of_brick ? rank_of_brick[b] : b;
    if (r < 0 || r >= world || seen[r]) { delete M; return failf(RAMSES_AMD_EINVAL, "rank_of_brick is not a permutation of the ranks"); }
    seen[r] = 1;
    M->rank_of_brick[b] = r;
    if (r == rank) mine = b;
  }
  for (int d = 0; d < 3; d++) {
    M->pgrid[d] = pgrid[d];
    M->dims[d] = (1 << level) / pgrid[d];
    if (M->dims[d] < MIN_FUSED) {
      const int dd = M->dims[d];
      delete M;
      return failf(RAMSES_AMD_EINVAL, "per-rank brick of level %d on %d x %d x %d ranks: every extent must be >= %d (got %d)", level, pgrid[0], pgrid[1], pgrid[2], MIN_FUSED, dd);
    }
  }
  M->coords[0] = mine % pgrid[0]; M->coords[1] = (mine / pgrid[0]) % pgrid[1]; M->coords[2] = mine / (pgrid[0] * pgrid[1]);
  M->use_rccl = (transport == nullptr);
  {
    const char *e = getenv("RAMSES_AMD_MGDIST_RCCL_SELF");
    M->rccl_self = M->use_rccl && e && e[0] == '1' && ramses_amd_rccl_ready();
  }
  if (transport) M->tr = *transport; else std::memset(&M->tr, 0, sizeof(M->tr));
  if (M->use_rccl && world > 1 && !ramses_amd_rccl_ready()) {
    delete M;
    return failf(RAMSES_AMD_EINVAL, "distributed multigrid without a transport table needs the RCCL communicator (ramses_amd_rccl_init)");
  }
  int rc = 0;
  auto bail = [&](int code) { ramses_amd_mgdist_destroy(M); return code; };
  M->lev.resize(level + 1);
  int l = level, dl[3] = {M->dims[0], M->dims[1], M->dims[2]};
  while (std::min(dl[0], std::min(dl[1], dl[2])) >= MIN_FUSED && l >= 1) {
    if ((rc = build_level(M, M->lev[l], l, dl, true))) return bail(rc);
    l--;
    for (int d = 0; d < 3; d++) dl[d] /= 2;
  }
  M->lrep = l;
  for (int d = 0; d < 3; d++) M->rep_dims[d] = dl[d];
  hipError_t e = hipSuccess;
  auto dmalloc = [&](double **p, size_t n) { if (e == hipSuccess) e = hipMalloc(p, sizeof(double) * n); };
  if (M->lrep >= 1) {
    if ((rc = build_level(M, M->rep_local, M->lrep, dl, false))) return bail(rc);
    const size_t cube = (size_t)1 << (3 * M->lrep), part = (size_t)dl[0] * dl[1] * dl[2];
    const int64_t nwork = ramses_amd_mg_workspace_doubles(M->lrep + 1);
    if (nwork < 0) return bail((int)nwork);
    dmalloc(&M->rep_rhs, cube); dmalloc(&M->rep_u1, cube); dmalloc(&M->rep_work, (size_t)nwork);
    dmalloc(&M->rep_parts, part * world); dmalloc(&M->rep_mine, part);
    if (e == hipSuccess) e = hipMemset(M->rep_work, 0, sizeof(double) * (size_t)nwork);
    if (!M->use_rccl && e == hipSuccess) e = hipHostMalloc(&M->h_mine, sizeof(double) * part);
    if (!M->use_rccl && e == hipSuccess) e = hipHostMalloc(&M->h_parts, sizeof(double) * part * world);
  }
  dmalloc(&M->work, MG_MAX_PARTIALS + 8);
  dmalloc(&M->norm, 2);
  dmalloc(&M->dense, (size_t)M->dims[0] * M->dims[1] * M->dims[2]);
  if (e == hipSuccess) e = hipMalloc(&M->d_rank_of_brick, sizeof(int) * world);
  if (e == hipSuccess) e = hipMemcpy(M->d_rank_of_brick, M->rank_of_brick.data(), sizeof(int) * world, hipMemcpyHostToDevice);
  if (e != hipSuccess) return bail(failf(RAMSES_AMD_EHIP, "distributed multigrid: device allocation: %s", hipGetErrorString(e)));
  *out = M;
  return 0;
}

int ramses_amd_mgdist_destroy(ramses_amd_mgdist *M) {
  if (!M) return 0;
  for (Level &L : M->lev) free_level(L);
  free_level(M->rep_local);
  double **dev[] = {&M->rep_rhs, &M->rep_u1, &M->rep_work, &M->rep_parts, &M->rep_mine, &M->work, &M->norm, &M->dense};
  for (double **p : dev) if (*p) { (void)hipFree(*p); *p = nullptr; }
  if (M->d_rank_of_brick) (void)hipFree(M->d_rank_of_brick);
  if (M->d_order) (void)hipFree(M->d_order);
  if (M->ord_x) (void)hipFree(M->ord_x);
  if (M->ord_scratch) (void)hipFree(M->ord_scratch);
  if (M->h_mine) (void)hipHostFree(M->h_mine);
  if (M->h_parts) (void)hipHostFree(M->h_parts);
  delete M;
  return 0;
}

int ramses_amd_mgdist_info(const ramses_amd_mgdist *M, int *dims, int *coords, int *n_distributed_levels, int *first_replicated_level,
                           int *safe_mode, int64_t *exchanges) {
  if (!M) return failf(RAMSES_AMD_EINVAL, "NULL context");
  for (int d = 0; d < 3; d++) {
    if (dims) dims[d] = M->dims[d];
    if (coords) coords[d] = M->coords[d];
  }
  if (n_distributed_levels) *n_distributed_levels = M->level - M->lrep;
  if (first_replicated_level) *first_replicated_level = M->lrep;
  if (safe_mode) *safe_mode = M->safe_mode;
  if (exchanges) *exchanges = M->exchanges;
  return 0;
}

// The order in which the reference adds the squared residuals of this rank's cells (cmp_residual_norm2_fine: octant by octant
// over active(ilevel)%igrid): order[k] = index of the k-th cell of that loop in the rank's dense [nz][ny][nx] brick, a permutation
// of 0 .. N-1 (host array).  From then on the two norms of every iteration are strictly sequential sums in that order -- the
// reference's bits, and with them its convergence decision; n = 0 returns to the smoother's own reduction tree.
int ramses_amd_mgdist_set_order(ramses_amd_mgdist *M, const int *order, int64_t n) {
  if (!M) return failf(RAMSES_AMD_EINVAL, "NULL context");
  const long N = (long)M->dims[0] * M->dims[1] * M->dims[2];
  if (n == 0) { M->order_n = 0; return 0; }
  if (!order || n != N) return failf(RAMSES_AMD_EINVAL, "the order must name each of the brick's %ld cells once (got %ld entries)", N, (long)n);
  std::vector<unsigned char> seen((size_t)N, 0);
  for (long k = 0; k < N; k++) {
    if (order[k] < 0 || order[k] >= N || seen[(size_t)order[k]]) return failf(RAMSES_AMD_EINVAL, "the order is not a permutation of the brick's cells (entry %ld)", k);
    seen[(size_t)order[k]] = 1;
  }
  if (!M->d_order) HCHK(hipMalloc(&M->d_order, sizeof(int) * (size_t)N), "hipMalloc");
  if (!M->ord_x) HCHK(hipMalloc(&M->ord_x, sizeof(double) * (size_t)N), "hipMalloc");
  if (!M->ord_scratch) HCHK(hipMalloc(&M->ord_scratch, ramses_amd_ordered_sum_scratch(N)), "hipMalloc");
  HCHK(hipMemcpy(M->d_order, order, sizeof(int) * (size_t)N, hipMemcpyHostToDevice), "H2D order");
  M->order_n = N;
  return 0;
}

int ramses_amd_mgdist_set_safe_mode(ramses_amd_mgdist *M, int safe_mode) {
  if (!M) return failf(RAMSES_AMD_EINVAL, "NULL context");
  M->safe_mode = safe_mode ? 1 : 0;
  return 0;
}

// multigrid_fine from a zero first guess.  d_rho: this rank's dense [nz][ny][nx] brick of the density.
int ramses_amd_mgdist_solve(ramses_amd_mgdist *M, const double *d_rho, double rho_tot, double fourpi, double epsilon,
                            int *iters_out, double *err_out, void *stream) {
  if (!M || !d_rho) return failf(RAMSES_AMD_EINVAL, "NULL argument");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  Level &L = M->lev[M->level];
  double *phi = L.u[0], *phi2 = L.u[3], *f1 = L.u[2], *f2 = L.u[1];
  const long N = (long)L.n[0] * L.n[1] * L.n[2];
  M->phi_fresh = false;      // until this solve has gone through: an early return leaves a zeroed / half-smoothed field behind
  HCHK(hipMemsetAsync(phi, 0, sizeof(double) * L.cells, s), "memset");
  HCHK(mg_launch_rhs(d_rho, M->dense, N, fourpi, rho_tot, s), "mg rhs launch");
  RCHK(interior_copy(L, f2, M->dense, 0, s));
  RCHK(exchange(M, L, f2, s));
  int it = 0, safe = M->safe_mode;
  double err = 1.0, i_res_norm2 = 0.0, res_norm2 = 0.0;
  for (;;) {
    it++;
    if (it > 1) RCHK(exchange(M, L, phi, s));
    const bool ordered = M->order_n == N && M->d_order;
    HCHK(fused(M, L, phi, phi2, f2, f1, (it == 1 && !ordered) ? M->norm : nullptr, s), "mg fused smoother launch");
    if (it == 1 && ordered) RCHK(ordered_norm(M, L, f1, M->norm, s));
    if (it == 1) RCHK(allreduce_sum(M, M->norm, &i_res_norm2, s));
    if (M->level > 1) {
      RCHK(restrict_to(M, M->level, f1, s));
      RCHK(coarse_cycle(M, M->level - 1, safe, s));
      RCHK(interp_from(M, L, phi2, M->level - 1, s));
    }
    RCHK(exchange(M, L, phi2, s));
    // post-smoothing; only the norm of the residual is needed
    if (ordered) {
      // (the residual leaves the chip once more per iteration: f1 is free after the restriction)
      HCHK(fused(M, L, phi2, phi, f2, f1, nullptr, s), "mg fused smoother launch");
      RCHK(ordered_norm(M, L, f1, M->norm + 1, s));
    } else {
      HCHK(fused(M, L, phi2, phi, f2, nullptr, M->norm + 1, s), "mg fused smoother launch");
    }
    RCHK(allreduce_sum(M, M->norm + 1, &res_norm2, s));
    const double last_err = err;
    err = std::sqrt(res_norm2 / (i_res_norm2 + 1e-20 * (rho_tot * rho_tot)));
    if (err < epsilon || it >= MAXITER) break;
    if (err > last_err * SAFE_FACTOR && !safe) safe = 1;
  }
  M->safe_mode = safe;
  M->phi_fresh = true;
  M->last_iters = it; M->last_err = err;
  if (iters_out) *iters_out = it;
  if (err_out) *err_out = err;
  return 0;
}

// phi of this rank's brick as a dense [nz][ny][nx] array
int ramses_amd_mgdist_get_phi(ramses_amd_mgdist *M, double *d_phi, void *stream) {
  if (!M || !d_phi) return failf(RAMSES_AMD_EINVAL, "NULL argument");
  Level &L = M->lev[M->level];
  return interior_copy(L, L.u[0], d_phi, 1, reinterpret_cast<hipStream_t>(stream));
}

// first guess / restart: phi of this rank's brick from a dense array (ghosts are exchanged by the next call)
int ramses_amd_mgdist_set_phi(ramses_amd_mgdist *M, const double *d_phi, void *stream) {
  if (!M || !d_phi) return failf(RAMSES_AMD_EINVAL, "NULL argument");
  Level &L = M->lev[M->level];
  M->phi_fresh = false;
  RCHK(interior_copy(L, L.u[0], const_cast<double *>(d_phi), 0, reinterpret_cast<hipStream_t>(stream)));
  M->phi_fresh = true;       // the context holds the caller's potential: ramses_amd_mgdist_force may differentiate it
  return 0;
}

// force_fine: halo of phi, then gradient_phi into the dense [3][nz][ny][nx] array d_f
int ramses_amd_mgdist_force(ramses_amd_mgdist *M, double *d_f, void *stream) {
  if (!M || !d_f) return failf(RAMSES_AMD_EINVAL, "NULL argument");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  Level &L = M->lev[M->level];
  RCHK(exchange(M, L, L.u[0], s));
  return ramses_amd_gradient_phi_ghost(L.u[0], d_f, L.n[0], L.n[1], L.n[2], NG, std::ldexp(1.0, -M->level), s);
}

// ---------------------------------------------------------------------------
// The Fortran shim's side: the reference's own arrays (rho, phi as cell vectors; the rank's octs of the level).
// ---------------------------------------------------------------------------
namespace {
std::vector<int> g_h_order;          // the order list as the host built it last (uploaded again only when it changes)
int64_t g_phi_bytes = 0, g_rho_up_bytes = 0;   // what the Fortran entries moved over PCIe (ramses_amd_mgdist_traffic)
// integer position (in cells of the level) of an oct's first cell, from its centre xg (amr/amr_commons.f90:67-75)
inline bool oct_cell_origin(const double *xg, int64_t ngridmax, int ig, int n, int *o) {
  for (int d = 0; d < 3; d++) {
    const double c = xg[(size_t)d * ngridmax + (ig - 1)] * n;     // centre in cells: an odd multiple of 1 ... i.e. origin + 1
    const long v = (long)std::floor(c + 0.5) - 1;
    if (v < 0 || v + 1 >= n || (v & 1)) return false;
    o[d] = (int)v;
  }
  return true;
}
struct DevArr {
  void *p = nullptr;
  size_t cap = 0;
  hipError_t ensure(size_t bytes) {
    if (bytes <= cap) return hipSuccess;
    if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
    hipError_t e = hipMalloc(&p, bytes);
    if (e == hipSuccess) cap = bytes;
    return e;
  }
};
}  // namespace

// The deep-halo plan of one rank, host only (no device): what ramses_amd_mgdist_create builds for a level whose bricks have
// dims[3] cells -- for the CPU tests of the multi-rank protocol.  send_boxes / recv_boxes: 26 x (org x,y,z, ext x,y,z) in
// allocated coordinates (ng ghost layers), send_offs / recv_offs: positions in the message buffers (doubles); the
// messages: *npeer <= 26 peers, seg_peer[i] with seg_send_off/cnt[i] and seg_recv_off/cnt[i] (the caller's own rank appears
// where the box wraps onto itself).  Returns the buffer length in *total.
int ramses_amd_mgdist_plan(const int *pgrid, int rank, const int *rank_of_brick, const int *dims, int ng,
                           int *send_boxes, int64_t *send_offs, int *recv_boxes, int64_t *recv_offs,
                           int *npeer, int *seg_peer, int64_t *seg_send_off, int64_t *seg_send_cnt,
                           int64_t *seg_recv_off, int64_t *seg_recv_cnt, int64_t *total) {
  if (!pgrid || !dims || !send_boxes || !send_offs || !recv_boxes || !recv_offs || !npeer || !seg_peer || !seg_send_off ||
      !seg_send_cnt || !seg_recv_off || !seg_recv_cnt || !total || ng < 1)
    return failf(RAMSES_AMD_EINVAL, "bad argument");
  int world = 1;
  for (int d = 0; d < 3; d++) {
    if (pgrid[d] < 1 || dims[d] < ng) return failf(RAMSES_AMD_EINVAL, "bad rank grid / brick");
    world *= pgrid[d];
  }
  std::vector<int> rob(world);
  int mine = -1;
  for (int b = 0; b < world; b++) {
    rob[b] = rank_of_brick ? rank_of_brick[b] : b;
    if (rob[b] == rank) mine = b;
  }
  if (mine < 0) return failf(RAMSES_AMD_EINVAL, "rank %d owns no brick", rank);
  const int coords[3] = {mine % pgrid[0], (mine / pgrid[0]) % pgrid[1], mine / (pgrid[0] * pgrid[1])};
  HaloPlan P;
  RCHK(plan_regions(pgrid, coords, rob.data(), dims, ng, P));
  std::memcpy(send_boxes, P.boxes_s, sizeof(P.boxes_s)); std::memcpy(recv_boxes, P.boxes_r, sizeof(P.boxes_r));
  std::memcpy(send_offs, P.offs_s, sizeof(P.offs_s)); std::memcpy(recv_offs, P.offs_r, sizeof(P.offs_r));
  *npeer = (int)P.segs_s.size();
  for (size_t i = 0; i < P.segs_s.size(); i++) {
    seg_peer[i] = P.segs_s[i].peer;
    seg_send_off[i] = P.segs_s[i].off; seg_send_cnt[i] = P.segs_s[i].cnt;
    seg_recv_off[i] = P.segs_r[i].off; seg_recv_cnt[i] = P.segs_r[i].cnt;
  }
  *total = P.total;
  return 0;
}

// The box the rank's octs of the level fill: lo[3] (first cell) and dims[3] (cells), host only.  RAMSES_AMD_EUNSUPPORTED
// when they do not fill a box (the caller then keeps the multigrid of AMR levels).
int ramses_amd_mgdist_oct_box(int ilevel, int ngrid, const int *igrid, const double *xg, int64_t ngridmax, int *lo, int *dims) {
  if (!igrid || !xg || !lo || !dims || ilevel < 1 || ilevel > 11 || ngrid < 1) return failf(RAMSES_AMD_EINVAL, "bad argument");
  const int n = 1 << ilevel;
  int mn[3] = {n, n, n}, mx[3] = {-1, -1, -1};
  for (int g = 0; g < ngrid; g++) {
    int o[3];
    if (!oct_cell_origin(xg, ngridmax, igrid[g], n, o)) return failf(RAMSES_AMD_EINVAL, "oct %d of level %d does not sit on the level lattice", igrid[g], ilevel);
    for (int d = 0; d < 3; d++) { mn[d] = std::min(mn[d], o[d]); mx[d] = std::max(mx[d], o[d] + 2); }
  }
  long vol = 1;
  for (int d = 0; d < 3; d++) { lo[d] = mn[d]; dims[d] = mx[d] - mn[d]; vol *= dims[d]; }
  if (vol != (long)ngrid * 8) return failf(RAMSES_AMD_EUNSUPPORTED, "the rank's %d octs of level %d do not fill their bounding box (%d x %d x %d cells)", ngrid, ilevel, dims[0], dims[1], dims[2]);
  return 0;
}

// multigrid_fine(ilevel,icount) of the reference on its own arrays, several ranks: rho of the rank's octs is gathered
// into the brick, the distributed solve runs, phi of the rank's octs is written back (the caller refreshes the virtual
// octs with its own make_virtual_fine_dp, as the reference does at the end of multigrid_fine).  lo = the box origin
// ramses_amd_mgdist_oct_box returned.
int ramses_amd_mgdist_multigrid_f90(ramses_amd_mgdist *M, int ilevel, int ngrid, const int *igrid, const double *xg,
                                    int64_t ngridmax, int64_t ncoarse, const int *lo, const double *rho, double *phi,
                                    double rho_tot, double fourpi, double epsilon, int *safe_mode, int *iters, double *err) {
  if (!M || !igrid || !xg || !lo || !rho || !phi || !safe_mode) return failf(RAMSES_AMD_EINVAL, "NULL argument");
  if (ilevel != M->level) return failf(RAMSES_AMD_EINVAL, "context built for level %d, called for level %d", M->level, ilevel);
  const int n = 1 << ilevel;
  const long N = (long)M->dims[0] * M->dims[1] * M->dims[2];
  if ((long)ngrid * 8 != N) return failf(RAMSES_AMD_EINVAL, "the rank holds %d octs, its brick %ld cells", ngrid, N);
  for (int d = 0; d < 3; d++)
    if (lo[d] != M->coords[d] * M->dims[d]) return failf(RAMSES_AMD_EINVAL, "the rank's box does not start where its brick does");
  const long ncell = ncoarse + 8 * ngridmax;
  hipStream_t s = nullptr;
  static DevArr d_rho, d_phi;
  static std::vector<long> org;
  org.resize(ngrid);
  for (int g = 0; g < ngrid; g++) {
    int o[3];
    if (!oct_cell_origin(xg, ngridmax, igrid[g], n, o)) return failf(RAMSES_AMD_EINVAL, "oct %d of level %d does not sit on the level lattice", igrid[g], ilevel);
    for (int d = 0; d < 3; d++) {
      o[d] -= lo[d];
      if (o[d] < 0 || o[d] + 2 > M->dims[d]) return failf(RAMSES_AMD_EINVAL, "oct %d lies outside the rank's box", igrid[g]);
    }
    org[g] = o[0] + (long)M->dims[0] * (o[1] + (long)M->dims[1] * o[2]);
  }
  // Only the rank's own N cells cross PCIe (the cell vectors hold ncoarse + 8 ngridmax doubles, most of them other levels' or
  // nobody's): the brick of the density is assembled on the host from the rank's octs, the brick of the potential is spread
  // into the cell vector there -- two transfers of N doubles per solve instead of three of ncell.
  static std::vector<double> h_brick;
  h_brick.resize((size_t)N);
  const long py = M->dims[0], pz = (long)M->dims[0] * M->dims[1];
  for (int g = 0; g < ngrid; g++)
    for (int ind = 0; ind < 8; ind++)
      h_brick[(size_t)(org[g] + (ind & 1) + py * ((ind >> 1) & 1) + pz * ((ind >> 2) & 1))] = rho[ncoarse + (long)ind * ngridmax + (igrid[g] - 1)];
  HCHK(d_rho.ensure(sizeof(double) * N), "hipMalloc");
  HCHK(d_phi.ensure(sizeof(double) * N), "hipMalloc");
  HCHK(hipMemcpyAsync(d_rho.p, h_brick.data(), sizeof(double) * N, hipMemcpyHostToDevice, s), "H2D rho");
  g_rho_up_bytes += (int64_t)sizeof(double) * N;
  {
    // the convergence test sums the squared residuals as cmp_residual_norm2_fine does: octant by octant over this very list
    static std::vector<int> order;
    order.resize((size_t)N);
    for (int ind = 0; ind < 8; ind++)
      for (int g = 0; g < ngrid; g++)
        order[(size_t)ind * ngrid + g] = (int)(org[g] + (ind & 1) + py * ((ind >> 1) & 1) + pz * ((ind >> 2) & 1));
    RCHK(ramses_amd_mgdist_set_order(M, order.data(), N));
    g_h_order = order;
  }
  M->safe_mode = *safe_mode ? 1 : 0;
  RCHK(ramses_amd_mgdist_solve(M, reinterpret_cast<const double *>(d_rho.p), rho_tot, fourpi, epsilon, iters, err, s));
  *safe_mode = M->safe_mode;
  RCHK(ramses_amd_mgdist_get_phi(M, reinterpret_cast<double *>(d_phi.p), s));
  HCHK(hipMemcpyAsync(h_brick.data(), d_phi.p, sizeof(double) * N, hipMemcpyDeviceToHost, s), "D2H phi");
  g_phi_bytes += (int64_t)sizeof(double) * N;
  HCHK(hipStreamSynchronize(s), "sync");
  // phi of the rank's own cells into the cell vector (the other cells keep the host's values)
  for (int g = 0; g < ngrid; g++)
    for (int ind = 0; ind < 8; ind++)
      phi[ncoarse + (long)ind * ngridmax + (igrid[g] - 1)] = h_brick[(size_t)(org[g] + (ind & 1) + py * ((ind >> 1) & 1) + pz * ((ind >> 2) & 1))];
  HCHK(hipStreamSynchronize(s), "sync");
  return 0;
}

// force_fine(ilevel,icount) of the reference on its own arrays, several ranks, right after ramses_amd_mgdist_multigrid_f90 of
// the same level: the halo of phi and gradient_phi run on the brick the solve left on the device (poisson/force_fine.f90:
// 113-127,199-324), f(:,1:3) of the rank's own cells is written into the host cell vectors, and the two local diagnostics of
// :150-181 are returned -- diag[0] = sum over the leaf cells of fact*f**2 in the reference's order (batches of nvector octs,
// octant by octant, direction by direction), diag[1] = max |rho| -- for the caller's MPI_ALLREDUCEs.  The caller refreshes
// the virtual octs of f with its own make_virtual_fine_dp.
int ramses_amd_mgdist_force_f90(ramses_amd_mgdist *M, int ilevel, int ngrid, const int *igrid, const double *xg,
                                int64_t ngridmax, int64_t ncoarse, const int *lo, double *f, const double *rho, const int *son,
                                int nvector, double fact, double *diag) {
  if (!M || !igrid || !xg || !lo || !f || !rho || !son || !diag || nvector < 1) return failf(RAMSES_AMD_EINVAL, "bad argument");
  if (ilevel != M->level) return failf(RAMSES_AMD_EINVAL, "context built for level %d, called for level %d", M->level, ilevel);
  if (!M->phi_fresh) return failf(RAMSES_AMD_EINVAL, "force_fine: the context holds no potential (ramses_amd_mgdist_multigrid_f90 first)");
  for (int d = 0; d < 3; d++)
    if (lo[d] != M->coords[d] * M->dims[d])
      return failf(RAMSES_AMD_EINVAL, "force_fine: lo = (%d,%d,%d) is not the origin of this rank's brick (%d,%d,%d)", lo[0], lo[1], lo[2],
                   M->coords[0] * M->dims[0], M->coords[1] * M->dims[1], M->coords[2] * M->dims[2]);
  const int n = 1 << ilevel;
  const long N = (long)M->dims[0] * M->dims[1] * M->dims[2];
  if ((long)ngrid * 8 != N) return failf(RAMSES_AMD_EINVAL, "the rank holds %d octs, its brick %ld cells", ngrid, N);
  const long ncell = ncoarse + 8 * ngridmax;
  hipStream_t s = nullptr;
  static DevArr d_f;
  static std::vector<double> h_f;
  HCHK(d_f.ensure(sizeof(double) * 3 * N), "hipMalloc");
  h_f.resize(3 * (size_t)N);
  RCHK(ramses_amd_mgdist_force(M, reinterpret_cast<double *>(d_f.p), s));
  HCHK(hipMemcpyAsync(h_f.data(), d_f.p, sizeof(double) * 3 * N, hipMemcpyDeviceToHost, s), "D2H f");
  HCHK(hipStreamSynchronize(s), "sync");
  const long py = M->dims[0], pz = (long)M->dims[0] * M->dims[1];
  for (int g = 0; g < ngrid; g++) {
    int o[3];
    if (!oct_cell_origin(xg, ngridmax, igrid[g], n, o)) return failf(RAMSES_AMD_EINVAL, "oct %d of level %d does not sit on the level lattice", igrid[g], ilevel);
    for (int d = 0; d < 3; d++) {
      o[d] -= lo[d];
      if (o[d] < 0 || o[d] + 2 > M->dims[d]) return failf(RAMSES_AMD_EINVAL, "oct %d lies outside the rank's box", igrid[g]);
    }
    const long org = o[0] + py * o[1] + pz * o[2];
    for (int ind = 0; ind < 8; ind++) {
      const long icell = ncoarse + (long)ind * ngridmax + (igrid[g] - 1);
      const long b = org + (ind & 1) + py * ((ind >> 1) & 1) + pz * ((ind >> 2) & 1);
      for (int d = 0; d < 3; d++) f[(long)d * ncell + icell] = h_f[(size_t)d * N + b];
    }
  }
  double epot = 0.0, rmax = 0.0;
  for (int g0 = 0; g0 < ngrid; g0 += nvector) {
    const int nb = std::min(nvector, ngrid - g0);
    for (int ind = 0; ind < 8; ind++) {
      const long skip = ncoarse + (long)ind * ngridmax;
      for (int d = 0; d < 3; d++)
        for (int i = 0; i < nb; i++) {
          const long icell = skip + (igrid[g0 + i] - 1);
          if (son[icell] == 0) { const double v = f[(long)d * ncell + icell]; epot = epot + fact * (v * v); }
        }
      for (int i = 0; i < nb; i++) rmax = std::max(rmax, std::fabs(rho[skip + (igrid[g0 + i] - 1)]));
    }
  }
  diag[0] = epot;
  diag[1] = rmax;
  return 0;
}


// multigrid_fine of the same level for a run whose cell vectors are resident and that has ONE level (levelmin = nlevelmax: nobody
// interpolates from this potential, nobody refines it): the right-hand side comes from rho_fine's deposit on the device
// (ramses_amd_amrres_rho_to_brick through the order list), the potential stays on the rank's brick -- force_fine differentiates
// it there, ramses_amd_mgdist_fetch_phi_f90 brings it to the host vector for backup_poisson.  No level array crosses PCIe.
extern "C" int ramses_amd_amrres_rho_to_brick(int ngrid, const int *igrid, const int *d_order, double *d_brick);
extern "C" int ramses_amd_amrres_rho_absmax(int ngrid, const int *igrid, double *out);
namespace {
int build_order(ramses_amd_mgdist *M, int ilevel, int ngrid, const int *igrid, const double *xg, int64_t ngridmax, const int *lo) {
  const int n = 1 << ilevel;
  const long N = (long)ngrid * 8;
  const long py = M->dims[0], pz = (long)M->dims[0] * M->dims[1];
  static std::vector<int> order;
  order.resize((size_t)N);
  for (int g = 0; g < ngrid; g++) {
    int o[3];
    if (!oct_cell_origin(xg, ngridmax, igrid[g], n, o)) return failf(RAMSES_AMD_EINVAL, "oct %d of level %d does not sit on the level lattice", igrid[g], ilevel);
    for (int d = 0; d < 3; d++) {
      o[d] -= lo[d];
      if (o[d] < 0 || o[d] + 2 > M->dims[d]) return failf(RAMSES_AMD_EINVAL, "oct %d lies outside the rank's box", igrid[g]);
    }
    const long org = o[0] + py * o[1] + pz * o[2];
    for (int ind = 0; ind < 8; ind++) order[(size_t)ind * ngrid + g] = (int)(org + (ind & 1) + py * ((ind >> 1) & 1) + pz * ((ind >> 2) & 1));
  }
  if (M->order_n == N && M->d_order && g_h_order.size() == (size_t)N && memcmp(g_h_order.data(), order.data(), sizeof(int) * (size_t)N) == 0) return 0;
  RCHK(ramses_amd_mgdist_set_order(M, order.data(), N));
  g_h_order = order;
  return 0;
}
}  // namespace
int ramses_amd_mgdist_multigrid_resident_f90(ramses_amd_mgdist *M, int ilevel, int ngrid, const int *igrid, const double *xg, int64_t ngridmax,
                                             const int *lo, double rho_tot, double fourpi, double epsilon, int *safe_mode, int *iters, double *err) {
  if (!M || !igrid || !xg || !lo || !safe_mode) return failf(RAMSES_AMD_EINVAL, "NULL argument");
  if (ilevel != M->level) return failf(RAMSES_AMD_EINVAL, "context built for level %d, called for level %d", M->level, ilevel);
  const long N = (long)M->dims[0] * M->dims[1] * M->dims[2];
  if ((long)ngrid * 8 != N) return failf(RAMSES_AMD_EINVAL, "the rank holds %d octs, its brick %ld cells", ngrid, N);
  for (int d = 0; d < 3; d++)
    if (lo[d] != M->coords[d] * M->dims[d]) return failf(RAMSES_AMD_EINVAL, "the rank's box does not start where its brick does");
  hipStream_t s = nullptr;
  static DevArr d_rho;
  HCHK(d_rho.ensure(sizeof(double) * N), "hipMalloc");
  if (int rc = build_order(M, ilevel, ngrid, igrid, xg, ngridmax, lo)) return rc;
  RCHK(ramses_amd_amrres_rho_to_brick(ngrid, igrid, M->d_order, reinterpret_cast<double *>(d_rho.p)));
  M->safe_mode = *safe_mode ? 1 : 0;
  RCHK(ramses_amd_mgdist_solve(M, reinterpret_cast<const double *>(d_rho.p), rho_tot, fourpi, epsilon, iters, err, s));
  *safe_mode = M->safe_mode;
  HCHK(hipStreamSynchronize(s), "sync");
  return 0;
}
// phi of the rank's own cells from the brick into the host vector (backup_poisson; the caller refreshes the virtual octs)
int ramses_amd_mgdist_fetch_phi_f90(ramses_amd_mgdist *M, int ngrid, const int *igrid, int64_t ngridmax, int64_t ncoarse, double *phi) {
  if (!M || !igrid || !phi) return failf(RAMSES_AMD_EINVAL, "NULL argument");
  const long N = (long)M->dims[0] * M->dims[1] * M->dims[2];
  if ((long)ngrid * 8 != N || g_h_order.size() != (size_t)N) return failf(RAMSES_AMD_EINVAL, "fetch_phi: not the list of the last solve");
  hipStream_t s = nullptr;
  static DevArr d_phi;
  static std::vector<double> h;
  HCHK(d_phi.ensure(sizeof(double) * N), "hipMalloc");
  h.resize((size_t)N);
  RCHK(ramses_amd_mgdist_get_phi(M, reinterpret_cast<double *>(d_phi.p), s));
  HCHK(hipMemcpyAsync(h.data(), d_phi.p, sizeof(double) * N, hipMemcpyDeviceToHost, s), "D2H phi");
  HCHK(hipStreamSynchronize(s), "sync");
  g_phi_bytes += (int64_t)sizeof(double) * N;
  for (int ind = 0; ind < 8; ind++)
    for (int g = 0; g < ngrid; g++) phi[ncoarse + (long)ind * ngridmax + (igrid[g] - 1)] = h[(size_t)g_h_order[(size_t)ind * ngrid + g]];
  return 0;
}
int ramses_amd_mgdist_force_resident_dev_f90(ramses_amd_mgdist *M, int ilevel, int ngrid, const int *igrid, int64_t ngridmax, int64_t ncoarse,
                                             int nvector, double fact, double *diag);
// bytes of rho (host -> device) and phi (device -> host) the Fortran entries of the distributed solve moved since the start
int ramses_amd_mgdist_traffic(int64_t *out2) {
  if (!out2) return failf(RAMSES_AMD_EINVAL, "NULL argument");
  out2[0] = g_rho_up_bytes; out2[1] = g_phi_bytes;
  return 0;
}

// The same for a run whose cell vectors are resident (ramses_amd_amrres_*), on a level without finer octs (every cell a leaf:
// a uniform self-gravitating run): f goes from the brick into the resident acceleration ON THE DEVICE -- gathered into the
// packed order [3][8][ngrid] through the order list of the solve (ramses_amd_mgdist_set_order: octant by octant over igrid) and
// filed by ramses_amd_amrres_take_f_device -- and the potential-energy sum runs there too, term by term in the reference's
// order (batches of nvector octs, octant by octant, direction by direction: poisson/force_fine.f90:150-176) through the
// order-exact device sum.  max |rho| is read off the host's rho (no transfer).  Nothing of f crosses PCIe; the caller
// exchanges the virtual octs on the device (ramses_amd_amrres_halo_*, direction 7).
extern "C" int ramses_amd_amrres_take_f_device(int ngrid, const int *igrid, const double *d_fpack);
namespace {
__global__ __launch_bounds__(256) void mgdist_f_pack_kernel(const double *__restrict__ fb, const int *__restrict__ order, long N, double *__restrict__ fp) {
  const long k = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (k < N) {
    const long c = order[k];
#pragma unroll
    for (int d = 0; d < 3; d++) fp[(long)d * N + k] = fb[(long)d * N + c];
  }
}
// term t of the reference's loop: batch c of nvector octs, octant ind, direction d, oct i of the batch
__global__ __launch_bounds__(256) void mgdist_epot_terms_kernel(const double *__restrict__ fp, long ngrid, int nvector, double fact, double *__restrict__ x) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long N = 8 * ngrid;
  if (t >= 3 * N) return;
  const long c = t / (24L * nvector);
  const long g0 = c * nvector;
  const int nb = (int)(ngrid - g0 < nvector ? ngrid - g0 : nvector);
  const long r = t - c * 24L * nvector;
  const int ind = (int)(r / (3L * nb)), d = (int)((r / nb) % 3), i = (int)(r % nb);
  const double v = fp[(long)d * N + (long)ind * ngrid + g0 + i];
  x[t] = fact * (v * v);
}
}  // namespace
int ramses_amd_mgdist_force_resident_f90(ramses_amd_mgdist *M, int ilevel, int ngrid, const int *igrid, int64_t ngridmax, int64_t ncoarse,
                                         const double *rho, int nvector, double fact, double *diag) {
  // (rho = NULL: the deposit lives on the device -- ramses_amd_amrres_rho_keep -- and max |rho| is taken there)
  if (!M || !igrid || !diag || nvector < 1) return failf(RAMSES_AMD_EINVAL, "bad argument");
  if (ilevel != M->level) return failf(RAMSES_AMD_EINVAL, "context built for level %d, called for level %d", M->level, ilevel);
  if (!M->phi_fresh) return failf(RAMSES_AMD_EINVAL, "force_fine: the context holds no potential (ramses_amd_mgdist_multigrid_f90 first)");
  const long N = (long)M->dims[0] * M->dims[1] * M->dims[2];
  if ((long)ngrid * 8 != N) return failf(RAMSES_AMD_EINVAL, "the rank holds %d octs, its brick %ld cells", ngrid, N);
  if (M->order_n != N || !M->d_order) return failf(RAMSES_AMD_EINVAL, "force_fine (resident): the solve left no oct order (ramses_amd_mgdist_set_order)");
  hipStream_t s = nullptr;
  static DevArr d_f, d_fp, d_x, d_scr, d_out;
  HCHK(d_f.ensure(sizeof(double) * 3 * N), "hipMalloc"); HCHK(d_fp.ensure(sizeof(double) * 3 * N), "hipMalloc");
  HCHK(d_x.ensure(sizeof(double) * 3 * N), "hipMalloc"); HCHK(d_scr.ensure(ramses_amd_ordered_sum_scratch(3 * N)), "hipMalloc");
  HCHK(d_out.ensure(sizeof(double)), "hipMalloc");
  RCHK(ramses_amd_mgdist_force(M, reinterpret_cast<double *>(d_f.p), s));
  const dim3 b(256);
  hipLaunchKernelGGL(mgdist_f_pack_kernel, dim3((unsigned)((N + 255) / 256)), b, 0, s, reinterpret_cast<const double *>(d_f.p), M->d_order, N, reinterpret_cast<double *>(d_fp.p));
  hipLaunchKernelGGL(mgdist_epot_terms_kernel, dim3((unsigned)((3 * N + 255) / 256)), b, 0, s, reinterpret_cast<const double *>(d_fp.p), (long)ngrid, nvector, fact,
                     reinterpret_cast<double *>(d_x.p));
  HCHK(hipGetLastError(), "force_fine (resident) launch");
  RCHK(ramses_amd_ordered_sum_device(reinterpret_cast<const double *>(d_x.p), 3 * N, reinterpret_cast<double *>(d_out.p), d_scr.p, s));
  RCHK(ramses_amd_amrres_take_f_device(ngrid, igrid, reinterpret_cast<const double *>(d_fp.p)));
  double epot = 0.0;
  HCHK(hipMemcpyAsync(&epot, d_out.p, sizeof(double), hipMemcpyDeviceToHost, s), "D2H epot");
  double rmax = 0.0;
  if (rho) {
    for (int ind = 0; ind < 8; ind++) {
      const double *r = rho + ncoarse + (long)ind * ngridmax - 1;
      for (int g = 0; g < ngrid; g++) rmax = std::max(rmax, std::fabs(r[igrid[g]]));
    }
  } else {
    RCHK(ramses_amd_amrres_rho_absmax(ngrid, igrid, &rmax));
  }
  HCHK(hipStreamSynchronize(s), "sync");
  diag[0] = epot;
  diag[1] = rmax;
  return 0;
}

// (the Fortran binding of the case rho = NULL: the deposit lives on the device)
int ramses_amd_mgdist_force_resident_dev_f90(ramses_amd_mgdist *M, int ilevel, int ngrid, const int *igrid, int64_t ngridmax, int64_t ncoarse,
                                             int nvector, double fact, double *diag) {
  return ramses_amd_mgdist_force_resident_f90(M, ilevel, ngrid, igrid, ngridmax, ncoarse, nullptr, nvector, fact, diag);
}

}  // extern "C"
