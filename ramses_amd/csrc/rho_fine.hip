// rho_fine.hip -- rho_fine's hydro deposit on the device-resident level brick
// (reference: pm/rho_fine.f90: multipole_fine :666-820, cic_from_multipole :825-891, cic_cell :896-1142;
// SURVEY.md 8f rank 2).  Fully refined periodic level, nx=ny=nz=1, no particles.
//
// The reference turns every leaf cell into a pseudo-particle (mass m = max(rho,smallr)*vol at the
// centre of mass (m*x)/m -- the cell centre up to rounding) and CIC-deposits it: a target cell
// receives up to 27 contributions, and it adds them in the order of the reference's loop nest
// (batch of nvector octs, ind_son, CIC corner, oct in the batch).  Floating-point addition is not
// associative, so the device does the same sums in the same order:
//   * deposit: one thread per TARGET cell gathers the contributions of the 3^3 cells around it,
//     tags each with its position in that loop nest, and adds them in tag order (the order-tagged
//     gather validated on the CPU by oracle/rho_fine_oracle.c: ora_rho_deposit_gather);
//   * multipole(1:4) (rho_tot = multipole(1)/scale^3 enters the right-hand side of the Poisson solve):
//     strictly sequential sums over all cells in list order -- one lane adds, the other threads of
//     the workgroup stream the operands into LDS ahead of it (one workgroup per component).
// This unit is compiled with -ffp-contract=off: IEEE operations in the reference's order.
#include <hip/hip_runtime.h>

#include "rho_args.hpp"

namespace ramses_amd {

// octidx[oct position] = index of the oct in the level's list (active(ilevel)%igrid order)
__global__ __launch_bounds__(256) void oct_index_kernel(const long *__restrict__ octorg, int ngrid, int n,
                                                         int *__restrict__ octidx) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= ngrid) return;
  const long o = octorg[g];
  const int x = (int)(o % n), y = (int)((o / n) % n), z = (int)(o / ((long)n * n));
  const int no = n >> 1;
  octidx[(x >> 1) + no * ((y >> 1) + no * (z >> 1))] = g;
}

constexpr int DEP_THREADS = 128;

__global__ __launch_bounds__(DEP_THREADS) void rho_deposit_kernel(RhoArgs A) {
  // candidates of this thread: [27][DEP_THREADS] keys and values in LDS (27*128*16 B = 55 KB)
  __shared__ long skey[27][DEP_THREADS];
  __shared__ double sval[27][DEP_THREADS];
  const int n = A.n, no = n >> 1;
  const long N = (long)n * n * n;
  const long t = (long)blockIdx.x * DEP_THREADS + threadIdx.x;
  if (t >= N) return;                      // (no barrier below: each thread uses its own LDS column)
  const int tx = (int)(t % n), ty = (int)((t / n) % n), tz = (int)(t / ((long)n * n));
  const int tc[3] = {tx, ty, tz};
  const double dx = A.dx, scale = A.scale;
  int cnt = 0;
  for (int oz = -1; oz <= 1; oz++)
    for (int oy = -1; oy <= 1; oy++)
      for (int ox = -1; ox <= 1; ox++) {
        const int o[3] = {ox, oy, oz};
        int sc[3];
#pragma unroll
        for (int d = 0; d < 3; d++) {
          int s = tc[d] - o[d];
          s = s < 0 ? s + n : (s >= n ? s - n : s);
          sc[d] = s;
        }
        const long scell = sc[0] + (long)n * (sc[1] + (long)n * sc[2]);
        // multipole_fine (:747-784): mass and mass*position of the source cell
        const double mm = __builtin_fmax(A.dens[scell], A.smallr) * A.vol_loc;
        double w[3];
        int b[3];
        bool hit = true;
#pragma unroll
        for (int d = 0; d < 3; d++) {
          const int bit = sc[d] & 1;
          const double xg = (double)(2 * (sc[d] >> 1) + 1) * dx;          // oct centre (exact)
          const double xc = ((double)bit - 0.5) * dx;
          const double xx = (xg + xc - 0.0) * scale;
          const double mx = mm * xx;
          // cic_from_multipole / cic_cell: centre of mass in cells of the 6^3 block around the source oct
          double x = mx / mm;
          x = x / scale + 0.0;
          x = x - (xg - 3.0 * dx);
          x = x / dx;
          double dd = x + 0.5;
          const int id = (int)dd;
          dd = dd - id;
          const double dg = 1.0 - dd;
          const int ig = id - 1;
          const int kt = 2 + bit + o[d];                                   // block coordinate of the target cell
          if (kt == ig) { b[d] = 0; w[d] = dg; }
          else if (kt == id) { b[d] = 1; w[d] = dd; }
          else hit = false;
        }
        if (!hit) continue;
        const double vol = w[0] * w[1] * w[2];
        const double vol2 = mm * vol / A.vol_loc;
        if (vol2 == 0.0) continue;                                         // + 0.0 leaves the (non-negative) sum unchanged
        const int ind_son = (sc[0] & 1) + 2 * (sc[1] & 1) + 4 * (sc[2] & 1);
        const int ind = b[0] + 2 * b[1] + 4 * b[2];
        const long i = A.octidx[(sc[0] >> 1) + no * ((sc[1] >> 1) + no * (sc[2] >> 1))];
        const long batch = i / A.nvector, j = i % A.nvector;
        skey[cnt][threadIdx.x] = ((batch * 8 + ind_son) * 8 + ind) * A.nvector + j;
        sval[cnt][threadIdx.x] = vol2;
        cnt++;
      }
  // add in tag order (selection: cnt is 1..8 in practice)
  double r = 0.0;
  long last = -1;
  for (int k = 0; k < cnt; k++) {
    long best = 0x7fffffffffffffffL;
    double v = 0.0;
    for (int c = 0; c < cnt; c++) {
      const long key = skey[c][threadIdx.x];
      if (key > last && key < best) { best = key; v = sval[c][threadIdx.x]; }
    }
    r = r + v;
    last = best;
  }
  A.rho[t] = r;
}

// multipole(d), d = 0..3: sequential sum over the cells in the order (batch of nvector octs, ind_son, oct in
// the batch) of cic_from_multipole (:858-866).  One workgroup per component; thread 0 adds, threads 1.. stage
// the next chunk of operands in LDS.
constexpr int MP_THREADS = 256;
constexpr int MP_CHUNK = 2040;     // 8 operands per staging thread
__global__ __launch_bounds__(MP_THREADS) void multipole_kernel(RhoArgs A, double *__restrict__ out) {
  __shared__ double buf[2][MP_CHUNK];
  const int comp = blockIdx.x;       // 0: mass, 1..3: mass * position
  const int n = A.n;
  const long ncells = (long)A.ngrid * 8;
  const int nv = A.nvector;
  const long nfull = A.ngrid / nv;                 // full batches
  const long full_cells = nfull * nv * 8;
  const int nlast = A.ngrid - (int)(nfull * nv);   // octs of the last (partial) batch
  auto operand = [&](long p) -> double {
    long batch, r;
    int np;
    if (p < full_cells) { batch = p / (8L * nv); r = p % (8L * nv); np = nv; }
    else { batch = nfull; r = p - full_cells; np = nlast; }
    const int ind_son = (int)(r / np), j = (int)(r % np);
    const long org = A.octorg[batch * nv + j];
    const int bx = ind_son & 1, by = (ind_son >> 1) & 1, bz = ind_son >> 2;
    const long cell = org + bx + (long)n * (by + (long)n * bz);
    const double mm = __builtin_fmax(A.dens[cell], A.smallr) * A.vol_loc;
    if (comp == 0) return mm;
    const int d = comp - 1;
    const int c0 = d == 0 ? (int)(org % n) : (d == 1 ? (int)((org / n) % n) : (int)(org / ((long)n * n)));
    const int bit = d == 0 ? bx : (d == 1 ? by : bz);
    const double xg = (double)(c0 + 1) * A.dx;                      // c0 even: oct centre = (c0 + 1) dx
    const double xc = ((double)bit - 0.5) * A.dx;
    const double xx = (xg + xc - 0.0) * A.scale;
    return mm * xx;
  };
  const long nchunk = (ncells + MP_CHUNK - 1) / MP_CHUNK;
  // first chunk: staged by everybody but thread 0 as well (uniform code)
  auto stage = [&](long k) {
    if (threadIdx.x == 0) return;
    const long base = k * MP_CHUNK;
    for (int e = threadIdx.x - 1; e < MP_CHUNK; e += MP_THREADS - 1) {
      const long p = base + e;
      buf[k & 1][e] = p < ncells ? operand(p) : 0.0;
    }
  };
  stage(0);
  __syncthreads();
  double s = 0.0;
  for (long k = 0; k < nchunk; k++) {
    if (threadIdx.x == 0) {
      const long left = ncells - k * MP_CHUNK;
      const int m = left < MP_CHUNK ? (int)left : MP_CHUNK;
      const double *b = buf[k & 1];
      int e = 0;
      for (; e + 8 <= m; e += 8) {
        const double a0 = b[e], a1 = b[e + 1], a2 = b[e + 2], a3 = b[e + 3], a4 = b[e + 4], a5 = b[e + 5], a6 = b[e + 6], a7 = b[e + 7];
        s = s + a0; s = s + a1; s = s + a2; s = s + a3; s = s + a4; s = s + a5; s = s + a6; s = s + a7;
      }
      for (; e < m; e++) s = s + b[e];
    } else if (k + 1 < nchunk) {
      stage(k + 1);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) out[comp] = s;
}

hipError_t launch_oct_index(const long *octorg, int ngrid, int n, int *octidx, hipStream_t s) {
  hipLaunchKernelGGL(oct_index_kernel, dim3((ngrid + 255) / 256), dim3(256), 0, s, octorg, ngrid, n, octidx);
  return hipGetLastError();
}
hipError_t launch_rho_deposit(const RhoArgs &A, hipStream_t s) {
  const long N = (long)A.n * A.n * A.n;
  hipLaunchKernelGGL(rho_deposit_kernel, dim3((unsigned)((N + DEP_THREADS - 1) / DEP_THREADS)), dim3(DEP_THREADS), 0, s, A);
  return hipGetLastError();
}
hipError_t launch_multipole(const RhoArgs &A, double *out4, hipStream_t s) {
  hipLaunchKernelGGL(multipole_kernel, dim3(4), dim3(MP_THREADS), 0, s, A, out4);
  return hipGetLastError();
}

}  // namespace ramses_amd
