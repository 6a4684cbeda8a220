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
//     strictly sequential sums over all cells in list order, reproduced bit for bit by a parallel scan
//     of parity functions (see multipole_kernel; one workgroup per component).
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

// multipole(d), d = 0..3: the SEQUENTIAL sum  s <- fl(s + a_i)  over the cells in the order (batch of nvector
// octs, ind_son, oct in the batch) of cic_from_multipole (:858-866), reproduced bit for bit by a parallel scan.
//
// While the running sum stays inside one binade [2^E, 2^(E+1)) its spacing is u = 2^(E-52), and adding a
// positive a_i is an INTEGER operation on S = s/u: S <- S + n_i + (rem_i > u/2) + (rem_i == u/2 and S + n_i odd),
// with n_i = floor(a_i/u) and rem_i the part of a_i below u (round to nearest, ties to even).  The increment
// depends on what came before only through the parity of S, so a run of elements is a function
// parity -> (increment for parity 0, increment for parity 1), and these functions compose associatively: a
// workgroup scans them like a prefix sum.  Elements that could leave the binade (checked with a margin of one
// unit per element) end the scan: the thread that owns them adds its elements with real floating-point adds,
// the exponent is re-read and the scan restarts behind them.  The sum crosses ~log2(N) binades in all.
// One workgroup per component; all additions are either exact integer sums or IEEE adds in the original order.
constexpr int MP_THREADS = 1024;
constexpr int MP_K = 8;                 // consecutive elements per thread and chunk
struct ParFn { long t0, t1; };          // increment of S for incoming parity 0 / 1
__device__ __forceinline__ ParFn par_compose(const ParFn &f, const ParFn &g) {   // f first, then g
  ParFn r;
  r.t0 = f.t0 + ((f.t0 & 1) ? g.t1 : g.t0);
  r.t1 = f.t1 + (((1 + f.t1) & 1) ? g.t1 : g.t0);
  // saturate (elements that leave the binade carry 2^53): anything beyond 2^53 only has to stay beyond it
  const long cap = 1L << 60;
  r.t0 = r.t0 < cap ? r.t0 : cap;
  r.t1 = r.t1 < cap ? r.t1 : cap;
  return r;
}
constexpr long MP_POISON = 1L << 53;

__global__ __launch_bounds__(MP_THREADS) void multipole_kernel(RhoArgs A, double *__restrict__ out) {
  __shared__ ParFn wavefn[MP_THREADS / 64];
  __shared__ double sh_s;
  __shared__ long sh_next;
  __shared__ int sh_cross;
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
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  if (tid == 0) {
    // the first elements cross a binade at almost every addition: plain sequential adds
    double s = 0.0;
    const long m = ncells < 64 ? ncells : 64;
    for (long p = 0; p < m; p++) s = s + operand(p);
    sh_s = s;
    sh_next = m;
  }
  __syncthreads();
  while (true) {
    const long i0 = sh_next;
    if (i0 >= ncells) break;
    const double s = sh_s;
    const long sbits = __double_as_longlong(s);
    const int sexp = (int)((sbits >> 52) & 0x7ff);
    // (s is a positive normal number here: sums of positive normal operands)
    const long S = (sbits & 0xfffffffffffffL) | (1L << 52);
    const int qu = sexp - 1075;                       // s = S * 2^qu
    // ---- this thread's MP_K elements as one parity function (+ an upper bound of their increments) ----
    double a[MP_K];
    const long base = i0 + (long)tid * MP_K;
#pragma unroll
    for (int e = 0; e < MP_K; e++) a[e] = (base + e) < ncells ? operand(base + e) : 0.0;
    ParFn f = {0, 0};
    long ub = 0;
#pragma unroll
    for (int e = 0; e < MP_K; e++) {
      const long ab = __double_as_longlong(a[e]);
      const int aexp = (int)((ab >> 52) & 0x7ff);
      const long m = aexp ? ((ab & 0xfffffffffffffL) | (1L << 52)) : (ab & 0xfffffffffffffL);
      const int q = (aexp ? aexp : 1) - 1075;         // a = m * 2^q
      const int k = qu - q;                           // a / u = m / 2^k
      long nint, inc_gt;
      int tie;
      if (m == 0) { nint = 0; inc_gt = 0; tie = 0; }
      else if (k <= 0) { nint = MP_POISON; inc_gt = 0; tie = 0; }       // a >= 2^E: leaves the binade
      else if (k >= 64) { nint = 0; inc_gt = 0; tie = 0; }
      else {
        nint = m >> k;
        const long rem = m & ((1L << k) - 1), half = 1L << (k - 1);
        inc_gt = rem > half ? 1 : 0;
        tie = rem == half ? 1 : 0;
      }
      const long bsum = nint + inc_gt;
      ParFn g;
      g.t0 = bsum + (tie ? (nint & 1) : 0);           // incoming parity 0: S + n odd  <=>  n odd
      g.t1 = bsum + (tie ? ((nint + 1) & 1) : 0);
      f = par_compose(f, g);
      ub += nint + 1;
      if (ub > MP_POISON) ub = MP_POISON;
    }
    // ---- inclusive scan of the functions over the workgroup (wave shuffles, then the 16 wave totals) ----
    ParFn inc = f;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      ParFn o;
      o.t0 = __shfl_up(inc.t0, off, 64);
      o.t1 = __shfl_up(inc.t1, off, 64);
      if (lane >= off) inc = par_compose(o, inc);
    }
    if (lane == 63) wavefn[wv] = inc;
    __syncthreads();
    ParFn pre = {0, 0};                                // everything before this thread's wave
    for (int w = 0; w < wv; w++) pre = par_compose(pre, wavefn[w]);
    ParFn excl;                                        // everything before this thread
    {
      ParFn o;
      o.t0 = __shfl_up(inc.t0, 1, 64);
      o.t1 = __shfl_up(inc.t1, 1, 64);
      if (lane == 0) { o.t0 = 0; o.t1 = 0; }
      excl = par_compose(pre, o);
    }
    const int p0 = (int)(S & 1);
    const long S_t = S + (p0 ? excl.t1 : excl.t0);     // S on entry of this thread's elements
    const bool unsafe = (ub >= MP_POISON) || (S_t + ub >= MP_POISON) || (S_t >= MP_POISON);
    if (tid == 0) sh_cross = MP_THREADS;
    __syncthreads();
    if (unsafe) atomicMin(&sh_cross, tid);
    __syncthreads();
    const int tc = sh_cross;
    if (tc == MP_THREADS) {
      if (tid == MP_THREADS - 1) {
        const ParFn tot = par_compose(excl, f);
        const long S_end = S + (p0 ? tot.t1 : tot.t0);
        sh_s = __builtin_ldexp((double)S_end, qu);
        sh_next = i0 + (long)MP_THREADS * MP_K;
      }
    } else if (tid == tc) {
      // everything before this thread stayed inside the binade; its own elements are added one by one
      double sc = __builtin_ldexp((double)S_t, qu);
#pragma unroll
      for (int e = 0; e < MP_K; e++) sc = sc + a[e];
      sh_s = sc;
      sh_next = base + MP_K;
    }
    __syncthreads();
  }
  if (tid == 0) out[comp] = sh_s;
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

#include "warm.hpp"
RAMSES_AMD_TU_WARM(rho_fine)
