// parity_scan.hpp -- strictly SEQUENTIAL floating-point sums  s <- fl(s + a_i), i = 0..N-1,  reproduced bit for bit IN
// PARALLEL (gfx950).  Used where the reference adds in a fixed order and the result feeds back into the run:
//   multipole(1:4) of rho_fine           pm/rho_fine.f90:858-866          (positive terms, 4 components; rho_fine.hip)
//   r.r, p.Ap, rhs_norm of phi_fine_cg   poisson/phi_fine_cg.f90:63-70,98-105,146-153   (signed terms; cg_amr.hip)
//
// While the running sum stays inside one binade [2^E, 2^(E+1)) its spacing is u = 2^(E-52), and adding a_i (either sign)
// is an INTEGER operation on S = s/u:  S <- S + n_i + (rem_i > 1/2) + (rem_i == 1/2 and S + n_i odd),  with
// n_i = floor(a_i/u) and rem_i = a_i/u - n_i in [0,1) (round to nearest, ties to even).  The increment depends on what
// came before only through the PARITY of S, so a run of elements is a function parity -> (increment for parity 0,
// increment for parity 1); these functions compose associatively and are scanned like a prefix sum.  The function is the
// true one only while S stays strictly inside (2^52, 2^53): every run carries bounds lo <= every partial increment <= hi
// (lo: the negative n_i added up, hi: the positive n_i + 1 added up), and a run whose bounds could leave the binade is
// not trusted -- its elements are added with real floating-point adds by the thread that owns them.
//
// The list is cut into segments of PS_SEG elements.
//   pass 0  ps_sum_kernel     plain sums per segment (any order; only used to PREDICT the binade)
//           ps_prefix_kernel  running sum at every segment start, approximately
//   pass 1  ps_fn_kernel      the parity function of every segment in the predicted binade, all segments at once
//   pass 2  ps_walk_kernel    one workgroup per component walks the segments in order with EXACT integer arithmetic:
//                             a segment whose prediction holds (the exact running sum has the predicted exponent and the
//                             segment's bounds keep it inside the binade) costs one table look-up; the others (the ~log2 N
//                             binade crossings, the first segment, a misprediction next to a power of two, a running sum
//                             that is not a positive normal number) take the workgroup scan, which adds the crossing
//                             elements with real floating-point adds.
// All additions are exact integer sums or IEEE adds in the original order.  Compile with -ffp-contract=off.
#pragma once
#include <hip/hip_runtime.h>

namespace ramses_amd {
namespace pscan {

constexpr int PS_THREADS = 1024;
constexpr int PS_K = 8;                 // consecutive elements per thread and chunk
constexpr long PS_SEG = (long)PS_THREADS * PS_K;
constexpr long PS_POISON = 1L << 53;    // S reaches the next binade
constexpr long PS_FLOOR = 1L << 52;     // S == 2^52: a negative term would round on the finer grid below
constexpr long PS_CAP = 1L << 60;

// increments of S for incoming parity 0 / 1 and bounds of every partial increment inside the run
struct ParFn { long t0, t1, lo, hi; };
__device__ __forceinline__ long ps_clamp(long v) { return v > PS_CAP ? PS_CAP : (v < -PS_CAP ? -PS_CAP : v); }
__device__ __forceinline__ ParFn par_identity() { return ParFn{0, 0, 0, 0}; }
__device__ __forceinline__ ParFn par_compose(const ParFn &f, const ParFn &g) {   // f first, then g
  ParFn r;
  r.t0 = ps_clamp(f.t0 + ((f.t0 & 1) ? g.t1 : g.t0));
  r.t1 = ps_clamp(f.t1 + (((1 + f.t1) & 1) ? g.t1 : g.t0));
  // the bounds add up: partial increments of (f, then g) lie in [min(f.lo, f.lo + g.lo), ...] -- both are sums of
  // one-signed parts, so f.lo + g.lo <= every partial <= f.hi + g.hi
  r.lo = ps_clamp(f.lo + g.lo);
  r.hi = ps_clamp(f.hi + g.hi);
  return r;
}
__device__ __forceinline__ bool par_safe(long S, const ParFn &f) {   // S on entry of the run
  return S > PS_FLOOR && S + f.lo > PS_FLOOR && S + f.hi < PS_POISON;
}

// one element as a parity function in the binade whose unit is 2^qu
__device__ __forceinline__ ParFn par_element(double a, int qu) {
  const long ab = __double_as_longlong(a);
  const bool neg = ab < 0;
  const int aexp = (int)((ab >> 52) & 0x7ff);
  const long m = aexp ? ((ab & 0xfffffffffffffL) | (1L << 52)) : (ab & 0xfffffffffffffL);
  const int q = (aexp ? aexp : 1) - 1075;         // |a| = m * 2^q
  const int k = qu - q;                           // |a| / u = m / 2^k
  ParFn g = {0, 0, 0, 0};
  if (m == 0) return g;
  if (k <= 0 || aexp == 0x7ff) {                  // |a| >= 2^E (or inf / nan): leaves the binade
    g.lo = -PS_CAP; g.hi = PS_CAP;
    return g;
  }
  if (k >= 64) {                                  // |a| < u / 2^11: s + a rounds back to s (S > 2^52 is checked by the caller)
    if (neg) g.lo = -1; else g.hi = 1;
    return g;
  }
  const long nn = m >> k, rem = m & ((1L << k) - 1), half = 1L << (k - 1);
  long n;
  bool gt, tie;
  if (!neg) { n = nn; gt = rem > half; tie = rem == half; }
  else if (rem == 0) { n = -nn; gt = false; tie = false; }
  else { n = -nn - 1; gt = rem < half; tie = rem == half; }          // a/u = n + (2^k - rem)/2^k
  const long bsum = n + (gt ? 1 : 0);
  g.t0 = bsum + (tie ? (n & 1) : 0);              // incoming parity 0: S + n odd  <=>  n odd
  g.t1 = bsum + (tie ? ((n + 1) & 1) : 0);
  if (n < 0) g.lo = n;
  if (n + 1 > 0) g.hi = n + 1;
  return g;
}

// a running sum the scan can work on: a normal number of either sign (a negative one is scanned as its mirror image:
// fl(s + a) = -fl((-s) + (-a)))
__device__ __forceinline__ bool ps_normal(double s) { const double m = __builtin_fabs(s); return m >= 2.3e-308 && m < 1.7e308; }
constexpr int PS_NOPRED = (int)0x80000000;
__device__ __forceinline__ int ps_code(int qu, bool neg) { return qu * 2 + (neg ? 1 : 0); }

// pass 0: plain per-segment sums (prediction only).  Src: count(), operator()(comp, position)
template <class Src, int NC>
__global__ __launch_bounds__(PS_THREADS) void ps_sum_kernel(Src S, long nseg, double *__restrict__ segsum) {
  __shared__ double wsum[PS_THREADS / 64];
  const long seg = blockIdx.x;
  const long ncells = S.count();
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  for (int comp = 0; comp < NC; comp++) {
    double a = 0.0;
    const long base = seg * PS_SEG + (long)tid * PS_K;
#pragma unroll
    for (int e = 0; e < PS_K; e++) a += (base + e) < ncells ? S(comp, base + e) : 0.0;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) a += __shfl_down(a, off, 64);
    if (lane == 0) wsum[wv] = a;
    __syncthreads();
    if (tid == 0) {
      double t = 0.0;
      for (int w = 0; w < PS_THREADS / 64; w++) t += wsum[w];
      segsum[(long)comp * nseg + seg] = t;
    }
    __syncthreads();
  }
}
// running sum at the start of every segment (exclusive prefix of the segment sums), one workgroup per component
template <int NC>
__global__ __launch_bounds__(PS_THREADS) void ps_prefix_kernel(long nseg, const double *__restrict__ segsum, double *__restrict__ pre) {
  __shared__ double part[PS_THREADS];
  const int comp = blockIdx.x, tid = threadIdx.x;
  const long per = (nseg + PS_THREADS - 1) / PS_THREADS;
  const long lo = (long)tid * per, hi = lo + per < nseg ? lo + per : nseg;
  double t = 0.0;
  for (long k = lo; k < hi; k++) t += segsum[(long)comp * nseg + k];
  part[tid] = t;
  __syncthreads();
  if (tid == 0) {
    double run = 0.0;
    for (int k = 0; k < PS_THREADS; k++) { const double v = part[k]; part[k] = run; run += v; }
  }
  __syncthreads();
  double run = part[tid];
  for (long k = lo; k < hi; k++) { pre[(long)comp * nseg + k] = run; run += segsum[(long)comp * nseg + k]; }
}
// pass 1: the parity function of every segment in the binade its (approximate) starting sum predicts;
// qupred = unit exponent and sign used (PS_NOPRED: no prediction -- the walk takes the slow path there)
template <class Src, int NC>
__global__ __launch_bounds__(PS_THREADS) void ps_fn_kernel(Src S, long nseg, const double *__restrict__ pre, ParFn *__restrict__ fn,
                                                           int *__restrict__ qupred) {
  __shared__ ParFn wavefn[PS_THREADS / 64];
  const long seg = blockIdx.x;
  const long ncells = S.count();
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  for (int comp = 0; comp < NC; comp++) {
    const double s0 = pre[(long)comp * nseg + seg];
    const long sbits = __double_as_longlong(s0);
    const int sexp = (int)((sbits >> 52) & 0x7ff);
    const bool usable = seg > 0 && ps_normal(s0);
    const bool neg = s0 < 0.0;
    const int qu = sexp - 1075;
    ParFn f = par_identity();
    if (usable) {
      const long base = seg * PS_SEG + (long)tid * PS_K;
#pragma unroll
      for (int e = 0; e < PS_K; e++) {
        const double a = (base + e) < ncells ? S(comp, base + e) : 0.0;
        f = par_compose(f, par_element(neg ? -a : a, qu));
      }
      // ordered reduction (composition is associative, not commutative): lanes, then waves
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        ParFn o;
        o.t0 = __shfl_down(f.t0, off, 64);
        o.t1 = __shfl_down(f.t1, off, 64);
        o.lo = __shfl_down(f.lo, off, 64);
        o.hi = __shfl_down(f.hi, off, 64);
        if ((lane & (2 * off - 1)) == 0) f = par_compose(f, o);
      }
    }
    if (lane == 0) wavefn[wv] = f;
    __syncthreads();
    if (tid == 0) {
      ParFn tot = wavefn[0];
      for (int w = 1; w < PS_THREADS / 64; w++) tot = par_compose(tot, wavefn[w]);
      fn[(long)comp * nseg + seg] = tot;
      qupred[(long)comp * nseg + seg] = usable ? ps_code(qu, neg) : PS_NOPRED;
    }
    __syncthreads();
  }
}

// pass 2: the exact walk.  ps_slow_range adds the elements [i0, lim) to sh_s with the workgroup scan (what leaves the
// binade is added with real floating-point adds by the thread that owns it).
template <class Src>
__device__ void ps_slow_range(const Src &S, int comp, long i0_in, long lim, double &sh_s, long &sh_next, long &sh_first, int &sh_cross, ParFn *wavefn) {
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  if (tid == 0) {
    double s = sh_s;
    long p = i0_in;
    if (i0_in == 0) {
      // the first elements cross a binade at almost every addition: plain sequential adds
      const long m = lim < 64 ? lim : 64;
      for (; p < m; p++) s = s + S(comp, p);
    }
    sh_s = s;
    sh_next = p;
  }
  __syncthreads();
  while (true) {
    const bool zero = sh_s == 0.0;
    __syncthreads();                                   // (everybody has read sh_s before thread 0 writes it again)
    if (zero) {
      // nothing added yet (or everything cancelled): zeros change nothing -- find the first non-zero element in parallel
      if (tid == 0) sh_first = lim;
      __syncthreads();
      const long b0 = sh_next + (long)tid * PS_K;
      for (int e = 0; e < PS_K; e++) {
        if (b0 + e < lim && S(comp, b0 + e) != 0.0) { atomicMin(reinterpret_cast<unsigned long long *>(&sh_first), (unsigned long long)(b0 + e)); break; }
      }
      __syncthreads();
      if (tid == 0) { const long end = sh_next + PS_SEG < lim ? sh_next + PS_SEG : lim; sh_next = sh_first < end ? sh_first : end; }
      __syncthreads();
    }
    if (tid == 0) {
      // a sum that is zero, subnormal or not finite cannot be scanned: add one by one until it is a normal number
      double s = sh_s;
      long p = sh_next;
      while (p < lim && !ps_normal(s) && (s != 0.0 || p == sh_next)) { s = s + S(comp, p); p++; }
      sh_s = s;
      sh_next = p;
    }
    __syncthreads();
    const long i0 = sh_next;
    if (i0 >= lim) break;
    if (!ps_normal(sh_s)) continue;                    // (uniform: back to the zero skip / the one-by-one adds)
    const bool neg = sh_s < 0.0;
    const double s = __builtin_fabs(sh_s);
    const long sbits = __double_as_longlong(s);
    const int sexp = (int)((sbits >> 52) & 0x7ff);
    const long Sx = (sbits & 0xfffffffffffffL) | (1L << 52);
    const int qu = sexp - 1075;                       // |s| = Sx * 2^qu
    // ---- this thread's PS_K elements as one parity function ----
    double a[PS_K];
    const long base = i0 + (long)tid * PS_K;
#pragma unroll
    for (int e = 0; e < PS_K; e++) { const double v = (base + e) < lim ? S(comp, base + e) : 0.0; a[e] = neg ? -v : v; }
    ParFn f = par_identity();
#pragma unroll
    for (int e = 0; e < PS_K; e++) f = par_compose(f, par_element(a[e], qu));
    // ---- inclusive scan of the functions over the workgroup (wave shuffles, then the 16 wave totals) ----
    ParFn inc = f;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      ParFn o;
      o.t0 = __shfl_up(inc.t0, off, 64);
      o.t1 = __shfl_up(inc.t1, off, 64);
      o.lo = __shfl_up(inc.lo, off, 64);
      o.hi = __shfl_up(inc.hi, off, 64);
      if (lane >= off) inc = par_compose(o, inc);
    }
    if (lane == 63) wavefn[wv] = inc;
    __syncthreads();
    ParFn prew = par_identity();                       // everything before this thread's wave
    for (int w = 0; w < wv; w++) prew = par_compose(prew, wavefn[w]);
    ParFn excl;                                        // everything before this thread
    {
      ParFn o;
      o.t0 = __shfl_up(inc.t0, 1, 64);
      o.t1 = __shfl_up(inc.t1, 1, 64);
      o.lo = __shfl_up(inc.lo, 1, 64);
      o.hi = __shfl_up(inc.hi, 1, 64);
      if (lane == 0) o = par_identity();
      excl = par_compose(prew, o);
    }
    const int p0 = (int)(Sx & 1);
    const long S_t = ps_clamp(Sx + (p0 ? excl.t1 : excl.t0));    // S on entry of this thread's elements (if all before it is safe)
    const bool unsafe = !par_safe(S_t, f);
    if (tid == 0) sh_cross = PS_THREADS;
    __syncthreads();
    if (unsafe) atomicMin(&sh_cross, tid);
    __syncthreads();
    const int tc = sh_cross;
    if (tc == PS_THREADS) {
      if (tid == PS_THREADS - 1) {
        const ParFn tot = par_compose(excl, f);
        const long S_end = Sx + (p0 ? tot.t1 : tot.t0);
        const double se = __builtin_ldexp((double)S_end, qu);
        sh_s = neg ? -se : se;
        sh_next = i0 + PS_SEG;
      }
    } else if (tid == tc) {
      // everything before this thread stayed inside the binade; its own elements are added one by one
      double sc = __builtin_ldexp((double)S_t, qu);
#pragma unroll
      for (int e = 0; e < PS_K; e++) sc = sc + a[e];
      sh_s = neg ? -sc : sc;
      sh_next = base + PS_K;
    }
    __syncthreads();
  }
}

constexpr int PS_WALK_TILE = 1024;      // segments whose functions sit in LDS at a time
// out[comp] = the sum; `fin`, if given, is called by one thread with (comp, sum) after the store (the caller's epilogue)
template <class Src, class Fin>
__global__ __launch_bounds__(PS_THREADS) void ps_walk_kernel(Src S, long nseg, const ParFn *__restrict__ fn, const int *__restrict__ qupred,
                                                             double *__restrict__ out, int *__restrict__ nslow, Fin fin) {
  __shared__ ParFn wavefn[PS_THREADS / 64];
  __shared__ ParFn tfn[PS_WALK_TILE];
  __shared__ int tqu[PS_WALK_TILE];
  __shared__ double sh_s;
  __shared__ long sh_next, sh_seg, sh_first;
  __shared__ int sh_cross, sh_slow;
  const int comp = blockIdx.x, tid = threadIdx.x;
  const long ncells = S.count();
  if (tid == 0) { sh_s = 0.0; sh_seg = 0; sh_slow = 0; }
  __syncthreads();
  int slow_count = 0;
  for (long t0 = 0; t0 < nseg; t0 += PS_WALK_TILE) {
    const int nt = (int)(nseg - t0 < PS_WALK_TILE ? nseg - t0 : PS_WALK_TILE);
    for (int k = tid; k < nt; k += PS_THREADS) { tfn[k] = fn[(long)comp * nseg + t0 + k]; tqu[k] = qupred[(long)comp * nseg + t0 + k]; }
    __syncthreads();
    while (true) {
      if (tid == 0) {
        long seg = sh_seg;
        double s = sh_s;
        int slow = 0;
        while (seg < t0 + nt) {
          const bool neg = s < 0.0;
          const long sbits = __double_as_longlong(__builtin_fabs(s));
          const int sexp = (int)((sbits >> 52) & 0x7ff);
          const long Sx = (sbits & 0xfffffffffffffL) | (1L << 52);
          const int k = (int)(seg - t0);
          if (!ps_normal(s) || ps_code(sexp - 1075, neg) != tqu[k] || !par_safe(Sx, tfn[k])) { slow = 1; break; }
          const long inc = (Sx & 1) ? tfn[k].t1 : tfn[k].t0;
          const double sn = __builtin_ldexp((double)(Sx + inc), sexp - 1075);
          s = neg ? -sn : sn;
          seg++;
        }
        sh_s = s; sh_seg = seg; sh_slow = slow;
      }
      __syncthreads();
      if (!sh_slow) break;                             // the tile is done
      const long seg = sh_seg;
      const long lim = (seg + 1) * PS_SEG < ncells ? (seg + 1) * PS_SEG : ncells;
      ps_slow_range(S, comp, seg * PS_SEG, lim, sh_s, sh_next, sh_first, sh_cross, wavefn);
      slow_count++;
      if (tid == 0) sh_seg = seg + 1;
      __syncthreads();
    }
  }
  if (tid == 0) {
    out[comp] = sh_s;
    if (nslow) nslow[comp] = slow_count;
    fin(comp, sh_s);
  }
}

struct NoFin { __device__ void operator()(int, double) const {} };

inline size_t scratch_bytes(long n, int nc) {
  const long nseg = (n + PS_SEG - 1) / PS_SEG + 1;
  return (sizeof(double) * 2 + sizeof(ParFn) + sizeof(int)) * (size_t)nc * (size_t)nseg + 64;
}

// out[0..NC-1] = the sequential sums of the NC components of S (n elements each); 4 launches on stream s
template <class Src, int NC, class Fin = NoFin>
inline hipError_t launch(const Src &S, long n, double *out, void *scratch, hipStream_t s, Fin fin = Fin()) {
  const long nseg = (n + PS_SEG - 1) / PS_SEG;
  if (nseg < 1) {
    // (no element: the sum is zero; the epilogue still runs)
    hipError_t e = hipMemsetAsync(out, 0, sizeof(double) * NC, s);
    if (e != hipSuccess) return e;
  }
  const long ns = nseg < 1 ? 1 : nseg;
  // scratch: segsum[NC*nseg] doubles, pre[NC*nseg] doubles, fn[NC*nseg] ParFn, qupred[NC*nseg] ints, nslow[NC]
  char *w = reinterpret_cast<char *>(scratch);
  double *segsum = reinterpret_cast<double *>(w); w += sizeof(double) * NC * ns;
  double *pre = reinterpret_cast<double *>(w); w += sizeof(double) * NC * ns;
  ParFn *fn = reinterpret_cast<ParFn *>(w); w += sizeof(ParFn) * NC * ns;
  int *qupred = reinterpret_cast<int *>(w); w += sizeof(int) * NC * ns;
  int *nslow = reinterpret_cast<int *>(w);
  if (nseg >= 1) {
    hipLaunchKernelGGL((ps_sum_kernel<Src, NC>), dim3((unsigned)nseg), dim3(PS_THREADS), 0, s, S, nseg, segsum);
    hipLaunchKernelGGL((ps_prefix_kernel<NC>), dim3(NC), dim3(PS_THREADS), 0, s, nseg, segsum, pre);
    hipLaunchKernelGGL((ps_fn_kernel<Src, NC>), dim3((unsigned)nseg), dim3(PS_THREADS), 0, s, S, nseg, pre, fn, qupred);
  }
  hipLaunchKernelGGL((ps_walk_kernel<Src, Fin>), dim3(NC), dim3(PS_THREADS), 0, s, S, nseg < 0 ? 0 : nseg, fn, qupred, out, nslow, fin);
  return hipGetLastError();
}

}  // namespace pscan
}  // namespace ramses_amd
