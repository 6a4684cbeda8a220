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

#include "parity_scan.hpp"
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
// octs, ind_son, oct in the batch) of cic_from_multipole (:858-866), reproduced bit for bit IN PARALLEL by the scan of
// parity functions of parity_scan.hpp (round 2: one workgroup per component, 32 ms at 256^3; round 3: segments scanned
// by many workgroups and walked with exact integer arithmetic, well under 1 ms).
// Where the operands come from.  Brick: the uniform resident level (mass and mass * position computed from the
// density brick).  Vec: the multipoles of the cells of an AMR level, already in a (4, ncell) cell vector.
struct MpBrickSrc {
  RhoArgs A;
  __device__ long count() const { return (long)A.ngrid * 8; }
  __device__ double operator()(int comp, long p) const {
    const int nv = A.nvector, n = A.n;
    const long nfull = A.ngrid / nv, full_cells = nfull * nv * 8;
    long batch, r;
    int np;
    if (p < full_cells) { batch = p / (8L * nv); r = p % (8L * nv); np = nv; }
    else { batch = nfull; r = p - full_cells; np = A.ngrid - (int)(nfull * nv); }
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
  }
};
struct MpVecSrc {
  const double *mp;       // (4, ncell)
  const int *igrid;       // the level's list
  int ngrid, nvector;
  long ncell, ncoarse, ngridmax;
  __device__ long count() const { return (long)ngrid * 8; }
  __device__ double operator()(int comp, long p) const {
    const int nv = nvector;
    const long nfull = ngrid / nv, full_cells = nfull * nv * 8;
    long batch, r;
    int np;
    if (p < full_cells) { batch = p / (8L * nv); r = p % (8L * nv); np = nv; }
    else { batch = nfull; r = p - full_cells; np = ngrid - (int)(nfull * nv); }
    const int ind_son = (int)(r / np), j = (int)(r % np);
    const long cell = ncoarse + (long)ind_son * ngridmax + igrid[batch * nv + j] - 1;
    return mp[(long)comp * ncell + cell];
  }
};

size_t multipole_scratch_bytes(long ncells) { return pscan::scratch_bytes(ncells, 4); }

// ===========================================================================================================
// rho_fine's hydro deposit on the levels of an AMR run, on the reference's own cell vectors and tree
// (oracle: ora_rho_fine_amr / ora_rho_deposit_gather, pinned on dumps of the reference).
//   amr_multipole_kernel   multipole_fine(l), pm/rho_fine.f90:666-820: leaf cell -> (m, m x) with m = max(rho,smallr) vol at
//                          the cell centre; split cell -> the sum of its eight children's multipoles, child by child from zero
//   amr_deposit_kernel     cic_from_multipole(l) / cic_cell :825-1142 as a gather: one thread per TARGET cell of the level
//                          visits the 3^3 cells around it (through the FATHER cells: the 3^3 father cells around an oct
//                          exist by the refinement rules, get3cubefather's assumption), recomputes the CIC split of each
//                          source at its centre of mass, tags what it receives with the position of that addition in the
//                          reference's loop nest (batch of nvector octs, ind_son, CIC corner, oct in the batch) and adds in
//                          tag order.  A source whose oct does not exist contributes nothing; a corner whose target oct does
//                          not exist is dropped by the reference (:1128-1139) and is never asked for here.
// ===========================================================================================================
struct AmrRhoArgs {
  const double *dens;     // uold(:,1), cell vector
  double *mp;             // (4, ncell) multipoles (the reference's unew(:,1:4) scratch)
  double *rho;            // (ncell) out
  const double *xg;       // (3, ngridmax)
  const int *son, *nbor, *father;
  const int *igrid;       // the level's octs in list order
  const int *posof;       // oct -> position in that list (-1: not of this level's list)
  int ngrid, nvector;
  long ncell, ncoarse, ngridmax;
  double dx, scale, vol_loc, smallr;
  // several ranks (amr_deposit_hash_kernel): the targets are the first ntarget octs of igrid (the rank's own octs, which
  // deposit, followed by its reception octs); own octs by integer position in an open-addressing table
  int ntarget, level;
  unsigned long long *hkeys;
  int *hvals;
  unsigned hmask;
};

__global__ __launch_bounds__(256) void amr_multipole_kernel(AmrRhoArgs A) {
  const long total = (long)A.ngrid * 8;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    const int ind = (int)(t / A.ngrid), i = (int)(t % A.ngrid);
    const int g = A.igrid[i];
    const long c = A.ncoarse + (long)ind * A.ngridmax + g - 1;
    const int sn = A.son[c];
    double m[4];
    if (sn == 0) {
      const double mm = __builtin_fmax(A.dens[c], A.smallr) * A.vol_loc;
      m[0] = 0.0 + mm;
#pragma unroll
      for (int d = 0; d < 3; d++) {
        const double xc = ((double)((ind >> d) & 1) - 0.5) * A.dx;
        const double xx = (A.xg[(long)d * A.ngridmax + g - 1] + xc - 0.0) * A.scale;
        m[1 + d] = 0.0 + mm * xx;
      }
    } else {
#pragma unroll
      for (int d = 0; d < 4; d++) m[d] = 0.0;
      for (int is = 0; is < 8; is++) {
        const long cs = A.ncoarse + (long)is * A.ngridmax + sn - 1;
#pragma unroll
        for (int d = 0; d < 4; d++) m[d] = m[d] + A.mp[(long)d * A.ncell + cs];
      }
    }
#pragma unroll
    for (int d = 0; d < 4; d++) A.mp[(long)d * A.ncell + c] = m[d];
  }
}

// same-level neighbour of cell c (1-based, level >= 2) in direction dir, 0 if its oct does not exist
__device__ __forceinline__ long amr_rho_nbor_cell(const AmrRhoArgs &A, long c, int dir) {
  const int pos = (int)((c - A.ncoarse - 1) / A.ngridmax);
  const long g = c - A.ncoarse - (long)pos * A.ngridmax;
  const int axis = dir >> 1, up = dir & 1;
  const int bit = (pos >> axis) & 1;
  if (bit != up) return c + (up ? 1 : -1) * ((long)(1 << axis) * A.ngridmax);
  const int nb = A.nbor[(long)dir * A.ngridmax + g - 1];
  const int g2 = A.son[nb - 1];
  if (g2 == 0) return 0;
  return A.ncoarse + (long)(pos ^ (1 << axis)) * A.ngridmax + g2;
}

// what the source cell (oct g_s at position i_s of the list of depositing octs, octant bits sb) gives the cell at offset o
// (target = source + o): the CIC weight at its centre of mass and the tag of that addition in the reference's loop nest
// (batch of nvector octs, ind_son, CIC corner, oct in the batch); false: the target is not one of the source's 8 corners
__device__ __forceinline__ bool amr_cic_from_source(const AmrRhoArgs &A, int g_s, int i_s, const int (&sb)[3], const int (&o)[3], long &key,
                                                    double &val) {
  const double dx = A.dx, scale = A.scale;
  const int ind_son = sb[0] + 2 * sb[1] + 4 * sb[2];
  const long cs = A.ncoarse + (long)ind_son * A.ngridmax + g_s - 1;
  const double m0 = A.mp[cs];
  double w[3];
  int b[3];
#pragma unroll
  for (int d = 0; d < 3; d++) {
    const double xgs = A.xg[(long)d * A.ngridmax + g_s - 1];
    double x = A.mp[(long)(1 + d) * A.ncell + cs] / m0;              // centre of mass
    x = x / scale + 0.0;
    x = x - (xgs - 3.0 * dx);
    x = x / dx;
    double dd = x + 0.5;
    const int id = (int)dd;
    dd = dd - id;
    const double dg = 1.0 - dd;
    const int ig = id - 1;
    const int kt = 2 + sb[d] + o[d];                                   // block coordinate of the target cell
    if (kt == ig) { b[d] = 0; w[d] = dg; }
    else if (kt == id) { b[d] = 1; w[d] = dd; }
    else return false;
  }
  const double vol = w[0] * w[1] * w[2];
  const double vol2 = m0 * vol / A.vol_loc;
  if (vol2 == 0.0) return false;                                       // + 0.0 leaves the (non-negative) sum unchanged
  const int ind = b[0] + 2 * b[1] + 4 * b[2];
  const long batch = i_s / A.nvector, j = i_s % A.nvector;
  key = ((batch * 8 + ind_son) * 8 + ind) * A.nvector + j;
  val = vol2;
  return true;
}

constexpr int ADEP_THREADS = 128;
__global__ __launch_bounds__(ADEP_THREADS) void amr_deposit_kernel(AmrRhoArgs A) {
  __shared__ long skey[27][ADEP_THREADS];
  __shared__ double sval[27][ADEP_THREADS];
  const long total = (long)A.ngrid * 8;
  const long t = (long)blockIdx.x * ADEP_THREADS + threadIdx.x;
  if (t >= total) return;                     // (no barrier below: each thread uses its own LDS column)
  const int ind_t = (int)(t / A.ngrid), it = (int)(t % A.ngrid);
  const int g_t = A.igrid[it];
  const long F_t = A.father[g_t - 1];         // father cell of the target's oct
  int cnt = 0;
  for (int oz = -1; oz <= 1; oz++)
    for (int oy = -1; oy <= 1; oy++)
      for (int ox = -1; ox <= 1; ox++) {
        // source cell = target cell - o: its octant bits, and the step (per direction) to its oct's father cell
        const int o[3] = {ox, oy, oz};
        int sb[3], step[3];
#pragma unroll
        for (int d = 0; d < 3; d++) {
          const int st = ((ind_t >> d) & 1) - o[d];
          step[d] = st < 0 ? -1 : (st > 1 ? 1 : 0);
          sb[d] = st & 1;
        }
        long F_s = F_t;
        if (F_t > A.ncoarse) {
#pragma unroll
          for (int d = 0; d < 3; d++)
            if (step[d] != 0 && F_s > 0) F_s = amr_rho_nbor_cell(A, F_s, 2 * d + (step[d] > 0 ? 1 : 0));
        }
        if (F_s <= 0) continue;
        const int g_s = A.son[F_s - 1];
        if (g_s <= 0) continue;
        const int i_s = A.posof[g_s - 1];
        if (i_s < 0) continue;                 // not a source of this call's list
        long key;
        double val;
        if (!amr_cic_from_source(A, g_s, i_s, sb, o, key, val)) continue;
        skey[cnt][threadIdx.x] = key;
        sval[cnt][threadIdx.x] = val;
        cnt++;
      }
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
  A.rho[A.ncoarse + (long)ind_t * A.ngridmax + g_t - 1] = r;
}

// ---- several ranks ------------------------------------------------------------------------------------------------
// cic_cell loops over the rank's OWN octs only and deposits into own and virtual (reception) cells alike; the reference then
// returns the virtual cells' share to their owners (make_virtual_reverse_dp(rho), added peer by peer) and refreshes the virtual
// copies (make_virtual_fine_dp(rho)), pm/rho_fine.f90:58-60.  The gather below therefore runs over own + reception target cells
// and looks for the source cells among the own octs BY POSITION (integer oct coordinates from xg in a hash table): the tree
// pointers of a virtual oct (nbor) need not be complete (amr/refine_utils.f90:684-700 insists on them for own octs only).
__device__ __forceinline__ unsigned long long amr_rho_oct_key(int x, int y, int z) {
  return (((unsigned long long)z << 42) | ((unsigned long long)y << 21) | (unsigned long long)x) + 1ull;    // never 0 (= empty)
}
__device__ __forceinline__ unsigned amr_rho_hash(unsigned long long k) {
  k ^= k >> 33; k *= 0xff51afd7ed558ccdULL; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ULL; k ^= k >> 33;
  return (unsigned)k;
}
// integer position of oct g on its level (nx = ny = nz = 1: centre = (2 i + 1) dx, an exact binary fraction)
__device__ __forceinline__ void amr_rho_oct_pos(const AmrRhoArgs &A, int g, int (&p)[3]) {
  const double half = 0.5 / A.dx;      // 2^(level-1), exact
#pragma unroll
  for (int d = 0; d < 3; d++) p[d] = (int)(A.xg[(long)d * A.ngridmax + g - 1] * half);
}
__global__ __launch_bounds__(256) void amr_rho_hash_build_kernel(AmrRhoArgs A) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= A.ngrid) return;
  int p[3];
  amr_rho_oct_pos(A, A.igrid[i], p);
  const unsigned long long key = amr_rho_oct_key(p[0], p[1], p[2]);
  unsigned slot = amr_rho_hash(key) & A.hmask;
  for (;;) {
    const unsigned long long prev = atomicCAS(A.hkeys + slot, 0ull, key);
    if (prev == 0ull || prev == key) { A.hvals[slot] = i; return; }
    slot = (slot + 1) & A.hmask;
  }
}
__device__ __forceinline__ int amr_rho_hash_find(const AmrRhoArgs &A, int x, int y, int z) {
  const unsigned long long key = amr_rho_oct_key(x, y, z);
  unsigned slot = amr_rho_hash(key) & A.hmask;
  for (;;) {
    const unsigned long long k = A.hkeys[slot];
    if (k == key) return A.hvals[slot];
    if (k == 0ull) return -1;
    slot = (slot + 1) & A.hmask;
  }
}
__global__ __launch_bounds__(ADEP_THREADS) void amr_deposit_hash_kernel(AmrRhoArgs A) {
  __shared__ long skey[27][ADEP_THREADS];
  __shared__ double sval[27][ADEP_THREADS];
  const long total = (long)A.ntarget * 8;
  const long t = (long)blockIdx.x * ADEP_THREADS + threadIdx.x;
  if (t >= total) return;                     // (no barrier below: each thread uses its own LDS column)
  const int ind_t = (int)(t / A.ntarget), it = (int)(t % A.ntarget);
  const int g_t = A.igrid[it];
  int pt[3];
  amr_rho_oct_pos(A, g_t, pt);
  const int n = 1 << A.level;                 // cells per direction of the (periodic) box on this level
  int cnt = 0;
  for (int oz = -1; oz <= 1; oz++)
    for (int oy = -1; oy <= 1; oy++)
      for (int ox = -1; ox <= 1; ox++) {
        const int o[3] = {ox, oy, oz};
        int sb[3], so[3];
#pragma unroll
        for (int d = 0; d < 3; d++) {
          int sc = 2 * pt[d] + ((ind_t >> d) & 1) - o[d];      // source cell = target cell - o
          sc = sc < 0 ? sc + n : (sc >= n ? sc - n : sc);
          sb[d] = sc & 1;
          so[d] = sc >> 1;
        }
        const int i_s = amr_rho_hash_find(A, so[0], so[1], so[2]);
        if (i_s < 0) continue;                 // no oct there, or not one of this rank's own
        const int g_s = A.igrid[i_s];
        long key;
        double val;
        if (!amr_cic_from_source(A, g_s, i_s, sb, o, key, val)) continue;
        skey[cnt][threadIdx.x] = key;
        sval[cnt][threadIdx.x] = val;
        cnt++;
      }
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
  A.rho[A.ncoarse + (long)ind_t * A.ngridmax + g_t - 1] = r;
}

static AmrRhoArgs amr_rho_args(const double *dens, double *mp, double *rho, const double *xg, const int *son, const int *igrid, int ngrid,
                               int nvector, long ncoarse, long ngridmax, int ilevel, double boxlen_over_nx, double smallr) {
  AmrRhoArgs A;
  A.dens = dens; A.mp = mp; A.rho = rho; A.xg = xg; A.son = son; A.nbor = nullptr; A.father = nullptr;
  A.igrid = igrid; A.posof = nullptr; A.ngrid = ngrid; A.nvector = nvector;
  A.ncoarse = ncoarse; A.ngridmax = ngridmax; A.ncell = ncoarse + 8 * ngridmax;
  double dx = 1.0;
  for (int l = 0; l < ilevel; l++) dx *= 0.5;
  A.dx = dx; A.scale = boxlen_over_nx;
  const double dx_loc = dx * A.scale;
  A.vol_loc = dx_loc * dx_loc * dx_loc;
  A.smallr = smallr;
  A.ntarget = ngrid; A.level = ilevel; A.hkeys = nullptr; A.hvals = nullptr; A.hmask = 0;
  return A;
}
// several ranks, step 1 of a level: multipole_fine(l) on the rank's own octs (the children's multipoles of level l+1, own or
// received, are in mp); the caller then exchanges mp(:,1:4) of the level (make_virtual_fine_dp(unew(1,idim),l), :814-817)
hipError_t launch_amr_multipole_level(const double *dens, double *mp, const double *xg, const int *son, const int *igrid, int n_own,
                                      long ncoarse, long ngridmax, int ilevel, double boxlen_over_nx, double smallr, hipStream_t s) {
  if (n_own <= 0) return hipSuccess;
  AmrRhoArgs A = amr_rho_args(dens, mp, nullptr, xg, son, igrid, n_own, 1, ncoarse, ngridmax, ilevel, boxlen_over_nx, smallr);
  long gm = ((long)n_own * 8 + 255) / 256;
  if (gm > 16384) gm = 16384;
  hipLaunchKernelGGL(amr_multipole_kernel, dim3((unsigned)gm), dim3(256), 0, s, A);
  return hipGetLastError();
}
// step 2: cic_from_multipole(l) -- rho of the own AND reception cells (igrid: n_own own octs followed by the reception octs,
// n_all in all) from the own octs' multipoles; hkeys / hvals: hcap slots (a power of two >= 2 n_own), hkeys zeroed here
hipError_t launch_amr_deposit_level(double *mp, double *rho, const double *xg, const int *igrid, int n_own, int n_all, int nvector,
                                    long ncoarse, long ngridmax, int ilevel, double boxlen_over_nx, unsigned long long *hkeys, int *hvals,
                                    unsigned hcap, hipStream_t s) {
  if (n_all <= 0) return hipSuccess;
  AmrRhoArgs A = amr_rho_args(nullptr, mp, rho, xg, nullptr, igrid, n_own, nvector, ncoarse, ngridmax, ilevel, boxlen_over_nx, 0.0);
  A.ntarget = n_all; A.hkeys = hkeys; A.hvals = hvals; A.hmask = hcap - 1;
  hipError_t e = hipMemsetAsync(hkeys, 0, sizeof(unsigned long long) * (size_t)hcap, s);
  if (e != hipSuccess) return e;
  if (n_own > 0) hipLaunchKernelGGL(amr_rho_hash_build_kernel, dim3((n_own + 255) / 256), dim3(256), 0, s, A);
  const long total = (long)n_all * 8;
  hipLaunchKernelGGL(amr_deposit_hash_kernel, dim3((unsigned)((total + ADEP_THREADS - 1) / ADEP_THREADS)), dim3(ADEP_THREADS), 0, s, A);
  return hipGetLastError();
}

__global__ void amr_rho_posof_kernel(const int *igrid, int ngrid, int *posof, int set) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < ngrid) posof[igrid[i] - 1] = set ? i : -1;
}

// one level of the call: multipoles, then the deposit; posof (ngridmax ints, all -1 on entry and again on return)
hipError_t launch_amr_rho_level(const double *dens, double *mp, double *rho, const double *xg, const int *son, const int *nbor,
                                const int *father, const int *igrid, int *posof, int ngrid, int nvector, long ncoarse, long ngridmax,
                                int ilevel, double boxlen_over_nx, double smallr, hipStream_t s) {
  if (ngrid <= 0) return hipSuccess;
  AmrRhoArgs A;
  A.dens = dens; A.mp = mp; A.rho = rho; A.xg = xg; A.son = son; A.nbor = nbor; A.father = father;
  A.igrid = igrid; A.posof = posof; A.ngrid = ngrid; A.nvector = nvector;
  A.ncoarse = ncoarse; A.ngridmax = ngridmax; A.ncell = ncoarse + 8 * ngridmax;
  double dx = 1.0;
  for (int l = 0; l < ilevel; l++) dx *= 0.5;
  A.dx = dx; A.scale = boxlen_over_nx;
  const double dx_loc = dx * A.scale;
  A.vol_loc = dx_loc * dx_loc * dx_loc;
  A.smallr = smallr;
  const long total = (long)ngrid * 8;
  long gm = (total + 255) / 256;
  if (gm > 16384) gm = 16384;
  hipLaunchKernelGGL(amr_multipole_kernel, dim3((unsigned)gm), dim3(256), 0, s, A);
  hipLaunchKernelGGL(amr_rho_posof_kernel, dim3((ngrid + 255) / 256), dim3(256), 0, s, igrid, ngrid, posof, 1);
  hipLaunchKernelGGL(amr_deposit_kernel, dim3((unsigned)((total + ADEP_THREADS - 1) / ADEP_THREADS)), dim3(ADEP_THREADS), 0, s, A);
  hipLaunchKernelGGL(amr_rho_posof_kernel, dim3((ngrid + 255) / 256), dim3(256), 0, s, igrid, ngrid, posof, 0);
  return hipGetLastError();
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
hipError_t launch_multipole(const RhoArgs &A, double *out4, void *scratch, hipStream_t s) {
  MpBrickSrc S;
  S.A = A;
  return pscan::launch<MpBrickSrc, 4>(S, (long)A.ngrid * 8, out4, scratch, s);
}
hipError_t launch_multipole_vec(const double *mp, const int *igrid, int ngrid, int nvector, long ncell, long ncoarse, long ngridmax,
                                double *out4, void *scratch, hipStream_t s) {
  MpVecSrc S;
  S.mp = mp; S.igrid = igrid; S.ngrid = ngrid; S.nvector = nvector; S.ncell = ncell; S.ncoarse = ncoarse; S.ngridmax = ngridmax;
  return pscan::launch<MpVecSrc, 4>(S, (long)ngrid * 8, out4, scratch, s);
}

}  // namespace ramses_amd

#include "warm.hpp"
RAMSES_AMD_TU_WARM(rho_fine)
