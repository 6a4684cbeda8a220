// amr_layout.hpp -- the device's own numbering of the octs of a resident AMR run (csrc/capi_amr.hip).
//
// The reference addresses the cells of an oct as icell = ncoarse + (ind-1)*ngridmax + igrid
// (hydro/godunov_fine.f90:600-601) and numbers the octs in the order refine_fine / load_balance created them, so that the
// cells a sweep touches together lie ngridmax doubles apart and next to cells from the other end of the box.  The device keeps
// that FORMULA -- every kernel of the path walks the tree through it -- but not the numbers: the resident copy of the cell
// vectors is private to the device, so the octs get device indices of their own, chosen level by level such that
//
//   * a level stored in TILES (levels of 64^3 cells up to 4096^3): the periodic box of the level is cut into tiles of
//     32 x 4 x 4 octs (64 x 8 x 8 cells); every tile that holds an oct of the level, or borders one (room for the ghost octs a
//     sweep interpolates), owns 512 consecutive device indices, oct (lx, ly, lz) of the tile at lx + 32 (ly + 4 lz).  For one
//     octant position the cells of a tile are then a dense 32 x 4 x 4 brick of doubles: a wavefront of the dense sweep
//     (csrc/hydro_sweep.hip) reads a row of 64 cells as two 256-byte runs.  A fully refined level is simply a level all of
//     whose tiles exist -- "level-contiguous SoA blocks", and nothing but the tile directory between a cell and its address;
//   * the other levels (coarser than 64^3, finer than 4096^3, or when the tiles do not fit into the device's index space) are
//     numbered along the Z-order curve of their octs, siblings adjacent (what the tree-walking sweep reads best).
//
// Host indices exist only at the C ABI: oct lists are translated as they arrive (perm), cell indices that go back are
// translated on the way out (iperm), level data moves through the lists anyway.  The tree arrays son / nbor / father are
// rewritten into device numbers after every regrid.  A level whose set of octs did not change keeps its device indices and
// therefore its data: only the levels refine_fine rebuilt (which the host re-sends anyway, a suffix of the levels) move.
// RAMSES_AMD_DEVICE_ORDER=0 (or a coarse grid of more than one cell: physical boundaries) keeps the host's numbering.
#pragma once
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <cstdint>
#include <cstdlib>
#include <vector>

#include "sweep_args.hpp"

namespace ramses_amd {
namespace amrlayout {

struct Buf {
  void *p = nullptr;
  size_t cap = 0;
  hipError_t ensure(size_t bytes) {
    if (bytes <= cap && p) return hipSuccess;
    if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
    if (bytes == 0) bytes = 8;
    hipError_t e = hipMalloc(&p, bytes);
    if (e == hipSuccess) cap = bytes;
    return e;
  }
  template <class T> T *as() const { return reinterpret_cast<T *>(p); }
};

typedef unsigned long long u64;
constexpr int KEY_BITS = 19;                  // oct coordinates of levels <= 20
constexpr u64 KEY_MASK = (1ull << KEY_BITS) - 1;
constexpr int MAX_LEVEL = 20, TILE_MIN_LEVEL = 6, TILE_MAX_LEVEL = 12;   // tiles: 2^(L-1) octs per side in [32, 2048] (64^3 .. 4096^3 cells)
__host__ __device__ inline u64 key_pack(unsigned x, unsigned y, unsigned z) { return (u64)x | ((u64)y << KEY_BITS) | ((u64)z << (2 * KEY_BITS)); }
__host__ __device__ inline unsigned key_x(u64 k) { return (unsigned)(k & KEY_MASK); }
__host__ __device__ inline unsigned key_y(u64 k) { return (unsigned)((k >> KEY_BITS) & KEY_MASK); }
__host__ __device__ inline unsigned key_z(u64 k) { return (unsigned)((k >> (2 * KEY_BITS)) & KEY_MASK); }

// ---- kernels ------------------------------------------------------------------------------------------------------------
// the sons of the octs of one level: (host oct, coordinates) pairs of the next level, in whatever order the workgroups arrive
__global__ __launch_bounds__(1024) void bfs_kernel(const int *__restrict__ son_h, long ncoarse, long ngh, const int *__restrict__ phoct,
                                                   const u64 *__restrict__ pkey, int np, int *__restrict__ choct, u64 *__restrict__ ckey,
                                                   int *__restrict__ count, int cap) {
  __shared__ int wcount[16], wbase[16];
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  int s = 0;
  u64 k = 0;
  if (t < (long)np * 8) {
    const int ind = (int)(t / np), i = (int)(t % np);
    const int g = phoct[i];
    s = son_h[ncoarse + (long)ind * ngh + g - 1];
    if (s > 0) {
      const u64 pk = pkey[i];
      k = key_pack(2 * key_x(pk) + (ind & 1), 2 * key_y(pk) + ((ind >> 1) & 1), 2 * key_z(pk) + (ind >> 2));
    }
  }
  const bool have = s > 0;
  const unsigned long long m = __ballot(have);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) wcount[wave] = __popcll(m);
  __syncthreads();
  if (threadIdx.x == 0) {
    int tot = 0;
    for (int w = 0; w < (int)(blockDim.x >> 6); w++) { wbase[w] = tot; tot += wcount[w]; }
    const int b = tot > 0 ? atomicAdd(count, tot) : 0;
    for (int w = 0; w < (int)(blockDim.x >> 6); w++) wbase[w] += b;
  }
  __syncthreads();
  if (have) {
    const int pos = wbase[wave] + __popcll(m & ((1ull << lane) - 1ull));
    if (pos < cap) { choct[pos] = s; ckey[pos] = k; }
  }
}

// does the level hold exactly the octs (host index, position) it held before?
__global__ void same_level_kernel(const int *__restrict__ hoct, const u64 *__restrict__ key, int n, int level, const int *__restrict__ perm,
                                  const u64 *__restrict__ okey, int *__restrict__ mismatch) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int g = hoct[i];
  if (perm[g - 1] == 0 || okey[g - 1] != (((u64)level << (3 * KEY_BITS)) | key[i])) atomicAdd(mismatch, 1);
}

__device__ __forceinline__ int tile_of(unsigned x, unsigned y, unsigned z, int ntx, int nty) {
  return (int)(x / TILE_OX) + ntx * ((int)(y / TILE_OY) + nty * (int)(z / TILE_OZ));
}
// every tile that holds an oct, or one of an oct's 26 neighbours (periodic box of `no` octs per side)
__global__ void tile_mark_kernel(const u64 *__restrict__ key, int n, int no, int ntx, int nty, int *__restrict__ need) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long)n * 27) return;
  const int i = (int)(t / 27), o = (int)(t % 27);
  const u64 k = key[i];
  const unsigned x = (key_x(k) + no + (o % 3) - 1) & (no - 1), y = (key_y(k) + no + ((o / 3) % 3) - 1) & (no - 1),
                 z = (key_z(k) + no + (o / 9) - 1) & (no - 1);
  need[tile_of(x, y, z, ntx, nty)] = 1;
}
// dir[t] = 0-based cell index of the tile's first cell (octant position 1), -1: no tile; tileid[rank] = t
__global__ void tile_dir_kernel(const int *__restrict__ need, const int *__restrict__ rank, int nt, long base, long ncoarse, int *__restrict__ dir,
                                int *__restrict__ tileid) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nt) return;
  if (need[t]) { dir[t] = (int)(ncoarse + base - 1 + (long)rank[t] * TILE_OCTS); tileid[rank[t]] = t; }
  else dir[t] = -1;
}
__global__ void assign_tiles_kernel(const u64 *__restrict__ key, int n, int ntx, int nty, const int *__restrict__ rank, long base, int *__restrict__ doct) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const u64 k = key[i];
  const unsigned x = key_x(k), y = key_y(k), z = key_z(k);
  doct[i] = (int)(base + (long)rank[tile_of(x, y, z, ntx, nty)] * TILE_OCTS + (x % TILE_OX) + TILE_OX * ((y % TILE_OY) + TILE_OY * (z % TILE_OZ)));
}
// Z-order key of an oct position (siblings adjacent) and the identity permutation, for the sort of a level kept compact
__global__ void morton_kernel(const u64 *__restrict__ key, int n, u64 *__restrict__ mkey, int *__restrict__ idx) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const u64 k = key[i];
  const unsigned x = key_x(k), y = key_y(k), z = key_z(k);
  u64 m = 0;
  for (int b = 0; b < KEY_BITS; b++) m |= ((u64)((x >> b) & 1) << (3 * b)) | ((u64)((y >> b) & 1) << (3 * b + 1)) | ((u64)((z >> b) & 1) << (3 * b + 2));
  mkey[i] = m; idx[i] = i;
}
__global__ void assign_compact_kernel(const int *__restrict__ sorted_idx, int n, long base, int *__restrict__ doct) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r < n) doct[sorted_idx[r]] = (int)(base + r);
}
__global__ void perm_kernel(const int *__restrict__ hoct, const int *__restrict__ doct, const u64 *__restrict__ key, int n, int level, int *__restrict__ perm,
                            int *__restrict__ iperm, u64 *__restrict__ okey) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int g = hoct[i], d = doct[i];
  perm[g - 1] = d; iperm[d - 1] = g;
  okey[g - 1] = ((u64)level << (3 * KEY_BITS)) | key[i];
}
// host cell index (1-based) -> device cell index
__device__ __forceinline__ int cell_h2d(int c, const int *__restrict__ perm, long ncoarse, long ngh, long ngd) {
  if (c <= ncoarse) return c;
  const long r = (long)c - ncoarse - 1;
  const int ind = (int)(r / ngh), g = (int)(r % ngh) + 1;
  const int d = perm[g - 1];
  return d > 0 ? (int)(ncoarse + (long)ind * ngd + d) : 0;
}
// son / nbor / father of the octs of one level into device numbers; the status byte of their cells (refined or not)
__global__ void xlate_tree_kernel(const int *__restrict__ hoct, const int *__restrict__ doct, int n, const int *__restrict__ son_h,
                                  const int *__restrict__ nbor_h, const int *__restrict__ father_h, const int *__restrict__ perm, long ncoarse, long ngh,
                                  long ngd, int *__restrict__ son_d, int *__restrict__ nbor_d, int *__restrict__ father_d, unsigned char *__restrict__ stat) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long)n * 15) return;
  const int k = (int)(t / n), i = (int)(t % n);
  const int g = hoct[i], d = doct[i];
  if (k < 8) {
    const int s = son_h[ncoarse + (long)k * ngh + g - 1];
    const int sd = s > 0 ? perm[s - 1] : s;
    const long c = ncoarse + (long)k * ngd + d - 1;
    son_d[c] = sd;
    stat[c] = (unsigned char)((stat[c] & ~CELL_REFINED) | (sd > 0 ? CELL_REFINED : 0));     // (the other bits belong to the level's sweep plan)
  } else if (k < 14) {
    nbor_d[(long)(k - 8) * ngd + d - 1] = cell_h2d(nbor_h[(long)(k - 8) * ngh + g - 1], perm, ncoarse, ngh, ngd);
  } else {
    father_d[d - 1] = cell_h2d(father_h[g - 1], perm, ncoarse, ngh, ngd);
  }
}
__global__ void xlate_coarse_kernel(const int *__restrict__ son_h, const int *__restrict__ perm, long ncoarse, int *__restrict__ son_d) {
  const long c = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= ncoarse) return;
  const int s = son_h[c];
  son_d[c] = s > 0 ? perm[s - 1] : s;
}
// an oct list from the host, in place: host indices -> device indices (an index that is not in the tree counts as bad)
__global__ void xlate_list_kernel(int *__restrict__ list, int n, const int *__restrict__ perm, long ngh, int *__restrict__ bad) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int g = list[i];
  int d = (g >= 1 && g <= ngh) ? perm[g - 1] : 0;
  if (d == 0) { atomicAdd(bad, 1); d = 1; }
  list[i] = d;
}
__global__ void xlate_list_copy_kernel(const int *__restrict__ src, int *__restrict__ dst, int n, const int *__restrict__ perm, long ngh, int *__restrict__ bad) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int g = src[i];
  int d = (g >= 1 && g <= ngh) ? perm[g - 1] : 0;
  if (d == 0) { atomicAdd(bad, 1); d = 1; }
  dst[i] = d;
}
// device cell indices (1-based) -> host cell indices, in place
__global__ void cells_d2h_kernel(int *__restrict__ cells, int n, const int *__restrict__ iperm, long ncoarse, long ngh, long ngd) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int c = cells[i];
  if (c <= ncoarse) return;
  const long r = (long)c - ncoarse - 1;
  const int ind = (int)(r / ngd), d = (int)(r % ngd) + 1;
  cells[i] = (int)(ncoarse + (long)ind * ngh + iperm[d - 1]);
}
// one variable of the octs of one level between a vector in the host's numbering and one in the device's
template <bool TO_DEVICE>
__global__ void move_var_kernel(const int *__restrict__ hoct, const int *__restrict__ doct, int n, long ncoarse, long ngh, long ngd, double *__restrict__ vh,
                                double *__restrict__ vd) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long)n * 8) return;
  const int ind = (int)(t / n), i = (int)(t % n);
  const long ch = ncoarse + (long)ind * ngh + hoct[i] - 1, cd = ncoarse + (long)ind * ngd + doct[i] - 1;
  if (TO_DEVICE) vd[cd] = vh[ch];
  else vh[ch] = vd[cd];
}
// an oct-indexed array of ncomp components (xg): host numbering -> device numbering
__global__ void move_oct_kernel(const int *__restrict__ hoct, const int *__restrict__ doct, int n, int ncomp, long ngh, long ngd, const double *__restrict__ vh,
                                double *__restrict__ vd) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long)n * ncomp) return;
  const int k = (int)(t / n), i = (int)(t % n);
  vd[(long)k * ngd + doct[i] - 1] = vh[(long)k * ngh + hoct[i] - 1];
}

// ---- host side ----------------------------------------------------------------------------------------------------------
struct LevelMap {
  int level = 0, n = 0;
  int version = 0;             // changes when the level is laid out again (what was derived from its device indices is stale)
  int layout = 0;              // 0: compact (Z-order), 1: tiles
  long base = 0, cap = 0;      // device indices [base, base + cap), 1-based
  int no = 0, ntx = 0, nty = 0, ntz = 0, ntiles = 0;
  Buf hoct, key, doct;         // the level's octs: host index, position, device index (same order)
  Buf dir, tileid;             // tiles: directory [ntz][nty][ntx], rank -> tile
};

inline int grid1(long n, int b = 256) { long g = (n + b - 1) / b; return (int)(g < 1 ? 1 : g); }

struct DevMap {
  bool on = false;             // false: the host's numbering on the device too (perm = identity, nothing is translated)
  long ngh = 0, ngd = 0, ncoarse = 0;
  int serial = 0;              // bumped by every build(): whatever was derived from device indices is stale
  int nlev = 0;
  std::vector<LevelMap> lev;   // [0] unused, [l] level l
  Buf perm, iperm, okey, son_h, nbor_h, father_h;
  Buf need, rank, cubtmp, cnt, mkey, mkey2, idx, idx2;
  Buf c_hoct[2], c_key[2];     // BFS double buffer
  int first_changed = 0;       // of the last build: the coarsest level that was laid out again (nlev + 1: none)
  long kept_end = 1;           // of the last build: first device index after the levels that kept their layout
  int version_counter = 0;
  long tiles_levels = 0;       // how many levels are stored in tiles (diagnostics)
  const char *why_not = "";
  bool overflow = false;       // the last build failed because the levels that kept their tiles left too little room for the finer ones

#define LCHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return e_; } while (0)

  static void drop(LevelMap &L) {
    for (Buf *b : {&L.hoct, &L.key, &L.doct, &L.dir, &L.tileid}) { if (b->p) (void)hipFree(b->p); b->p = nullptr; b->cap = 0; }
  }
  void reset(long ngh_, long ngd_, long ncoarse_, bool want) {
    for (size_t l = 1; l < lev.size(); l++) drop(lev[l]);
    ngh = ngh_; ngd = ngd_; ncoarse = ncoarse_; nlev = 0; lev.clear(); lev.resize(1);
    on = want && ncoarse == 1;
    const char *e = getenv("RAMSES_AMD_DEVICE_ORDER");
    if (e && e[0] == '0') on = false;
    if (!on) ngd = ngh;
    serial++;
    first_changed = 1;
  }

  // the next build lays every level out again (their device data is the caller's to move)
  void forget() { nlev = 0; }

  hipError_t scan_need(int nt, int &ntiles, hipStream_t s) {
    size_t bytes = 0;
    LCHK(hipcub::DeviceScan::ExclusiveSum(nullptr, bytes, need.as<int>(), rank.as<int>(), nt, s));
    LCHK(cubtmp.ensure(bytes));
    LCHK(hipcub::DeviceScan::ExclusiveSum(cubtmp.p, bytes, need.as<int>(), rank.as<int>(), nt, s));
    int last[2];
    LCHK(hipMemcpyAsync(&last[0], need.as<int>() + nt - 1, sizeof(int), hipMemcpyDeviceToHost, s));
    LCHK(hipMemcpyAsync(&last[1], rank.as<int>() + nt - 1, sizeof(int), hipMemcpyDeviceToHost, s));
    LCHK(hipStreamSynchronize(s));
    ntiles = last[0] + last[1];
    return hipSuccess;
  }

  // The tree after load / refine_fine (host arrays son(1:ncell), nbor(1:ngridmax,1:6), father(1:ngridmax)): lay the levels
  // out, rewrite the tree in device numbers into son_d / nbor_d / father_d (device arrays of the device's ncell / ngd) and the
  // status bytes.  With `on` false the arrays are copied as they are.
  hipError_t build(const int *son, const int *nbor, const int *father, int *son_d, int *nbor_d, int *father_d, unsigned char *stat, hipStream_t s) {
    serial++;
    overflow = false;
    const size_t ncell_h = (size_t)(ncoarse + 8 * ngh), ncell_d = (size_t)(ncoarse + 8 * ngd);
    if (!on) {
      LCHK(hipMemcpyAsync(son_d, son, sizeof(int) * ncell_h, hipMemcpyHostToDevice, s));
      LCHK(hipMemcpyAsync(nbor_d, nbor, sizeof(int) * 6 * (size_t)ngh, hipMemcpyHostToDevice, s));
      LCHK(hipMemcpyAsync(father_d, father, sizeof(int) * (size_t)ngh, hipMemcpyHostToDevice, s));
      LCHK(hipStreamSynchronize(s));
      first_changed = 1;
      return hipSuccess;
    }
    LCHK(son_h.ensure(sizeof(int) * ncell_h)); LCHK(nbor_h.ensure(sizeof(int) * 6 * (size_t)ngh)); LCHK(father_h.ensure(sizeof(int) * (size_t)ngh));
    LCHK(hipMemcpyAsync(son_h.p, son, sizeof(int) * ncell_h, hipMemcpyHostToDevice, s));
    LCHK(hipMemcpyAsync(nbor_h.p, nbor, sizeof(int) * 6 * (size_t)ngh, hipMemcpyHostToDevice, s));
    LCHK(hipMemcpyAsync(father_h.p, father, sizeof(int) * (size_t)ngh, hipMemcpyHostToDevice, s));
    const bool fresh = perm.p == nullptr || perm.cap < sizeof(int) * (size_t)ngh || nlev == 0;
    LCHK(perm.ensure(sizeof(int) * (size_t)ngh)); LCHK(iperm.ensure(sizeof(int) * (size_t)ngd)); LCHK(okey.ensure(sizeof(u64) * (size_t)ngh));
    LCHK(cnt.ensure(sizeof(int) * 4));
    if (fresh) {
      LCHK(hipMemsetAsync(perm.p, 0, sizeof(int) * (size_t)ngh, s));
      LCHK(hipMemsetAsync(okey.p, 0, sizeof(u64) * (size_t)ngh, s));
      for (size_t l = 1; l < lev.size(); l++) lev[l].n = 0;
      nlev = 0;
    }
    // ---- 1. the octs of every level, top down, into scratch lists; a level whose set of (host oct, position) pairs is the one
    //         it held before keeps its record (list order, device indices, directory); from the first level that differs on,
    //         every level is taken over from the scratch lists and laid out again (the host re-sends those levels: a level can
    //         only change when refine_fine rebuilt it, and it rebuilds a suffix of the levels).  No allocation unless a list grows.
    if (lev.size() < (size_t)MAX_LEVEL + 2) lev.resize((size_t)MAX_LEVEL + 2);
    const int g1 = son[0];
    if (g1 <= 0) { why_not = "the coarse cell has no oct"; return hipErrorInvalidValue; }
    bool keeping = !fresh;
    int k0 = 0, nnew = 0;
    for (int l = 1; l <= MAX_LEVEL + 1; l++) {
      Buf &bo = c_hoct[l & 1], &bk = c_key[l & 1];
      int nc = 0;
      if (l == 1) {
        const u64 kz = 0;
        LCHK(bo.ensure(sizeof(int))); LCHK(bk.ensure(sizeof(u64)));
        LCHK(hipMemcpyAsync(bo.p, &g1, sizeof(int), hipMemcpyHostToDevice, s));
        LCHK(hipMemcpyAsync(bk.p, &kz, sizeof(u64), hipMemcpyHostToDevice, s));
        LCHK(hipStreamSynchronize(s));
        nc = 1;
      } else {
        LevelMap &P = lev[l - 1];
        const long capl = std::min<long>((long)P.n * 8, ngh);
        LCHK(bo.ensure(sizeof(int) * (size_t)capl)); LCHK(bk.ensure(sizeof(u64) * (size_t)capl));
        LCHK(hipMemsetAsync(cnt.p, 0, sizeof(int), s));
        hipLaunchKernelGGL(bfs_kernel, dim3(grid1((long)P.n * 8, 1024)), dim3(1024), 0, s, son_h.as<int>(), ncoarse, ngh, P.hoct.as<int>(), P.key.as<u64>(), P.n,
                           bo.as<int>(), bk.as<u64>(), cnt.as<int>(), (int)capl);
        LCHK(hipMemcpyAsync(&nc, cnt.p, sizeof(int), hipMemcpyDeviceToHost, s));
        LCHK(hipStreamSynchronize(s));
        if (nc > capl) { why_not = "more octs in the tree than ngridmax"; return hipErrorInvalidValue; }
      }
      if (nc == 0) break;
      if (l > MAX_LEVEL) { why_not = "more than 20 levels"; return hipErrorInvalidValue; }
      nnew = l;
      LevelMap &L = lev[l];
      if (keeping) {
        bool same = l <= nlev && nc == L.n;
        if (same) {
          LCHK(hipMemsetAsync(cnt.p, 0, sizeof(int), s));
          hipLaunchKernelGGL(same_level_kernel, dim3(grid1(nc)), dim3(256), 0, s, bo.as<int>(), bk.as<u64>(), nc, l, perm.as<int>(), okey.as<u64>(), cnt.as<int>());
          int mis = 0;
          LCHK(hipMemcpyAsync(&mis, cnt.p, sizeof(int), hipMemcpyDeviceToHost, s));
          LCHK(hipStreamSynchronize(s));
          same = mis == 0;
        }
        if (same) continue;            // the level keeps its record; its old list (the same set) feeds the next level
        keeping = false;
      }
      if (k0 == 0) k0 = l;
      L.level = l; L.n = nc; L.no = 1 << (l - 1);
      LCHK(L.hoct.ensure(sizeof(int) * (size_t)nc)); LCHK(L.key.ensure(sizeof(u64) * (size_t)nc)); LCHK(L.doct.ensure(sizeof(int) * (size_t)nc));
      LCHK(hipMemcpyAsync(L.hoct.p, bo.p, sizeof(int) * (size_t)nc, hipMemcpyDeviceToDevice, s));
      LCHK(hipMemcpyAsync(L.key.p, bk.p, sizeof(u64) * (size_t)nc, hipMemcpyDeviceToDevice, s));
    }
    if (k0 == 0) k0 = nnew + 1;        // nothing changed (or only levels at the fine end disappeared)
    for (int l = nnew + 1; l <= nlev; l++) lev[l].n = 0;
    first_changed = k0;
    nlev = nnew;
    // ---- 3. lay out the levels from k0 on: tiles where they fit, Z-order otherwise -----------------------------------------
    long next = 1;
    for (int l = 1; l < k0; l++) next = std::max(next, lev[l].base + lev[l].cap);
    kept_end = next;
    std::vector<long> after((size_t)nlev + 2, 0);            // octs of the finer levels (they need at least that much)
    for (int l = nlev; l >= 1; l--) after[l] = after[l + 1] + lev[l].n;
    const char *et = getenv("RAMSES_AMD_TILES");
    const bool tiles_on = !(et && et[0] == '0');
    for (int l = k0; l <= nlev; l++) {
      LevelMap &L = lev[l];
      L.layout = 0; L.base = next; L.cap = L.n; L.ntiles = 0;
      L.version = ++version_counter;
      if (tiles_on && l >= TILE_MIN_LEVEL && l <= TILE_MAX_LEVEL) {
        L.ntx = L.no / TILE_OX; L.nty = L.no / TILE_OY; L.ntz = L.no / TILE_OZ;
        const int nt = L.ntx * L.nty * L.ntz;
        LCHK(need.ensure(sizeof(int) * (size_t)nt)); LCHK(rank.ensure(sizeof(int) * (size_t)nt));
        LCHK(hipMemsetAsync(need.p, 0, sizeof(int) * (size_t)nt, s));
        hipLaunchKernelGGL(tile_mark_kernel, dim3(grid1((long)L.n * 27)), dim3(256), 0, s, L.key.as<u64>(), L.n, L.no, L.ntx, L.nty, need.as<int>());
        int ntiles = 0;
        LCHK(scan_need(nt, ntiles, s));
        const long capt = (long)ntiles * TILE_OCTS;
        if (next + capt + after[l + 1] <= ngd + 1) {
          L.layout = 1; L.cap = capt; L.ntiles = ntiles;
          LCHK(L.dir.ensure(sizeof(int) * (size_t)nt)); LCHK(L.tileid.ensure(sizeof(int) * (size_t)(ntiles > 0 ? ntiles : 1)));
          hipLaunchKernelGGL(tile_dir_kernel, dim3(grid1(nt)), dim3(256), 0, s, need.as<int>(), rank.as<int>(), nt, L.base, ncoarse, L.dir.as<int>(), L.tileid.as<int>());
          hipLaunchKernelGGL(assign_tiles_kernel, dim3(grid1(L.n)), dim3(256), 0, s, L.key.as<u64>(), L.n, L.ntx, L.nty, rank.as<int>(), L.base, L.doct.as<int>());
          LCHK(hipStreamSynchronize(s));           // (need / rank are reused by the next level)
        }
      }
      if (L.layout == 0) {
        LCHK(mkey.ensure(sizeof(u64) * (size_t)L.n)); LCHK(mkey2.ensure(sizeof(u64) * (size_t)L.n));
        LCHK(idx.ensure(sizeof(int) * (size_t)L.n)); LCHK(idx2.ensure(sizeof(int) * (size_t)L.n));
        hipLaunchKernelGGL(morton_kernel, dim3(grid1(L.n)), dim3(256), 0, s, L.key.as<u64>(), L.n, mkey.as<u64>(), idx.as<int>());
        size_t bytes = 0;
        LCHK(hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, mkey.as<u64>(), mkey2.as<u64>(), idx.as<int>(), idx2.as<int>(), L.n, 0, 3 * KEY_BITS, s));
        LCHK(cubtmp.ensure(bytes));
        LCHK(hipcub::DeviceRadixSort::SortPairs(cubtmp.p, bytes, mkey.as<u64>(), mkey2.as<u64>(), idx.as<int>(), idx2.as<int>(), L.n, 0, 3 * KEY_BITS, s));
        hipLaunchKernelGGL(assign_compact_kernel, dim3(grid1(L.n)), dim3(256), 0, s, idx2.as<int>(), L.n, L.base, L.doct.as<int>());
        LCHK(hipStreamSynchronize(s));
      }
      next = L.base + L.cap;
      if (next > ngd + 1) {
        // The tree itself fits (the BFS counted <= ngridmax octs): what is in the way are the free slots of the tiles of levels
        // that kept their layout -- their fit was checked against the finer levels of THAT moment.  The caller parks the kept
        // levels' data, forgets the layout (forget()) and builds again from level 1 (csrc/capi_amr.hip ramses_amd_amrres_tree).
        why_not = "the levels do not fit into the device's index space";
        overflow = k0 > 1;
        return hipErrorInvalidValue;
      }
    }
    // ---- 4. perm / iperm / okey, the tree in device numbers, the status bytes -----------------------------------------------
    LCHK(hipMemsetAsync(perm.p, 0, sizeof(int) * (size_t)ngh, s));
    LCHK(hipMemsetAsync(iperm.p, 0, sizeof(int) * (size_t)ngd, s));
    for (int l = 1; l <= nlev; l++)
      hipLaunchKernelGGL(perm_kernel, dim3(grid1(lev[l].n)), dim3(256), 0, s, lev[l].hoct.as<int>(), lev[l].doct.as<int>(), lev[l].key.as<u64>(), lev[l].n, l,
                         perm.as<int>(), iperm.as<int>(), okey.as<u64>());
    LCHK(hipMemsetAsync(son_d, 0, sizeof(int) * ncell_d, s));
    LCHK(hipMemsetAsync(nbor_d, 0, sizeof(int) * 6 * (size_t)ngd, s));
    LCHK(hipMemsetAsync(father_d, 0, sizeof(int) * (size_t)ngd, s));
    // the status bytes of the index range that was laid out again start from zero; the kept levels keep what their sweep plans
    // wrote (only the "refined" bit follows the tree, below)
    if (fresh) LCHK(hipMemsetAsync(stat, 0, ncell_d, s));
    else
      for (int ind = 0; ind < 8; ind++)
        LCHK(hipMemsetAsync(stat + ncoarse + (size_t)ind * ngd + (kept_end - 1), 0, (size_t)(ngd - kept_end + 1), s));
    hipLaunchKernelGGL(xlate_coarse_kernel, dim3(grid1(ncoarse)), dim3(256), 0, s, son_h.as<int>(), perm.as<int>(), ncoarse, son_d);
    tiles_levels = 0;
    for (int l = 1; l <= nlev; l++) {
      LevelMap &L = lev[l];
      if (L.layout == 1) tiles_levels++;
      hipLaunchKernelGGL(xlate_tree_kernel, dim3(grid1((long)L.n * 15)), dim3(256), 0, s, L.hoct.as<int>(), L.doct.as<int>(), L.n, son_h.as<int>(), nbor_h.as<int>(),
                         father_h.as<int>(), perm.as<int>(), ncoarse, ngh, ngd, son_d, nbor_d, father_d, stat);
    }
    LCHK(hipGetLastError());
    LCHK(hipStreamSynchronize(s));
    return hipSuccess;
  }
#undef LCHK
};

}  // namespace amrlayout
}  // namespace ramses_amd
