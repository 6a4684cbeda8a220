// hydro_sweep.hip -- the Godunov sweep of one fully refined level brick on
// MI355X (gfx950): set_unew + godunov_fine/godfine1 + unsplit fused in one
// kernel (reference: hydro/godunov_fine.f90:5-130,486-911, hydro/umuscl.f90).
//
// Design (see DESIGN.md "Godunov sweep kernel"):
//  * a workgroup owns a (64-4) x (BY-4) column tile in (x,y) and MARCHES along
//    z; one wavefront = one y row of 64 x-columns, so every HBM access of a
//    wave is a contiguous 512 B row segment;
//  * the primitive planes c-1, c, c+1 live in an LDS ring (y/x neighbours and
//    the thread's own z neighbours are LDS reads); the traced +y state and the
//    y flux cross waves through LDS; the traced +x state and the x flux cross
//    lanes with wavefront DPP shifts (no LDS); the z direction stays in
//    registers (previous plane's +z state, z flux, partial update);
//  * every cell is converted to primitives once, traced once, and every
//    interface flux is computed once per tile (the reference recomputes a 6^3
//    stencil per 2^3 oct: 27x load and ~8x flop redundancy);
//  * halo rows exit early by role (wave-uniform), so the 2-cell ghost ring
//    costs ctoprim+trace only;
//  * uold is read once from HBM (+ tile halo and one L2 re-read) and unew
//    written once: 80 B per cell update algorithmic HBM traffic.
//
// Compiled twice: strict (-ffp-contract=off, reference operation order,
// bit-identical to the reference) and fast (-DRAMSES_AMD_FAST: FMA
// contraction, rcp/rsq-based division and square root, fused LLF).
#include <hip/hip_runtime.h>

#include "hydro_core.hpp"
#include "sweep_args.hpp"

namespace ramses_amd {

#ifdef RAMSES_AMD_FAST
#define SWEEP_NS fastmode
#else
#define SWEEP_NS strictmode
#endif

namespace SWEEP_NS {

constexpr int BX = 64;   // lanes along x = one wavefront
#ifndef TILE_SWEEP_BY
#define TILE_SWEEP_BY 12   // rows of a workgroup of the sweep of a level in tiles (8: two waves per SIMD, 256 VGPRs)
#endif

// LDS plane of NV doubles per column: [n][ty][tx]; NV = rho, u, v, w, P + passive scalars
template <int BY, int NV>
struct Plane {
  double v[NV][BY][BX];
};

// The LDS of a workgroup.  The primitives of plane c-1 are read by their own column only (the z slope), so unless the 27-point
// slope is in use the ring holds TWO planes: plane c+1 takes the slot of plane c-1 once the thread has taken its z slope
// (nobody else reads that slot between the barrier of iteration c-1 and the one of iteration c).  PARK (the sweep of a level
// in tiles, and the 12-row kernels with gravity: both spill otherwise): the partial update of plane c-1 that the full rows carry
// across the trace of plane c -- u + x flux difference and the own -y flux -- waits in LDS instead of in 20 VGPRs; the y slots
// then hold the rows that use them (1 .. BY-2) only.
template <int ST, int BY, int NV, bool MASK, bool GRAV>
struct Lds {
  static constexpr int RING = (ST == 3) ? 3 : 2;
#ifndef SWEEP_PARK_PLAIN
#ifdef RAMSES_AMD_FAST
#define SWEEP_PARK_PLAIN 1     // the fast 12-row kernels park too: with the plane held in registers (KEEP) 3.23 -> 3.08 ms at 512^3
#else
#define SWEEP_PARK_PLAIN 0     // (the strict ones do not: 5.40 -> 5.50 ms; profiles/r06_ab_sweep.txt)
#endif
#endif
  static constexpr bool PARK = MASK || ((GRAV || SWEEP_PARK_PLAIN) && BY == 12);
  static constexpr int MR = PARK ? BY - 2 : BY;          // rows of a y slot plane
  static constexpr int M0 = PARK ? 1 : 0;                // first row that owns a slot
  static constexpr size_t q_off = 0;
  static constexpr size_t m_off = q_off + RING * sizeof(Plane<BY, NV>);
  static constexpr size_t park_off = m_off + 2 * sizeof(Plane<MR, NV>);
  static constexpr size_t mask_off = park_off + (PARK ? sizeof(double) * 2 * NV * (BY - 4) * BX : 0);   // MASK: [3][BY][BX] status bytes
  static constexpr size_t sloc_off = mask_off + (MASK ? 3 * BY * BX : 0);                                // MASK: [BY][BX] lane part of the cell index
  static constexpr size_t bytes = sloc_off + (MASK ? 4 * BY * BX : 0);
};

// wavefront shift by one lane: lane i receives the value of lane i-1 (shr) or
// i+1 (shl); the edge lane keeps its own value (a halo lane, never stored).
__device__ __forceinline__ double wave_shr1(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(lo, lo, 0x138, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(hi, hi, 0x138, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_shl1(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(lo, lo, 0x130, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(hi, hi, 0x130, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}

// Row roles.  One wave = one tile row, and what a row has to produce depends
// only on its position in the tile, so each role gets its own straight-line
// instantiation of the marching loop (no per-plane role branches, no dead
// results kept alive); all of them execute the same barriers.
//   ROLE_HALO    : row 0        primitives only
//   ROLE_LOW     : row 1        + slopes, the traced +y state (left state of row 2's y flux)
//   ROLE_FULL    : rows 2..BY-3 everything, and the update
//   ROLE_HIGH    : row BY-2     + slopes, the y flux through its -y face (the +y face flux of row BY-3)
//   ROLE_HALO_HI : row BY-1     primitives only
enum { ROLE_HALO = 0, ROLE_LOW = 1, ROLE_HIGH = 2, ROLE_FULL = 3, ROLE_HALO_HI = 4 };
// (Handing row BY-2's y flux to the otherwise idle wave of row BY-1 paid with two barriers per plane only; with one
// barrier it costs in both builds -- measured, profiles/r02_ab_sweep.txt -- and is gone.)

// Raw buffer access: one scalar resource descriptor per variable (base of the
// variable's brick), a wave-uniform byte offset of the plane (soffset) and one
// 32-bit lane byte offset of the column, so that all address arithmetic of the
// marching loop is scalar.  A lane offset beyond num_records makes the hardware
// drop that lane's store: masked lanes and masked iterations need no branch,
// every memory instruction of the loop is issued unconditionally, and the
// compiler's vmcnt bookkeeping never has to wait for a store to be acknowledged
// before it can use a prefetched load.
typedef unsigned int v2u32 __attribute__((ext_vector_type(2)));
constexpr unsigned BUF_RANGE = 0x7fffffffu;     // lane offsets below this are in range
constexpr unsigned BUF_OOB = 0xffffffffu;       // dropped by the range check
#ifndef SWEEP_STORE_AUX
#define SWEEP_STORE_AUX 0     // cache policy of the stores of unew (gfx950: 1 sc0, 2 nt, 16 sc1); A/B knob, profiles/r06_store_policy.txt
#endif
__device__ __forceinline__ double plane_load(const double *var_base, unsigned plane_bytes, unsigned off) {
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(var_base), 0, BUF_RANGE, 0x00020000);
  return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, off, plane_bytes, 0));
}
__device__ __forceinline__ void plane_store(double *var_base, unsigned plane_bytes, unsigned off, double x) {
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(var_base, 0, BUF_RANGE, 0x00020000);
  __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u32, x), r, off, plane_bytes, SWEEP_STORE_AUX);
}

__device__ __forceinline__ int wave_shr1_i(int v) { return __builtin_amdgcn_update_dpp(v, v, 0x138, 0xf, 0xf, false); }
__device__ __forceinline__ int wave_shl1_i(int v) { return __builtin_amdgcn_update_dpp(v, v, 0x130, 0xf, 0xf, false); }
// status byte of a cell: lane part and wave-uniform part of the cell index; a lane without a tile (index beyond ncell) reads 0
__device__ __forceinline__ int stat_load(const unsigned char *base, unsigned ncell, unsigned lane_cell, unsigned plane_cell) {
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char *>(base), 0, ncell, 0x00020000);
  return (int)__builtin_amdgcn_raw_buffer_load_b8(r, lane_cell, plane_cell, 0);
}
// the tile directory entry of (tile column of the lane, tile plane): col and plane in ints
__device__ __forceinline__ int dir_load(const int *base, unsigned plane_ints, unsigned col) {
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<int *>(base), 0, BUF_RANGE, 0x00020000);
  return (int)__builtin_amdgcn_raw_buffer_load_b32(r, col * 4u, plane_ints * 4u, 0);
}

// MASK: the sweep of a level of a resident AMR run IN PLACE on the device's cell vectors (see SweepArgs::stat / dir / work):
// the level is stored in tiles of 32 x 4 x 4 octs, a lane finds the cell of its (plane, column) through the tile directory
// (one 4-byte load per 8 planes, issued four planes ahead; 256-byte runs of a variable along x inside a tile), the status byte of
// the cell says whether it is refined (fluxes reset), updated (stored) or a ghost (interpolated: fluxes filed for the coarser level)
template <int ST, int RS, int BY, bool GRAV, int SCHEME, int NV, int ROLE, bool MASK>
__device__ __forceinline__ void sweep_march(const SweepArgs &A, unsigned char *smem_raw) {
  const bool DXPOW2 = A.pow2 != 0;   // uniform
  typedef Lds<ST, BY, NV, MASK, GRAV> L;
  constexpr int RING = L::RING;
  constexpr bool PARK = L::PARK;
  constexpr int M0 = L::M0;
  Plane<BY, NV> *qring = reinterpret_cast<Plane<BY, NV> *>(smem_raw + L::q_off);  // [RING] primitives of planes c-1, c (, c+1)
  Plane<L::MR, NV> *mring = reinterpret_cast<Plane<L::MR, NV> *>(smem_raw + L::m_off);   // [2] +y traced state / y flux slots, by plane parity
  double (*park)[BY - 4][BX] = reinterpret_cast<double (*)[BY - 4][BX]>(smem_raw + L::park_off);   // PARK: [2 NV] of the full rows

  const int tx = threadIdx.x, ty = threadIdx.y;
  const HydroConst &P = A.P;

  // ---- tile decode (XCD-aware: block b runs on XCD b%8; give each XCD a
  // contiguous run of tiles so the halo re-reads of neighbouring tiles hit
  // the same L2) -------------------------------------------------------------
  // a launch covers up to 6 boxes of tiles x planes (one for a whole-brick or
  // interior sweep, six for the boundary shell); find this block's box (uniform)
  const int hb = blockIdx.x;
  int bi = 0;
#pragma unroll
  for (int i = 1; i < 6; i++)
    if (i < A.nbox && hb >= A.box[i].first) bi = i;
  const SweepBox &B = A.box[bi];
  // XCD-aware: block b runs on XCD b%8; inside its box a block is handed a tile so that each XCD works on a contiguous
  // run of them (neighbouring tiles then re-read each other's halo rows and columns in ONE L2) -- box by box, because the
  // blocks of different boxes cost very differently (the shell's 2-plane slabs next to its 32-plane columns) and every
  // XCD has to get its share of each kind: one run over the whole shell launch left five XCDs with the slabs (shell
  // 1.29 -> 1.04 ms at 512^3, the overlapped step 4.23 -> 4.00 ms)
  int lb = hb - B.first;
  {
    const int nxcd = 8;
    const int cnt = (bi + 1 < A.nbox ? A.box[bi + 1].first : (int)gridDim.x) - B.first;
    if (cnt % nxcd == 0) {
      const int per = cnt / nxcd;
      lb = (lb % nxcd) * per + lb / nxcd;
    }
  }
#ifndef SWEEP_TILE_YFAST
#define SWEEP_TILE_YFAST 0     // A/B knob: tiles of a box in y-first order (profiles/r06_store_policy.txt)
#endif
#if SWEEP_TILE_YFAST
  const int tiy = B.ty0 + lb % B.nty;
  const int tix = B.tx0 + (lb / B.nty) % B.ntx;
#else
  const int tix = B.tx0 + lb % B.ntx;
  const int tiy = B.ty0 + (lb / B.ntx) % B.nty;
#endif
  const int tiz = lb / (B.ntx * B.nty);
  int x0 = tix * (BX - 4);
  int y0 = tiy * (BY - 4);
  int z0 = B.zlo + tiz * B.zchunk;
  int z1 = min(z0 + B.zchunk, B.zhi);
  if (MASK) {
    // the launch's work list (the host put it in the order the XCDs should see it)
    const int4 w = reinterpret_cast<const int4 *>(A.work)[hb];
    x0 = w.x; y0 = w.y; z0 = w.z; z1 = w.w;
  }

  // ---- this thread's column ------------------------------------------------
  const int xu = x0 - 2 + tx;  // unwrapped interior coordinate
  const int yu = y0 - 2 + ty;
  int xi, yi;
  if (A.ng == 0) {
    xi = xu < 0 ? xu + A.nx : (xu >= A.nx ? xu - A.nx : xu);
    xi = xi >= A.nx ? xi % A.nx : xi;
    yi = yu < 0 ? yu + A.ny : (yu >= A.ny ? yu - A.ny : yu);
    yi = yi >= A.ny ? yi % A.ny : yi;
  } else {
    xi = min(max(xu, -A.ng), A.nx + A.ng - 1) + A.ng;
    yi = min(max(yu, -A.ng), A.ny + A.ng - 1) + A.ng;
  }
  // column offset inside a plane as a 32-bit lane value; plane/variable bases
  // are wave-uniform (scalar base + 32-bit lane offset addressing)
  // (byte offset < 2 GB per plane, checked by the launcher)
  const unsigned colb = (unsigned)(xi + yi * (int)A.pitch_y) * 8u;
  // MASK (the level in tiles, periodic box, ng = 0): the lane's tile column in a plane of the directory, its part of the
  // cell index inside a tile (octant bits of x and y at stride ngd, oct column and oct row) and of the octant position
  // Everything about the tiles is wave-uniform except the lane's place inside one: a row of 64 lanes lies in one tile row and
  // in (at most) two neighbouring tiles along x, so the directory entries are SCALAR loads into scalar registers, the lane
  // picks one of two with a constant lane mask, and the lane's own part of the cell index is parked in LDS -- the marching
  // loop is at its register limit, and a vector register spilled to scratch costs a full vmcnt(0) drain per use.
  int tyu = 0, yis = 0, xa = 0, tA = 0, tB = 0, drow = 0;
  bool inB = false;
  unsigned *sloc = reinterpret_cast<unsigned *>(smem_raw + L::sloc_off);   // MASK: [BY][BX] lane part of the cell index (bytes)
  if (MASK) {
    tyu = __builtin_amdgcn_readfirstlane(ty);
    const int ys = y0 - 2 + tyu;
    yis = ys < 0 ? ys + A.ny : (ys >= A.ny ? ys - A.ny : ys);
    const int xs = x0 - 2;
    xa = xs < 0 ? xs + A.nx : xs;                      // column of lane 0 (x0 < nx)
    tA = xa >> 6; tB = tA + 1 < A.ntx ? tA + 1 : 0;
    inB = (xa & 63) + tx >= 64;
    drow = A.ntx * (yis >> 3);
    const int lind = (xi & 1) + 2 * (yis & 1);
    sloc[ty * BX + tx] = ((unsigned)((long)lind * A.ngd) + (unsigned)(((xi >> 1) & (TILE_OX - 1)) + TILE_OX * ((yis >> 1) & (TILE_OY - 1)))) * 8u;
  }
  const double *__restrict__ uold = A.uold;
  double *__restrict__ unew = A.unew;
  const double *__restrict__ grav = A.grav;

#ifndef SWEEP_LATE_BASE
#define SWEEP_LATE_BASE 0      // (measured on MI355X, round 6: 512^3 fast 3.237 -> 3.272 ms, the tiled 256^3 level unchanged: the wait at the x flux is not what costs)
#endif
  constexpr bool LATE = SWEEP_LATE_BASE && NV == 5;   // (the passive-scalar fix of NV > 5 wants the old state in phase A)
  // KEEP (the fast build of the sweep of a level in tiles: the instantiations with ten registers to spare): the
  // conservative state of plane c+1 is held from its arrival to the x flux of the NEXT iteration, where the update of that plane
  // starts from it -- on a level that starts from uold (base_uold) the re-read of the plane disappears: 5 of the 11 loads a full
  // lane issues per plane.  That kernel is bound by L2 misses, not by instruction issue (profiles/r06_tile_sweep_pmc.txt).
#ifndef SWEEP_KEEP_PLAIN
#define SWEEP_KEEP_PLAIN 1     // the plain fast 12-row kernel holds the plane too instead of re-reading it from L2 (profiles/r06_ab_sweep.txt)
#endif
#ifndef SWEEP_KEEP_STRICT
#define SWEEP_KEEP_STRICT 0    // (A/B knob: the strict build holds the plane too)
#endif
#if defined(RAMSES_AMD_FAST) || SWEEP_KEEP_STRICT
#ifndef SWEEP_KEEP_LLF_ONLY
#define SWEEP_KEEP_LLF_ONLY 0   // A/B knob: only the LLF kernels hold the plane (the others are at the register limit without it)
#endif
#ifndef SWEEP_KEEP_NOT_HLLC
#define SWEEP_KEEP_NOT_HLLC 1   // the HLLC kernels do not hold the plane: with the fused fast HLLC flux they are at the register limit (hydro_core.hpp hllc_flux_fast)
#endif
  constexpr bool KEEP = (MASK || (SWEEP_KEEP_PLAIN && BY == 12)) && NV == 5 && !LATE && (!SWEEP_KEEP_LLF_ONLY || RS == RIEMANN_LLF) &&
                        (!SWEEP_KEEP_NOT_HLLC || RS != RIEMANN_HLLC);
#else
  constexpr bool KEEP = false;
#endif
  constexpr bool r_trace = ROLE == ROLE_LOW || ROLE == ROLE_HIGH || ROLE == ROLE_FULL;
  constexpr bool r_fxz = ROLE == ROLE_FULL;
  const bool r_upd = r_fxz && (tx >= 2) && (tx <= BX - 3) && (xu < A.nx) && (yu < A.ny);
  const unsigned colb_upd = r_upd ? colb : BUF_OOB;   // lanes that own no cell store nowhere

  const double dtdx = A.dt / A.dx;
  const double dtxhalf = A.dt * 0.5;

  auto plane_off = [&](int p) -> unsigned {   // uniform byte offset of plane p inside a variable
    int pz;
    if (A.ng == 0) { pz = p < 0 ? p + A.nz : (p >= A.nz ? p - A.nz : p); }
    else { pz = p + A.ng; }
    return (unsigned)pz * (unsigned)(A.pitch_z * 8);
  };
  auto wrap_z = [&](int p) -> int { return p < 0 ? p + A.nz : (p >= A.nz ? p - A.nz : p); };
  // MASK: a cell's index in a cell vector = tile base (scalar: sdir, by z-tile parity and x-tile A / B) + the lane's part
  // (LDS) + the plane's part (scalar, rides in the scalar offset of the buffer instructions).  TILE_VOID where the level has
  // no tile: with the lane's part still beyond every buffer range, so such a lane loads zeros and stores nothing.
  constexpr unsigned TILE_VOID = 0x80000000u;
  unsigned sA0 = TILE_VOID, sA1 = TILE_VOID, sB0 = TILE_VOID, sB1 = TILE_VOID;    // (four scalars: an indexed local array would live in scratch)
  auto tile_of_plane = [&](int p) -> int { return wrap_z(p) >> 3; };
  auto tile_set = [&](int tz) {
    // (through the constant address space: the directory is not written during the launch, and only such a load is scalar)
    typedef const int __attribute__((address_space(4))) *cdir_p;
    const cdir_p row = (cdir_p)(A.dir + (long)tz * (A.ntx * A.nty) + drow);
    const int ra = row[tA], rb = row[tB];
    const unsigned a = ra < 0 ? TILE_VOID : (unsigned)ra * 8u, b = rb < 0 ? TILE_VOID : (unsigned)rb * 8u;
    if (tz & 1) { sA1 = a; sB1 = b; } else { sA0 = a; sB0 = b; }
  };
  auto zpart = [&](int p) -> unsigned {          // cells
    const int pz = wrap_z(p);
    return (unsigned)((long)(pz & 1) * 4 * A.ngd) + (unsigned)(TILE_OX * TILE_OY * ((pz >> 1) & (TILE_OZ - 1)));
  };
  auto tbp = [&](int p) -> unsigned {
    const int par = tile_of_plane(p) & 1;
    const unsigned a = par ? sA1 : sA0, b = par ? sB1 : sB0;
    return (inB ? b : a) + sloc[ty * BX + tx];
  };
  // (MASK: plane p through the tiles; else plane offset + column)
  // (MASK: two variables share a buffer descriptor -- the odd one rides in the scalar offset: 16 scalar registers the
  //  masked instantiation does not have; pitch_var * 8 + the plane part < 2^32, checked by the launcher)
  const unsigned odd_var = (unsigned)(A.pitch_var * 8);
  auto load_u = [&](int p, double (&u)[NV]) {
    const unsigned pb = MASK ? zpart(p) * 8u : plane_off(p), off = MASK ? tbp(p) : colb;
#pragma unroll
    for (int n = 0; n < NV; n++)
      u[n] = MASK ? plane_load(uold + (long)(n & ~1) * A.pitch_var, pb + (n & 1) * odd_var, off) : plane_load(uold + (long)n * A.pitch_var, pb, off);
  };
  // MASK: the state the update starts from: unew, in place -- it holds what the finer level owes to this one (:752-790) -- or, on a
  // level without finer octs (set_unew has just made unew = uold there), uold again: the planes this workgroup read two iterations
  // ago, from L2 instead of a second stream from HBM
  const double *__restrict__ bsrc = A.base_uold ? uold : unew;
  auto load_base = [&](int p, double (&u)[NV]) {
    const unsigned pb = zpart(p) * 8u, off = tbp(p);
#pragma unroll
    for (int n = 0; n < NV; n++) u[n] = plane_load(bsrc + (long)(n & ~1) * A.pitch_var, pb + (n & 1) * odd_var, off);
  };
  int ok_zlo = 0;   // MASK: plane c-1's status byte of this column
  int spre = 0;     // MASK: plane c+1's status byte, on its way
  unsigned char *smask = smem_raw + L::mask_off;   // MASK: [3][BY][BX] status bytes of planes c-1, c, c+1, by plane mod 3
  auto load_g = [&](int p, double (&g)[3]) {
    if (GRAV) {
      const unsigned pb = MASK ? zpart(p) * 8u : plane_off(p), off = MASK ? tbp(p) : colb;
#pragma unroll
      for (int d = 0; d < 3; d++) g[d] = plane_load(grav + (long)d * A.pitch_var, pb, off);
    } else {
      g[0] = g[1] = g[2] = 0.0;
    }
  };

  // ---- register state carried along z --------------------------------------
  double qmz[NV];                 // qm along z of plane c-1 (state on its +z face)
  double fzlo[NV];                // z flux through the -z face of plane c-1
  double upre[NV], gpre[3];       // prefetch: plane c+1 on entry of iteration c
  double rold = 0.0, sold[NV > 5 ? NV - 5 : 1];   // uold density / scalars of plane c-1 (NV>5 only)
  double ukeep[NV];               // KEEP: the conservative state of plane c
#pragma unroll
  for (int n = 0; n < NV; n++) ukeep[n] = 0.0;

  // ring slots of planes c-1, c, c+1
  int sa = 0, sb = 1, sc = 2;

  // prologue: primitives of planes z0-2 -> slot sa, z0-1 -> slot sb; c starts at z0-1
  // MASK: the z-tile of plane z0-2 and the next one (the planes in flight -- c-1 .. c+2 -- never span more than two)
  if (MASK) {
    const int t0 = tile_of_plane(z0 - 2), t1 = t0 + 1 < A.ntz ? t0 + 1 : 0;
    tile_set(t0);
    tile_set(t1);
    __syncthreads();      // (sloc is read by its own thread only; the barrier orders the LDS write for the compiler's sake)
    if (r_trace) spre = stat_load(A.stat, (unsigned)A.pitch_var, tbp(z0 - 1) >> 3, zpart(z0 - 1));
  }
  {
    double u[NV], g[3], q[NV];
    load_u(z0 - 2, u); load_g(z0 - 2, g);
    ctoprim_cell<NV, GRAV>(u, g, dtxhalf, P, q);
#pragma unroll
    for (int n = 0; n < NV; n++) qring[sa].v[n][ty][tx] = q[n];
    load_u(z0 - 1, u); load_g(z0 - 1, g);
    if (KEEP) {
#pragma unroll
      for (int n = 0; n < NV; n++) ukeep[n] = u[n];
    }
    ctoprim_cell<NV, GRAV>(u, g, dtxhalf, P, q);
#pragma unroll
    for (int n = 0; n < NV; n++) qring[sb].v[n][ty][tx] = q[n];
    load_u(z0, upre); load_g(z0, gpre);
#pragma unroll
    for (int n = 0; n < NV; n++) { qmz[n] = 1.0; fzlo[n] = 0.0; }
  }
  // Enter the loop with no load in flight: otherwise the loop header inherits
  // "prefetch pending" from this path and waits (in issue order) behind the
  // stores of the previous iteration on every trip.
  __builtin_amdgcn_s_waitcnt(0x0f70);   // vmcnt(0)
  __syncthreads();

  // Neighbour columns by constant offsets from the thread's own LDS address (one address register,
  // immediate offsets).  Lanes 0 and 63 and a tile's first and last row then read a neighbouring row,
  // variable or nothing at all (out-of-range LDS reads return 0): whatever they compute from it stays in
  // the two halo columns / rows, which are never stored (rows 0 and BY-1 take no slopes, lanes 0, 1, 62, 63
  // own no cell, and the x flux an owned cell uses reaches two lanes at most).
  const int txm = tx - 1, txp = tx + 1;
  const int tym = ty - 1, typ = ty + 1;

  // ---- ONE barrier per plane --------------------------------------------------------------
  // The +y traced state and the y flux share ONE LDS slot per row, double-buffered by plane
  // parity: slot M[c&1][ty] is written by row ty with its +y state before the barrier of
  // iteration c, read after it by row ty+1, which then overwrites it with the flux through
  // that face; row ty picks the flux up after the NEXT barrier, when it finishes plane c-1
  // -> c.  Nobody else touches the slot, so the second barrier of the two-barrier loop (and
  // the lock-step of the heavy waves it enforced) is gone.  Same operations in the same
  // order per cell: ((u + (fx- - fx+)) + (fy- - fy+)) + (fz- - fz+).
  double partx[NV];                      // u + x flux difference of plane c-1
  double fyown[NV];                      // y flux through the -y face of plane c-1 (computed by this row; its copy
                                         // in slot ty-1 belongs to row ty-1, which reuses the slot without a barrier)
  double rnew = 0.0, snew[NV > 5 ? NV - 5 : 1];
  double bcar[NV];                       // LATE: the state the update of plane c-1 starts from
#pragma unroll
  for (int n = 0; n < NV; n++) { partx[n] = 0.0; fyown[n] = 0.0; bcar[n] = 0.0; }
  if (PARK && r_fxz) {
#pragma unroll
    for (int n = 0; n < 2 * NV; n++) park[n][ty - 2][tx] = 0.0;
  }

  for (int c = z0 - 1; c <= z1; c++) {
    Plane<L::MR, NV> &M = mring[c & 1];
    Plane<L::MR, NV> &Mprev = mring[(c & 1) ^ 1];
    // ---- phase A: plane c+1 arrives; trace plane c; x and z fluxes ------------------
    double qc[NV];
    ctoprim_cell<NV, GRAV>(upre, gpre, dtxhalf, P, qc);
    // plane c+1 goes into the ring: into its own slot (RING 3), or into the slot of plane c-1 as soon as this thread has taken
    // its z slope from it (RING 2; the rows that take no slopes have nothing to wait for)
    if (RING == 3 || !r_trace) {
#pragma unroll
      for (int n = 0; n < NV; n++) qring[RING == 3 ? sc : sa].v[n][ty][tx] = qc[n];
    }
    double ucur[NV];
    if (r_fxz && !LATE) {
      if (KEEP && (!MASK || A.base_uold)) {
#pragma unroll
        for (int n = 0; n < NV; n++) ucur[n] = ukeep[n];          // plane c, held since it arrived
      } else if (MASK) load_base(c, ucur);
      else load_u(c, ucur);
    }
    if (KEEP && r_fxz) {
#pragma unroll
      for (int n = 0; n < NV; n++) ukeep[n] = upre[n];            // plane c+1, for the next iteration
    }
    int okc = 0, ok_ym = 0;
    if (MASK && r_trace) {
      okc = spre;                  // loaded one plane ahead: nothing at the top of an iteration waits for memory
      smask[((c + 3) % 3 * BY + ty) * BX + tx] = (unsigned char)okc;   // row ty+1 reads it after the barrier (its -y neighbour), row ty-1 one plane later
    }

    double qpy[NV], dz[NV], px[NV];
    if (ST == 3) __syncthreads();  // the 27-point slope reads the neighbours' plane c+1 just written
    if constexpr (r_trace) {
      const Plane<BY, NV> &qs = qring[sb];
      const Plane<BY, NV> &qprev = qring[sa];
      double qb[NV], dq[3][NV];
      if (ST == 3) {
        const Plane<BY, NV> &qnext = qring[sc];
        const int xs[3] = {txm, tx, txp}, ys[3] = {tym, ty, typ};
#pragma unroll
        for (int n = 0; n < NV; n++) {
          double nb[27], d3[3];
#pragma unroll
          for (int dj = 0; dj < 3; dj++)
#pragma unroll
            for (int di = 0; di < 3; di++) {
              nb[di + 3 * dj] = qprev.v[n][ys[dj]][xs[di]];
              nb[di + 3 * dj + 9] = qs.v[n][ys[dj]][xs[di]];
              nb[di + 3 * dj + 18] = qnext.v[n][ys[dj]][xs[di]];
            }
          qb[n] = nb[13];
          slope3_var(nb, d3);
          dq[0][n] = d3[0]; dq[1][n] = d3[1]; dq[2][n] = d3[2];
        }
      } else if (ST == 4 || ST == 5 || ST == 6) {
        // the NDIM=1 slope types (embedded 1-D problems: ny = nz = 1, the transverse differences vanish)
#pragma unroll
        for (int n = 0; n < NV; n++) qb[n] = qs.v[n][ty][tx];
        const double dc0 = qb[1] * A.dt / A.dx, dc1 = qb[2] * A.dt / A.dx, dc2 = qb[3] * A.dt / A.dx;
#pragma unroll
        for (int n = 0; n < NV; n++) {
          dq[0][n] = slope1_1d<ST>(qs.v[n][ty][txm], qb[n], qs.v[n][ty][txp], dc0, n);
          dq[1][n] = slope1_1d<ST>(qs.v[n][tym][tx], qb[n], qs.v[n][typ][tx], dc1, n);
          dq[2][n] = slope1_1d<ST>(qprev.v[n][ty][tx], qb[n], qc[n], dc2, n);
        }
      } else {
#pragma unroll
        for (int n = 0; n < NV; n++) {
          qb[n] = qs.v[n][ty][tx];
          dq[0][n] = slope1<ST>(qs.v[n][ty][txm], qb[n], qs.v[n][ty][txp], P);
          dq[1][n] = slope1<ST>(qs.v[n][tym][tx], qb[n], qs.v[n][typ][tx], P);
          dq[2][n] = slope1<ST>(qprev.v[n][ty][tx], qb[n], qc[n], P);
        }
      }
      if (RING == 2) {
#pragma unroll
        for (int n = 0; n < NV; n++) qring[sa].v[n][ty][tx] = qc[n];
      }
      double qm[3][NV], qp[3][NV];
      if (SCHEME == 0) {
        trace3d_cell<NV>(qb, dq, dtdx, dtdx, dtdx, P, qm, qp);
      } else {
        const double cc = ctoprim_sound(qb[0], qb[4], P);
        tracexyz_cell<NV>(qb, dq, cc, dtdx, dtdx, dtdx, P, qm, qp);
      }
#pragma unroll
      for (int n = 0; n < NV; n++) M.v[n][ty - M0][tx] = qm[1][n];
#pragma unroll
      for (int n = 0; n < NV; n++) qpy[n] = qp[1][n];
      if constexpr (r_fxz) {
        double qL[NV], fx[NV], fz[NV];
#pragma unroll
        for (int n = 0; n < NV; n++) qL[n] = wave_shr1(qm[0][n]);  // +x state of column tx-1
        scaled_interface_flux<RS, NV, 0, !MASK>(qL, qp[0], P, A.dt, A.dx, A.rdx, dtdx, DXPOW2, fx);
        // z flux through the face between planes c-1 and c
        scaled_interface_flux<RS, NV, 2, !MASK>(qmz, qp[2], P, A.dt, A.dx, A.rdx, dtdx, DXPOW2, fz);
        if (MASK) {
          // hydro/godunov_fine.f90:720-747: the flux through a face is reset when the cell on either side is refined
          const int s_xm = wave_shr1_i(okc);
          const bool zx = ((okc | s_xm) & CELL_REFINED) != 0, zz = ((okc | ok_zlo) & CELL_REFINED) != 0;
#pragma unroll
          for (int n = 0; n < NV; n++) { fx[n] = zx ? 0.0 : fx[n]; fz[n] = zz ? 0.0 : fz[n]; }
        }
        double fxh[NV];
#pragma unroll
        for (int n = 0; n < NV; n++) {
          qmz[n] = qm[2][n];
          dz[n] = fzlo[n] - fz[n];          // z flux difference of plane c-1
          fzlo[n] = fz[n];
          fxh[n] = wave_shl1(fx[n]);        // -x face flux of column tx+1
          // (LATE: the state the update starts from joins in phase B of the next iteration -- a whole trace after its load)
          px[n] = LATE ? (fx[n] - fxh[n]) : ucur[n] + (fx[n] - fxh[n]);
        }
        if (NV > 5 && !MASK) {
          rnew = ucur[0];
#pragma unroll
          for (int n = 5; n < NV; n++) snew[n - 5] = ucur[n];
        }
      }
    }
    // prefetch plane c+2 after the register peak of the trace and flux phase (still ~1 us ahead of its use)
    __builtin_amdgcn_sched_barrier(0);
    {
      const int pn = min(c + 2, z1 + 1);
      if (MASK) {
        // the directory entries of the NEXT z-tile: loaded (scalar) when the prefetch is half way through this one, into the
        // registers of the tile before it, which no plane in flight uses any more
        const int lp = wrap_z(pn) & 7;
        const int tn = tile_of_plane(pn) + 1 < A.ntz ? tile_of_plane(pn) + 1 : 0;
        if (lp == 4) tile_set(tn);
      }
      load_u(pn, upre); load_g(pn, gpre);
      if (MASK && r_trace) { const int ps = min(c + 1, z1 + 1); spre = stat_load(A.stat, (unsigned)A.pitch_var, tbp(ps) >> 3, zpart(ps)); }
    }
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();  // the one barrier: +y states of plane c and y fluxes of plane c-1 visible

    // ---- phase B: y flux of plane c; finish plane c-1 --------------------------------
    double fy[NV];
    if constexpr (ROLE == ROLE_FULL || ROLE == ROLE_HIGH) {
      double qL[NV];
#pragma unroll
      for (int n = 0; n < NV; n++) qL[n] = M.v[n][tym - M0][tx];
      scaled_interface_flux<RS, NV, 1, !MASK>(qL, qpy, P, A.dt, A.dx, A.rdx, dtdx, DXPOW2, fy);
      if (MASK) {
        ok_ym = smask[((c + 3) % 3 * BY + tym) * BX + tx];
        const bool zy = ((okc | ok_ym) & CELL_REFINED) != 0;
#pragma unroll
        for (int n = 0; n < NV; n++) fy[n] = zy ? 0.0 : fy[n];
      }
      // the flux through this row's -y face is the +y face flux of row ty-1: into ITS slot
#pragma unroll
      for (int n = 0; n < NV; n++) M.v[n][tym - M0][tx] = fy[n];
    }
    if constexpr (r_fxz) {
      // plane c-1: its x part and own -y flux were kept in registers, the +y face flux was left
      // in this row's slot of the other buffer by row ty+1 before this iteration's barrier.
      // (The first two iterations of a chunk produce values from the not yet primed pipeline;
      // they are computed and dropped by the store's range check.)
      double un[NV], fyh[NV];
#pragma unroll
      for (int n = 0; n < NV; n++) {
        fyh[n] = Mprev.v[n][ty - M0][tx];
        const double pxn = PARK ? park[n][ty - 2][tx] : partx[n];
        const double fyn = PARK ? park[NV + n][ty - 2][tx] : fyown[n];
        const double part = (LATE ? bcar[n] + pxn : pxn) + (fyn - fyh[n]);
        un[n] = part + dz[n];
      }
      if (NV > 5 && !MASK) {
        // set_uold's passive-scalar fix near the density floor
        // (hydro/godunov_fine.f90:176-190), fused: the kernel's output is the new uold
        // (MASK: the kernel's output is unew; set_uold of the resident level applies the fix, csrc/capi_amr.hip lvl_set_uold)
        if (rold < P.smallr && un[0] > rold) {
#pragma unroll
          for (int n = 5; n < NV; n++) un[n] = sold[n - 5] * dmaxd(un[0], P.smallr) / P.smallr;
        } else if (un[0] < P.smallr && rold > un[0]) {
#pragma unroll
          for (int n = 5; n < NV; n++) un[n] = sold[n - 5] * P.smallr / dmaxd(rold, P.smallr);
        }
        rold = rnew;
#pragma unroll
        for (int n = 5; n < NV; n++) sold[n - 5] = snew[n - 5];
      }
#pragma unroll
      for (int n = 0; n < NV; n++) {
        if (PARK) { park[n][ty - 2][tx] = px[n]; park[NV + n][ty - 2][tx] = fy[n]; }
        else { partx[n] = px[n]; fyown[n] = fy[n]; }
      }
      {
        const unsigned pb = MASK ? zpart(c - 1) * 8u : plane_off(c - 1);
        const unsigned so = (c >= z0 + 1) ? (MASK ? ((r_upd && (ok_zlo & CELL_OWNED)) ? tbp(c - 1) : BUF_OOB) : colb_upd) : BUF_OOB;
#pragma unroll
        for (int n = 0; n < NV; n++) {
          if (MASK) plane_store(unew + (long)(n & ~1) * A.pitch_var, pb + (n & 1) * odd_var, so, un[n]);
          else plane_store(unew + (long)n * A.pitch_var, pb, so, un[n]);
        }
      }
      // LATE: the state the update of plane c starts from (MASK: unew, in place -- another array, an HBM miss; else the plane of
      // uold this workgroup read two iterations ago), requested now and used in phase B of the next iteration: at the top of
      // phase A it was due at the x flux, and every wave of the workgroup sat at that wait together
      if (LATE) { if (MASK) load_base(c, bcar); else load_u(c, bcar); }
    }
    // rotate the ring
    if (RING == 3) { const int t = sa; sa = sb; sb = sc; sc = t; }
    else { const int t = sa; sa = sb; sb = t; }
    if (MASK) ok_zlo = okc;
  }
}

template <int ST, int RS, int BY, bool GRAV, int SCHEME, int NV, bool MASK = false>
__global__ __launch_bounds__(BX *BY) void godunov_sweep_kernel(SweepArgs A) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int ty = threadIdx.y;   // wave-uniform
  // the full rows are the critical path between two barriers: let them win the issue
  // arbitration over the light rows (measured: -1.1 % at 512^3)
  if (ty >= 2 && ty <= BY - 3) __builtin_amdgcn_s_setprio(3);
  if (ty == 0) sweep_march<ST, RS, BY, GRAV, SCHEME, NV, ROLE_HALO, MASK>(A, smem_raw);
  else if (ty == BY - 1) sweep_march<ST, RS, BY, GRAV, SCHEME, NV, ROLE_HALO_HI, MASK>(A, smem_raw);
  else if (ty == 1) sweep_march<ST, RS, BY, GRAV, SCHEME, NV, ROLE_LOW, MASK>(A, smem_raw);
  else if (ty == BY - 2) sweep_march<ST, RS, BY, GRAV, SCHEME, NV, ROLE_HIGH, MASK>(A, smem_raw);
  else sweep_march<ST, RS, BY, GRAV, SCHEME, NV, ROLE_FULL, MASK>(A, smem_raw);
}

// ---------------------------------------------------------------------------
// surface pass of a level in tiles (SurfArgs): the fluxes owed to the coarser level
// ---------------------------------------------------------------------------
// index (0-based) of cell (x, y, z) of the level in a cell vector; the layout gave every position this pass asks for a tile
__device__ __forceinline__ long surf_cell(const SurfArgs &A, int x, int y, int z) {
  const int m = 2 * A.no - 1;
  x &= m; y &= m; z &= m;
  const int ox = x >> 1, oy = y >> 1, oz = z >> 1;
  const int t = (ox / TILE_OX) + A.ntx * ((oy / TILE_OY) + A.nty * (oz / TILE_OZ));
  const long c0 = A.dir[t];
  const int ind = (x & 1) + 2 * (y & 1) + 4 * (z & 1);
  return c0 + (ox % TILE_OX) + TILE_OX * ((oy % TILE_OY) + TILE_OY * (oz % TILE_OZ)) + (long)ind * A.ngd;
}
// One interface of direction DIR between the cells lo (left) and hi = lo + e_DIR: the twelve cells the two traces read -- four
// along DIR (lo - 1 .. hi + 1), the four transverse neighbours of each of the two -- are addressed and requested FIRST (the kernel
// is a gather of isolated 8-byte words: what it costs is the latency of dependent loads, so nothing may wait between them), then
// converted (ctoprim), then the slopes and the two traces -- what a lane of the marching kernel does for its cell in phase A --
// and the Riemann flux, scaled like the marching kernel's.
// (slope type 3: the 3 x 3 x 3 neighbourhoods of the two cells -- 36 cells, four slabs of nine along DIR)
template <int RS, int NV, bool GRAV, int SCHEME, int DIR>
__device__ __forceinline__ void surf_interface27(const SurfArgs &A, const int (&lo)[3], double (&fl)[NV]) {
  constexpr int T0 = DIR == 0 ? 1 : 0, T1 = DIR == 2 ? 1 : 2;
  long c[36];
#pragma unroll
  for (int a = 0; a < 4; a++)
#pragma unroll
    for (int u = 0; u < 3; u++)
#pragma unroll
      for (int v = 0; v < 3; v++) {
        int p[3] = {lo[0], lo[1], lo[2]};
        p[DIR] += a - 1; p[T0] += u - 1; p[T1] += v - 1;
        c[a * 9 + u * 3 + v] = surf_cell(A, p[0], p[1], p[2]);
      }
  double q[36][NV];
#pragma unroll
  for (int k = 0; k < 36; k++) {
    double u[NV], g[3];
#pragma unroll
    for (int n = 0; n < NV; n++) u[n] = A.uold[(long)n * A.ncell + c[k]];
#pragma unroll
    for (int d = 0; d < 3; d++) g[d] = GRAV ? A.grav[(long)d * A.ncell + c[k]] : 0.0;
    ctoprim_cell<NV, GRAV>(u, g, A.dt * 0.5, A.P, q[k]);
  }
  const double dtdx = A.dt / A.dx;
  double qL[NV], qR[NV];
#pragma unroll
  for (int w = 0; w < 2; w++) {
    double dq[3][NV], qm[3][NV], qp[3][NV];
#pragma unroll
    for (int n = 0; n < NV; n++) {
      double nb[27], d3[3];
#pragma unroll
      for (int dz = 0; dz < 3; dz++)
#pragma unroll
        for (int dy = 0; dy < 3; dy++)
#pragma unroll
          for (int dx = 0; dx < 3; dx++) {
            const int o[3] = {dx, dy, dz};
            nb[dx + 3 * dy + 9 * dz] = q[(w + o[DIR]) * 9 + o[T0] * 3 + o[T1]][n];
          }
      slope3_var(nb, d3);
      dq[0][n] = d3[0]; dq[1][n] = d3[1]; dq[2][n] = d3[2];
    }
    const double (&qb)[NV] = q[(1 + w) * 9 + 4];
    if (SCHEME == 0) trace3d_cell<NV>(qb, dq, dtdx, dtdx, dtdx, A.P, qm, qp);
    else tracexyz_cell<NV>(qb, dq, ctoprim_sound(qb[0], qb[4], A.P), dtdx, dtdx, dtdx, A.P, qm, qp);
#pragma unroll
    for (int n = 0; n < NV; n++) { if (w == 0) qL[n] = qm[DIR][n]; else qR[n] = qp[DIR][n]; }
  }
  scaled_interface_flux<RS, NV, DIR, false>(qL, qR, A.P, A.dt, A.dx, A.rdx, dtdx, A.pow2 != 0, fl);   // (as the marching kernel of a level in tiles does)
}
template <int ST, int RS, int NV, bool GRAV, int SCHEME, int DIR>
__device__ __forceinline__ void surf_interface(const SurfArgs &A, const int (&lo)[3], double (&fl)[NV]) {
  if constexpr (ST == 3) { surf_interface27<RS, NV, GRAV, SCHEME, DIR>(A, lo, fl); return; }
  constexpr int T0 = DIR == 0 ? 1 : 0, T1 = DIR == 2 ? 1 : 2;
  long c[12];
  // 0..3: along DIR at lo-1, lo, hi, hi+1;  4..7: lo -T0, +T0, -T1, +T1;  8..11: the same of hi
#pragma unroll
  for (int k = 0; k < 4; k++) {
    int p[3] = {lo[0], lo[1], lo[2]};
    p[DIR] += k - 1;
    c[k] = surf_cell(A, p[0], p[1], p[2]);
  }
#pragma unroll
  for (int w = 0; w < 2; w++)
#pragma unroll
    for (int k = 0; k < 4; k++) {
      int p[3] = {lo[0], lo[1], lo[2]};
      p[DIR] += w;
      p[k < 2 ? T0 : T1] += (k & 1) ? 1 : -1;
      c[4 + 4 * w + k] = surf_cell(A, p[0], p[1], p[2]);
    }
  double q[12][NV];
  {
    double u[12][NV], g[12][3];
#pragma unroll
    for (int k = 0; k < 12; k++) {
#pragma unroll
      for (int n = 0; n < NV; n++) u[k][n] = A.uold[(long)n * A.ncell + c[k]];
#pragma unroll
      for (int d = 0; d < 3; d++) g[k][d] = GRAV ? A.grav[(long)d * A.ncell + c[k]] : 0.0;
    }
#pragma unroll
    for (int k = 0; k < 12; k++) ctoprim_cell<NV, GRAV>(u[k], g[k], A.dt * 0.5, A.P, q[k]);
  }
  const double dtdx = A.dt / A.dx;
  double qL[NV], qR[NV];
#pragma unroll
  for (int w = 0; w < 2; w++) {
    double dq[3][NV], qm[3][NV], qp[3][NV];
#pragma unroll
    for (int n = 0; n < NV; n++) {
      dq[DIR][n] = slope1<ST>(q[w][n], q[1 + w][n], q[2 + w][n], A.P);
      dq[T0][n] = slope1<ST>(q[4 + 4 * w][n], q[1 + w][n], q[5 + 4 * w][n], A.P);
      dq[T1][n] = slope1<ST>(q[6 + 4 * w][n], q[1 + w][n], q[7 + 4 * w][n], A.P);
    }
    if (SCHEME == 0) trace3d_cell<NV>(q[1 + w], dq, dtdx, dtdx, dtdx, A.P, qm, qp);
    else tracexyz_cell<NV>(q[1 + w], dq, ctoprim_sound(q[1 + w][0], q[1 + w][4], A.P), dtdx, dtdx, dtdx, A.P, qm, qp);
#pragma unroll
    for (int n = 0; n < NV; n++) { if (w == 0) qL[n] = qm[DIR][n]; else qR[n] = qp[DIR][n]; }
  }
  scaled_interface_flux<RS, NV, DIR, false>(qL, qR, A.P, A.dt, A.dx, A.rdx, dtdx, A.pow2 != 0, fl);   // (as the marching kernel of a level in tiles does)
}
template <int ST, int RS, int NV, bool GRAV, int SCHEME = 0>
__global__ __launch_bounds__(128) void surface_flux_kernel(SurfArgs A) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long)A.nevent * 4) return;
  // (events arrive sorted by device oct and face -- round 6, session T: 2.59 -> 2.38 ms strict on the shell level against the
  //  (face, oct) order with 64 events of one fine face per wave -- and the four fine faces of an event sit in neighbouring
  //  lanes: an oct's records and the cells its faces read are touched by one wave)
  const int e = A.qminor ? (int)(t >> 2) : (int)(t % A.nevent), q = A.qminor ? (int)(t & 3) : (int)(t / A.nevent);
  const int ev = A.events[e];
  const int io = ev / 6, f = ev % 6;
  const int dirn = f >> 1, side = f & 1;
  // the oct's position from its device index (its slab of 512 indices is one tile)
  const long r = (long)A.ig[io] - A.base;
  const int tl = A.tileid[r / TILE_OCTS], l = (int)(r % TILE_OCTS);
  int p[3] = {2 * ((tl % A.ntx) * TILE_OX + l % TILE_OX), 2 * (((tl / A.ntx) % A.nty) * TILE_OY + (l / TILE_OX) % TILE_OY),
              2 * ((tl / (A.ntx * A.nty)) * TILE_OZ + l / (TILE_OX * TILE_OY))};
  // the updated cell behind fine face q of face f (q: the two transverse coordinates, lower axis first); the ghost cell is the
  // one beyond the face; lo = the left cell of the interface
  const int t0 = dirn == 0 ? 1 : 0, t1 = dirn == 2 ? 1 : 2;
  p[dirn] += side; p[t0] += q & 1; p[t1] += q >> 1;
  // hydro/godunov_fine.f90:720-747: reset when the cell on either side is refined (a ghost cell never is)
  const bool zero = (A.stat[surf_cell(A, p[0], p[1], p[2])] & CELL_REFINED) != 0;
  int lo[3] = {p[0], p[1], p[2]};
  if (!side) lo[dirn] -= 1;
  double fl[NV];
  if (dirn == 0) surf_interface<ST, RS, NV, GRAV, SCHEME, 0>(A, lo, fl);
  else if (dirn == 1) surf_interface<ST, RS, NV, GRAV, SCHEME, 1>(A, lo, fl);
  else surf_interface<ST, RS, NV, GRAV, SCHEME, 2>(A, lo, fl);
  double *dst = A.rec + ((long)e * 4 + q) * (NV + 2);
#pragma unroll
  for (int n = 0; n < NV; n++) dst[n] = zero ? 0.0 : fl[n];
}

template <int ST, int RS, int NV, int SCHEME = 0>
static hipError_t surface2(const SurfArgs &A, bool grav, hipStream_t s) {
  const dim3 grid((unsigned)(((long)A.nevent * 4 + 127) / 128)), block(128);
  if (grav) hipLaunchKernelGGL((surface_flux_kernel<ST, RS, NV, true, SCHEME>), grid, block, 0, s, A);
  else hipLaunchKernelGGL((surface_flux_kernel<ST, RS, NV, false, SCHEME>), grid, block, 0, s, A);
  return hipGetLastError();
}
template <int ST, int RS>
static hipError_t surface1(const SurfArgs &A, int nvar, int scheme, bool grav, hipStream_t s) {
  if constexpr (ST == 4 || ST == 5 || ST == 6) {
    return hipErrorInvalidValue;
  } else {
#ifndef SWEEP_FLAGSHIP_ONLY
    if (scheme == 1) return nvar == 5 ? surface2<ST, RS, 5, 1>(A, grav, s) : hipErrorInvalidValue;
#endif
    if (scheme != 0) return hipErrorInvalidValue;
    if (nvar == 5) return surface2<ST, RS, 5>(A, grav, s);
#ifndef SWEEP_FLAGSHIP_ONLY
    if (nvar == 6) return surface2<ST, RS, 6>(A, grav, s);
    if (nvar == 7) return surface2<ST, RS, 7>(A, grav, s);
#endif
    return hipErrorInvalidValue;
  }
}
template <int ST>
hipError_t surface0(const SurfArgs &A, int rs, int nvar, int scheme, bool grav, hipStream_t s) {
  switch (rs) {
    case RIEMANN_LLF: return surface1<ST, RIEMANN_LLF>(A, nvar, scheme, grav, s);
#ifndef SWEEP_FLAGSHIP_ONLY
    case RIEMANN_HLLC: return surface1<ST, RIEMANN_HLLC>(A, nvar, scheme, grav, s);
    case RIEMANN_HLL: return surface1<ST, RIEMANN_HLL>(A, nvar, scheme, grav, s);
    case RIEMANN_ACOUSTIC: return surface1<ST, RIEMANN_ACOUSTIC>(A, nvar, scheme, grav, s);
    case RIEMANN_EXACT: return surface1<ST, RIEMANN_EXACT>(A, nvar, scheme, grav, s);
#endif
  }
  return hipErrorInvalidValue;
}
// One translation unit per slope type (ramses_amd/build.py compiles this file once without SWEEP_ST -- the dispatchers -- and once
// per slope type with -DSWEEP_ST=<type>, 3 standing for 3, 4, 5 and 6, for each arithmetic: the instantiations of the option
// matrix build side by side instead of in two nine-minute compiles; the variant builds of scripts/build_variant.sh --
// SWEEP_FLAGSHIP_ONLY -- keep one unit)
#if defined(SWEEP_ST)
template hipError_t surface0<SWEEP_ST>(const SurfArgs &, int, int, int, bool, hipStream_t);
#elif !defined(SWEEP_FLAGSHIP_ONLY)
#define SWEEP_EXTERN_ST(K) extern template hipError_t surface0<K>(const SurfArgs &, int, int, int, bool, hipStream_t);
SWEEP_EXTERN_ST(0) SWEEP_EXTERN_ST(1) SWEEP_EXTERN_ST(2) SWEEP_EXTERN_ST(3) SWEEP_EXTERN_ST(7) SWEEP_EXTERN_ST(8)
#undef SWEEP_EXTERN_ST
#endif
#ifndef SWEEP_ST
hipError_t launch_surface_flux(const SurfArgs &A, int slope_type, int riemann, int nvar, int scheme, bool grav, hipStream_t s) {
  if (A.nevent <= 0) return hipSuccess;
  switch (slope_type) {
    case 1: return surface0<1>(A, riemann, nvar, scheme, grav, s);
#ifndef SWEEP_FLAGSHIP_ONLY
    case 0: return surface0<0>(A, riemann, nvar, scheme, grav, s);
    case 2: return surface0<2>(A, riemann, nvar, scheme, grav, s);
    case 3: return surface0<3>(A, riemann, nvar, scheme, grav, s);
    case 7: return surface0<7>(A, riemann, nvar, scheme, grav, s);
    case 8: return surface0<8>(A, riemann, nvar, scheme, grav, s);
#endif
  }
  return hipErrorInvalidValue;
}
#endif   // SWEEP_ST

// ---------------------------------------------------------------------------
// host-side dispatch
// ---------------------------------------------------------------------------
template <int ST, int RS, int BY, bool GRAV, int SCHEME, int NV, bool MASK = false>
static hipError_t launch3(const SweepArgs &A, hipStream_t s) {
  const size_t lds = Lds<ST, BY, NV, MASK, GRAV>::bytes;
  dim3 block(BX, BY);
  dim3 grid(A.nblocks);
  auto k = godunov_sweep_kernel<ST, RS, BY, GRAV, SCHEME, NV, MASK>;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k, grid, block, lds, s, A);
  return hipGetLastError();
}

template <int ST, int RS, int BY, int SCHEME, int NV>
static hipError_t launch2(const SweepArgs &A, bool grav, hipStream_t s) {
  return grav ? launch3<ST, RS, BY, true, SCHEME, NV>(A, s) : launch3<ST, RS, BY, false, SCHEME, NV>(A, s);
}

template <int ST, int RS>
static hipError_t launch1(SweepArgs &A, int by, int scheme, int nvar, bool grav, hipStream_t s) {
  // 8-row tiles (2 waves/SIMD, 256 VGPRs, smaller LDS planes) for the
  // register/LDS-hungry variants: the Newton solver, the 27-point slope, the
  // PLMDE tracing and runs with passive scalars
  const bool heavy = (RS == RIEMANN_EXACT) || (ST == 3) || (ST == 4) || (ST == 5) || (ST == 6) || (scheme != 0) || (nvar != 5);
  if (by == 0 || heavy) by = heavy ? 8 : 12;
  // tiles of the whole brick, then the boxes this launch covers (A.region)
  const int NTX = (A.nx + (BX - 4) - 1) / (BX - 4);
  const int NTY = (A.ny + (by - 4) - 1) / (by - 4);
  // Boundary shell / interior split used to overlap the halo exchange with the
  // interior sweep: shell = the tiles and planes that produce the cells within
  // 2 of a face (what the neighbours receive); zb planes at each z end.  The six
  // shell boxes go into ONE launch, cut into short z-chunks so that the thin
  // slabs still fill the chip.
  const int zb = 2;
  const bool splittable = NTX >= 3 && NTY >= 3 && A.nz >= 4 * zb;
  A.nbox = 0;
  int nblocks = 0;
  auto add_box = [&](int tx0, int tx1, int ty0, int ty1, int zlo, int zhi, int zchunk) {
    if (tx1 <= tx0 || ty1 <= ty0 || zhi <= zlo) return;
    SweepBox &B = A.box[A.nbox++];
    B.tx0 = tx0; B.ntx = tx1 - tx0; B.ty0 = ty0; B.nty = ty1 - ty0;
    B.zlo = zlo; B.zhi = zhi; B.zchunk = zchunk < (zhi - zlo) ? zchunk : (zhi - zlo);
    B.first = nblocks;
    nblocks += B.ntx * B.nty * ((zhi - zlo + B.zchunk - 1) / B.zchunk);
  };
  const int zc = A.zchunk;
  if (A.region == SWEEP_ALL || (A.region == SWEEP_SHELL && !splittable)) {
    add_box(0, NTX, 0, NTY, 0, A.nz, zc);
  } else if (A.region == SWEEP_INTERIOR) {
    if (splittable) add_box(1, NTX - 1, 1, NTY - 1, zb, A.nz - zb, zc);
  } else if (A.region == SWEEP_SHELL) {
    const int zs = 32;
    // (long blocks first, the 2-plane slabs fill the gaps at the end)
    add_box(0, 1, 1, NTY - 1, zb, A.nz - zb, zs);                       // x low tile column
    add_box(NTX - 1, NTX, 1, NTY - 1, zb, A.nz - zb, zs);               // x high tile column
    add_box(0, NTX, 0, 1, zb, A.nz - zb, zs);                           // y low tile row
    add_box(0, NTX, NTY - 1, NTY, zb, A.nz - zb, zs);                   // y high tile row
    add_box(0, NTX, 0, NTY, 0, zb, zs);                                 // z low slab
    add_box(0, NTX, 0, NTY, A.nz - zb, A.nz, zs);                       // z high slab
  } else {
    return hipErrorInvalidValue;
  }
  if (nblocks == 0) return hipSuccess;
  A.nblocks = nblocks;
  if (A.stat) {
    // a level of a resident AMR run in tiles: the 12-row muscl kernels on the periodic box of the level, one workgroup per
    // work item (anything else: the caller keeps the tree-walking sweep)
    if constexpr (ST != 4 && ST != 5 && ST != 6) {
      if (!A.dir || !A.work || A.ng != 0 || (scheme != 0 && scheme != 1) || A.nwork <= 0) return hipErrorInvalidValue;
      A.nblocks = A.nwork;
      A.nbox = 1;          // (the box decode runs, its result is replaced by the work item)
      // the plan's work items were cut for tile_sweep_rows(...) interior rows: 8 (12-row workgroups), or 4 for the variants that
      // need 256 registers -- the Newton solver, the 27-point slope, the PLMDE tracing, runs with passive scalars (round 6)
      if (by == 8) {
        if (scheme == 1) {
          if (nvar != 5) return hipErrorInvalidValue;
          return grav ? launch3<ST, RS, 8, true, 1, 5, true>(A, s) : launch3<ST, RS, 8, false, 1, 5, true>(A, s);
        }
        if (nvar == 5) {
          if constexpr (RS == RIEMANN_EXACT || ST == 3) return grav ? launch3<ST, RS, 8, true, 0, 5, true>(A, s) : launch3<ST, RS, 8, false, 0, 5, true>(A, s);
          else return hipErrorInvalidValue;
        }
        if (nvar == 6) return grav ? launch3<ST, RS, 8, true, 0, 6, true>(A, s) : launch3<ST, RS, 8, false, 0, 6, true>(A, s);
        if (nvar == 7) return grav ? launch3<ST, RS, 8, true, 0, 7, true>(A, s) : launch3<ST, RS, 8, false, 0, 7, true>(A, s);
        return hipErrorInvalidValue;
      }
      if constexpr (RS != RIEMANN_EXACT && ST != 3) {
        if (by == TILE_SWEEP_BY && nvar == 5 && scheme == 0)
          return grav ? launch3<ST, RS, TILE_SWEEP_BY, true, 0, 5, true>(A, s) : launch3<ST, RS, TILE_SWEEP_BY, false, 0, 5, true>(A, s);
      }
      return hipErrorInvalidValue;
    } else {
      return hipErrorInvalidValue;
    }
  }
  if constexpr (ST == 4 || ST == 5 || ST == 6) {
    // NDIM=1 slope types: the plain configuration only (the reference's 1-D tests: NVAR=3 embedded as 5, muscl, no gravity)
    if (nvar != 5 || scheme != 0 || grav) return hipErrorInvalidValue;
    return launch3<ST, RS, 8, false, 0, 5>(A, s);
  } else {
    if (nvar == 6) return scheme == 0 ? launch2<ST, RS, 8, 0, 6>(A, grav, s) : hipErrorInvalidValue;
    if (nvar == 7) return scheme == 0 ? launch2<ST, RS, 8, 0, 7>(A, grav, s) : hipErrorInvalidValue;
    if (nvar != 5) return hipErrorInvalidValue;
    if (scheme == 1) return launch2<ST, RS, 8, 1, 5>(A, grav, s);
    if (by == 8) return launch2<ST, RS, 8, 0, 5>(A, grav, s);
    if constexpr (ST != 3 && RS != RIEMANN_EXACT) {
      if (by == 12) return launch2<ST, RS, 12, 0, 5>(A, grav, s);
    }
    return hipErrorInvalidValue;
  }
}

template <int ST>
hipError_t launch0(SweepArgs &A, int rs, int by, int scheme, int nvar, bool grav, hipStream_t s) {
  switch (rs) {
    case RIEMANN_LLF: return launch1<ST, RIEMANN_LLF>(A, by, scheme, nvar, grav, s);
#ifndef SWEEP_FLAGSHIP_ONLY   // (scripts/sweep_regs.sh, build_ab.py: the LLF + minmod instantiations only, a one-minute compile)
    case RIEMANN_HLLC: return launch1<ST, RIEMANN_HLLC>(A, by, scheme, nvar, grav, s);
    case RIEMANN_HLL: return launch1<ST, RIEMANN_HLL>(A, by, scheme, nvar, grav, s);
    case RIEMANN_ACOUSTIC: return launch1<ST, RIEMANN_ACOUSTIC>(A, by, scheme, nvar, grav, s);
    case RIEMANN_EXACT: return launch1<ST, RIEMANN_EXACT>(A, by, scheme, nvar, grav, s);
#endif
  }
  return hipErrorInvalidValue;
}

#if defined(SWEEP_ST)
template hipError_t launch0<SWEEP_ST>(SweepArgs &, int, int, int, int, bool, hipStream_t);
#if SWEEP_ST == 3
template hipError_t launch0<4>(SweepArgs &, int, int, int, int, bool, hipStream_t);
template hipError_t launch0<5>(SweepArgs &, int, int, int, int, bool, hipStream_t);
template hipError_t launch0<6>(SweepArgs &, int, int, int, int, bool, hipStream_t);
#endif
#elif !defined(SWEEP_FLAGSHIP_ONLY)
#define SWEEP_EXTERN_ST(K) extern template hipError_t launch0<K>(SweepArgs &, int, int, int, int, bool, hipStream_t);
SWEEP_EXTERN_ST(0) SWEEP_EXTERN_ST(1) SWEEP_EXTERN_ST(2) SWEEP_EXTERN_ST(3) SWEEP_EXTERN_ST(4) SWEEP_EXTERN_ST(5) SWEEP_EXTERN_ST(6)
SWEEP_EXTERN_ST(7) SWEEP_EXTERN_ST(8)
#undef SWEEP_EXTERN_ST
#endif

#ifndef SWEEP_ST
// interior rows of a work item of the sweep of a level in tiles (the plan of csrc/capi_amr.hip cuts the level accordingly)
int tile_sweep_rows(int riemann, int nvar, int slope_type, int scheme) {
  return ((riemann == RIEMANN_EXACT || nvar != 5 || slope_type == 3 || scheme != 0) ? 8 : TILE_SWEEP_BY) - 4;
}

hipError_t launch_godunov_sweep(SweepArgs &A, int slope_type, int riemann, int by, int scheme, int nvar,
                                bool grav, hipStream_t s) {
  // lanes address a plane with a 32-bit byte offset
  if ((unsigned long)A.pitch_z * 8ul >= (1ul << 31) || (unsigned long)A.pitch_var * 8ul >= (1ul << 32))
    return hipErrorInvalidValue;
  switch (slope_type) {
    case 1: return launch0<1>(A, riemann, by, scheme, nvar, grav, s);
#ifndef SWEEP_FLAGSHIP_ONLY
    case 0: return launch0<0>(A, riemann, by, scheme, nvar, grav, s);
    case 2: return launch0<2>(A, riemann, by, scheme, nvar, grav, s);
    case 3: return launch0<3>(A, riemann, by, scheme, nvar, grav, s);
    case 4: return launch0<4>(A, riemann, by, scheme, nvar, grav, s);
    case 5: return launch0<5>(A, riemann, by, scheme, nvar, grav, s);
    case 6: return launch0<6>(A, riemann, by, scheme, nvar, grav, s);
    case 7: return launch0<7>(A, riemann, by, scheme, nvar, grav, s);
    case 8: return launch0<8>(A, riemann, by, scheme, nvar, grav, s);
#endif
  }
  return hipErrorInvalidValue;
}

#endif   // SWEEP_ST

}  // namespace SWEEP_NS
}  // namespace ramses_amd

// (the units of one slope type are not warmed up: a run uses one of the twelve, which loads with its first sweep)
#include "warm.hpp"
#if defined(SWEEP_ST)
#elif RAMSES_AMD_FAST
RAMSES_AMD_TU_WARM(hydro_sweep_fast)
#else
RAMSES_AMD_TU_WARM(hydro_sweep_strict)
#endif
