// halo_plan.hpp -- host-side plan of the GPU-resident virtual-boundary exchange of one
// fully refined level under MPI (no device code here: plain C++, also exercised on CPU).
//
// Reference: the communicators of amr/amr_commons.f90:108-119,170-179 --
// emission(icpu,l)%igrid / reception(icpu,l)%igrid built by build_comm
// (amr/virtual_boundaries.f90:1286-1648) -- and the pack/unpack loops of
// make_virtual_fine_dp (:454-464, :492-506) / make_virtual_reverse_dp (:693-983).
//
// One rank's share of a uniform level is a box of octs (the Hilbert split of a uniform grid over 2^k
// ranks).  On the device it is a brick with a ghost layer of one oct (two cells):
//   cell (i,j,k) of the box, i in [-2, 2*odim_x+2)  ->  (i+2) + pitch_y*(j+2) + pitch_z*(k+2)
// The plan turns the reference's oct lists into brick offsets:
//   act_org[g]            origin of active oct g (interior)
//   em_org[m]             origin of the m-th oct of the concatenated emission lists (interior)
//   rc_src[r], rc_org[r]  ghost position rc_org[r] receives oct rc_src[r] of the concatenated
//                         reception lists (an oct can serve two sides of a direction that only
//                         two ranks share; octs beyond the one-oct shell are not needed and skipped)
// A direction the box spans completely (odim == number of octs of the level) has no peer: its ghost
// layer is a periodic copy of the rank's own interior (self_axes bit d), applied after the unpack over the
// full allocated extent of the other directions.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace ramses_amd {

struct HaloPlan {
  int level = 0, no = 0;            // octs per direction of the whole level (2^(level-1))
  int olo[3] = {0, 0, 0};           // box origin, in octs
  int odim[3] = {0, 0, 0};          // box extent, in octs
  int self_axes = 0;                // bit d: direction d is spanned completely (periodic self-fill)
  int nx = 0, ny = 0, nz = 0;       // interior cells of the brick (2*odim)
  int64_t pitch_y = 0, pitch_z = 0, pitch_var = 0;
  int ncpu = 0;
  std::vector<int64_t> act_org;     // [ngrid]
  std::vector<int> em_first;        // [ncpu+1] prefix sums of the emission list lengths
  std::vector<int> rc_first;        // [ncpu+1]
  std::vector<int64_t> em_org;      // [em_first[ncpu]]
  std::vector<int> rc_src;          // [nrc_used]
  std::vector<int64_t> rc_org;      // [nrc_used]
  std::string error;
};

// oct coordinate (per direction) of oct slot `ig` (1-based) of a level with `no` octs per direction
inline int oct_coord(const double *xg, int64_t ngridmax, int ig, int d, int no) {
  const double x = xg[(int64_t)d * ngridmax + (ig - 1)] * no;   // centre = (o + 1/2)/no
  return (int)x;                                                // floor: x > 0
}

// Builds the plan.  Returns false (plan.error set) when the active octs do not fill a box, a
// list entry is inconsistent, or the one-oct shell is not completely covered by reception octs.
inline bool build_halo_plan(int level, int ngrid, const int *igrid, const double *xg, int64_t ngridmax, int ncpu,
                            const int *em_ngrid, const int *em_igrid, const int *rc_ngrid, const int *rc_igrid,
                            HaloPlan &P) {
  P = HaloPlan();
  P.level = level;
  P.ncpu = ncpu;
  if (level < 2 || level > 30 || ngrid < 1) { P.error = "bad level / empty level"; return false; }
  const int no = 1 << (level - 1);
  P.no = no;
  int lo[3] = {no, no, no}, hi[3] = {-1, -1, -1};
  for (int g = 0; g < ngrid; g++)
    for (int d = 0; d < 3; d++) {
      const int o = oct_coord(xg, ngridmax, igrid[g], d, no);
      if (o < 0 || o >= no) { P.error = "oct centre outside the box (nx=ny=nz=1 expected)"; return false; }
      if (o < lo[d]) lo[d] = o;
      if (o > hi[d]) hi[d] = o;
    }
  int64_t vol = 1;
  for (int d = 0; d < 3; d++) { P.olo[d] = lo[d]; P.odim[d] = hi[d] - lo[d] + 1; vol *= P.odim[d]; }
  if (vol != ngrid) {
    P.error = "the rank's octs of level " + std::to_string(level) + " do not fill a box (" + std::to_string(ngrid) +
              " octs in a bounding box of " + std::to_string((long long)vol) + ")";
    return false;
  }
  for (int d = 0; d < 3; d++)
    if (P.odim[d] == no) P.self_axes |= 1 << d;
  P.nx = 2 * P.odim[0]; P.ny = 2 * P.odim[1]; P.nz = 2 * P.odim[2];
  P.pitch_y = P.nx + 4;
  P.pitch_z = P.pitch_y * (P.ny + 4);
  P.pitch_var = P.pitch_z * (P.nz + 4);
  auto org_of = [&](const int r[3]) -> int64_t {
    return (int64_t)(2 * r[0] + 2) + P.pitch_y * (2 * r[1] + 2) + P.pitch_z * (2 * r[2] + 2);
  };
  // active octs: interior positions, each exactly once
  P.act_org.resize(ngrid);
  std::vector<char> seen((size_t)vol, 0);
  for (int g = 0; g < ngrid; g++) {
    int r[3];
    for (int d = 0; d < 3; d++) r[d] = oct_coord(xg, ngridmax, igrid[g], d, no) - lo[d];
    const size_t s = (size_t)r[0] + (size_t)P.odim[0] * (r[1] + (size_t)P.odim[1] * r[2]);
    if (seen[s]) { P.error = "two active octs at one position"; return false; }
    seen[s] = 1;
    P.act_org[g] = org_of(r);
  }
  // emission lists: own octs
  P.em_first.assign(ncpu + 1, 0);
  P.rc_first.assign(ncpu + 1, 0);
  for (int c = 0; c < ncpu; c++) {
    if (em_ngrid[c] < 0 || rc_ngrid[c] < 0) { P.error = "negative list length"; return false; }
    P.em_first[c + 1] = P.em_first[c] + em_ngrid[c];
    P.rc_first[c + 1] = P.rc_first[c] + rc_ngrid[c];
  }
  P.em_org.resize(P.em_first[ncpu]);
  for (int m = 0; m < P.em_first[ncpu]; m++) {
    int r[3];
    for (int d = 0; d < 3; d++) {
      r[d] = oct_coord(xg, ngridmax, em_igrid[m], d, no) - lo[d];
      if (r[d] < 0 || r[d] >= P.odim[d]) { P.error = "an emission oct lies outside the rank's box"; return false; }
    }
    P.em_org[m] = org_of(r);
  }
  // reception lists: the shell of one oct around the box in the directions shared with peers
  const int ext[3] = {P.odim[0] + ((P.self_axes & 1) ? 0 : 2), P.odim[1] + ((P.self_axes & 2) ? 0 : 2),
                      P.odim[2] + ((P.self_axes & 4) ? 0 : 2)};
  const int off[3] = {(P.self_axes & 1) ? 0 : 1, (P.self_axes & 2) ? 0 : 1, (P.self_axes & 4) ? 0 : 1};
  std::vector<char> cover((size_t)ext[0] * ext[1] * ext[2], 0);
  for (int m = 0; m < P.rc_first[ncpu]; m++) {
    int cand[3][2], nc[3];
    for (int d = 0; d < 3; d++) {
      const int o = oct_coord(xg, ngridmax, rc_igrid[m], d, no) - lo[d];
      nc[d] = 0;
      const int lo_ok = (P.self_axes >> d & 1) ? 0 : -1, hi_ok = (P.self_axes >> d & 1) ? P.odim[d] - 1 : P.odim[d];
      for (int w = -1; w <= 1; w++) {
        const int r = o + w * no;
        if (r >= lo_ok && r <= hi_ok && nc[d] < 2) cand[d][nc[d]++] = r;
      }
    }
    for (int a = 0; a < nc[0]; a++)
      for (int b = 0; b < nc[1]; b++)
        for (int c = 0; c < nc[2]; c++) {
          const int r[3] = {cand[0][a], cand[1][b], cand[2][c]};
          const bool interior = r[0] >= 0 && r[0] < P.odim[0] && r[1] >= 0 && r[1] < P.odim[1] && r[2] >= 0 && r[2] < P.odim[2];
          if (interior) { P.error = "a reception oct lies inside the rank's own box"; return false; }
          const size_t s = (size_t)(r[0] + off[0]) + (size_t)ext[0] * ((r[1] + off[1]) + (size_t)ext[1] * (r[2] + off[2]));
          if (cover[s]) { P.error = "two reception octs for one ghost position"; return false; }
          cover[s] = 1;
          P.rc_src.push_back(m);
          P.rc_org.push_back(org_of(r));
        }
  }
  const size_t need = (size_t)ext[0] * ext[1] * ext[2] - (size_t)vol;
  if (P.rc_src.size() != need) {
    P.error = "the reception lists cover " + std::to_string(P.rc_src.size()) + " of the " + std::to_string(need) +
              " ghost octs around the rank's box";
    return false;
  }
  return true;
}

}  // namespace ramses_amd
