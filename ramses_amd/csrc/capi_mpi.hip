// capi_mpi.hip -- the GPU-resident virtual-boundary exchange behind the C ABI
// (include/ramses_amd.h, "MPI: one rank per GPU"): what the Fortran shim
// ramses_amd/patch/virtual_boundaries.f90 and the Python host mirror bind.
//
//  * ramses_amd_rccl_*: neighbour send/recv over xGMI.  librccl.so is dlopen'ed on first use (single-rank
//    runs never load it); the communicator is bootstrapped from the caller's own launcher (the Fortran
//    shim broadcasts the 128-byte unique id with MPI_BCAST, the Python mirror with torch.distributed).
//    One grouped ncclSend/ncclRecv per exchange, one message per peer, all nvar fields fused
//    (reference: nvar rounds of MPI_ISEND/IRECV per exchange, amr/virtual_boundaries.f90:373-528).
//  * ramses_amd_mpires_*: one rank's share of a fully refined periodic level as a device-resident brick
//    with a one-oct ghost layer, driven by the reference's own communicators (emission/reception oct
//    lists of build_comm): courant_fine / set_unew / godunov_fine / set_uold / make_virtual_fine_dp /
//    make_virtual_reverse_dp of amr_step without the state ever crossing PCIe.  Transport: RCCL, or
//    (when RCCL cannot be brought up, e.g. several ranks sharing one GPU) the shim's own MPI on pinned
//    host buffers -- the exchange then says so.
#include <dlfcn.h>
#include <unistd.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/ramses_amd.h"
#include "halo_plan.hpp"
#include "misc_args.hpp"
#include "pack_args.hpp"

using namespace ramses_amd;

extern "C" int ramses_amd_set_error(int code, const char *msg);   // capi.hip: fills ramses_amd_last_error()

static int failf(int code, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  return ramses_amd_set_error(code, buf);
}
#define HCHK(call, what) do { hipError_t e_ = (call); if (e_ != hipSuccess) return failf(RAMSES_AMD_EHIP, "%s: %s", what, hipGetErrorString(e_)); } while (0)

// ---------------------------------------------------------------------------
// RCCL, loaded on demand
// ---------------------------------------------------------------------------
namespace {
struct Rccl {
  void *h = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
  ncclComm_t comm = nullptr;
  int nranks = 0, rank = -1;
};
Rccl g_rccl;

int rccl_load() {
  Rccl &R = g_rccl;
  if (R.h) return 0;
  const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
  for (const char *n : names) {
    R.h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (R.h) break;
  }
  if (!R.h) return failf(RAMSES_AMD_EUNSUPPORTED, "librccl.so cannot be loaded: %s", dlerror());
#define SYM(field, name) \
  *reinterpret_cast<void **>(&R.field) = dlsym(R.h, name); \
  if (!R.field) return failf(RAMSES_AMD_EUNSUPPORTED, "librccl.so lacks %s", name)
  SYM(GetUniqueId, "ncclGetUniqueId");
  SYM(CommInitRank, "ncclCommInitRank");
  SYM(CommDestroy, "ncclCommDestroy");
  SYM(Send, "ncclSend");
  SYM(Recv, "ncclRecv");
  SYM(GroupStart, "ncclGroupStart");
  SYM(GroupEnd, "ncclGroupEnd");
  SYM(AllReduce, "ncclAllReduce");
  SYM(AllGather, "ncclAllGather");
  SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
  return 0;
}
#define NCHK(call, what) do { ncclResult_t r_ = (call); if (r_ != ncclSuccess) return failf(RAMSES_AMD_EHIP, "%s: %s", what, g_rccl.GetErrorString(r_)); } while (0)
// inside ncclGroupStart ... ncclGroupEnd: close the group before reporting, so that the next exchange does not nest into it
#define NCHK_GROUP(call, what) do { ncclResult_t r_ = (call); if (r_ != ncclSuccess) { g_rccl.GroupEnd(); return failf(RAMSES_AMD_EHIP, "%s: %s", what, g_rccl.GetErrorString(r_)); } } while (0)
}  // namespace

extern "C" {

// A number that is equal for two processes exactly when they drive the same GPU of the same host
// (hash of the host name and the PCI bus id of the current device): RCCL refuses communicators with two
// ranks on one device, so the shim checks before it tries.
int ramses_amd_device_uid(int64_t *uid) {
  if (!uid) return failf(RAMSES_AMD_EINVAL, "NULL argument");
  int dev = 0;
  HCHK(hipGetDevice(&dev), "hipGetDevice");
  char bus[64] = "";
  HCHK(hipDeviceGetPCIBusId(bus, sizeof(bus), dev), "hipDeviceGetPCIBusId");
  char host[256] = "";
  gethostname(host, sizeof(host) - 1);
  uint64_t h = 1469598103934665603ull;   // FNV-1a
  for (const char *c = host; *c; c++) { h ^= (unsigned char)*c; h *= 1099511628211ull; }
  h ^= 0xff; h *= 1099511628211ull;
  for (const char *c = bus; *c; c++) { h ^= (unsigned char)*c; h *= 1099511628211ull; }
  *uid = (int64_t)(h >> 1);
  return 0;
}

// Local half of bringing RCCL up: dlopen + symbol lookup, no communication.  The launcher reduces the result over
// the ranks and enters the collective ncclCommInitRank only when every rank passed (a rank that cannot load the
// library would otherwise leave the others blocked in it).
int ramses_amd_rccl_probe(void) {
  if (int rc = rccl_load()) return rc;
  int dev = 0;
  HCHK(hipGetDevice(&dev), "hipGetDevice");
  return 0;
}

int ramses_amd_rccl_unique_id(char *id128) {
  if (!id128) return failf(RAMSES_AMD_EINVAL, "NULL argument");
  if (int rc = rccl_load()) return rc;
  ncclUniqueId id;
  NCHK(g_rccl.GetUniqueId(&id), "ncclGetUniqueId");
  static_assert(sizeof(id) == RAMSES_AMD_RCCL_ID_BYTES, "ncclUniqueId size");
  std::memcpy(id128, &id, sizeof(id));
  return 0;
}

int ramses_amd_rccl_init(const char *id128, int nranks, int rank) {
  if (!id128 || nranks < 1 || rank < 0 || rank >= nranks) return failf(RAMSES_AMD_EINVAL, "bad argument");
  if (int rc = rccl_load()) return rc;
  Rccl &R = g_rccl;
  if (R.comm) { R.CommDestroy(R.comm); R.comm = nullptr; }
  ncclUniqueId id;
  std::memcpy(&id, id128, sizeof(id));
  NCHK(R.CommInitRank(&R.comm, nranks, id, rank), "ncclCommInitRank");
  R.nranks = nranks; R.rank = rank;
  return 0;
}

int ramses_amd_rccl_ready(void) { return g_rccl.comm != nullptr; }

int ramses_amd_rccl_finalize(void) {
  Rccl &R = g_rccl;
  if (R.comm) { R.CommDestroy(R.comm); R.comm = nullptr; }
  R.nranks = 0; R.rank = -1;
  return 0;
}

// One grouped neighbour exchange: message i goes to / comes from rank peer[i]; offsets and counts in
// doubles into the device buffers.  Asynchronous on stream.
int ramses_amd_rccl_exchange(int npeer, const int *peer, const double *d_send, const int64_t *send_off,
                             const int64_t *send_cnt, double *d_recv, const int64_t *recv_off,
                             const int64_t *recv_cnt, void *stream) {
  Rccl &R = g_rccl;
  if (!R.comm) return failf(RAMSES_AMD_EINVAL, "RCCL communicator not initialised (ramses_amd_rccl_init)");
  if (npeer < 0 || (npeer > 0 && (!peer || !send_off || !send_cnt || !recv_off || !recv_cnt)))
    return failf(RAMSES_AMD_EINVAL, "bad argument");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  NCHK(R.GroupStart(), "ncclGroupStart");
  for (int i = 0; i < npeer; i++) {
    // peer[i] == own rank is legal: RCCL matches a send to self with the receive from self of the same group
    if (peer[i] < 0 || peer[i] >= R.nranks) { R.GroupEnd(); return failf(RAMSES_AMD_EINVAL, "bad peer rank %d", peer[i]); }
    if (peer[i] == R.rank && send_cnt[i] != recv_cnt[i]) { R.GroupEnd(); return failf(RAMSES_AMD_EINVAL, "message to self: %lld doubles sent, %lld expected", (long long)send_cnt[i], (long long)recv_cnt[i]); }
    if (recv_cnt[i] > 0) NCHK_GROUP(R.Recv(d_recv + recv_off[i], (size_t)recv_cnt[i], ncclDouble, peer[i], R.comm, s), "ncclRecv");
    if (send_cnt[i] > 0) NCHK_GROUP(R.Send(d_send + send_off[i], (size_t)send_cnt[i], ncclDouble, peer[i], R.comm, s), "ncclSend");
  }
  NCHK(R.GroupEnd(), "ncclGroupEnd");
  return 0;
}

// The same with one pointer per message (the Python mirror's tensors): send i to send_peer[i], receive i from
// recv_peer[i]; per peer, messages match in posting order.
int ramses_amd_rccl_sendrecv(int nsend, const double *const *send_ptr, const int64_t *send_cnt, const int *send_peer,
                             int nrecv, double *const *recv_ptr, const int64_t *recv_cnt, const int *recv_peer, void *stream) {
  Rccl &R = g_rccl;
  if (!R.comm) return failf(RAMSES_AMD_EINVAL, "RCCL communicator not initialised (ramses_amd_rccl_init)");
  if (nsend < 0 || nrecv < 0 || (nsend > 0 && (!send_ptr || !send_cnt || !send_peer)) || (nrecv > 0 && (!recv_ptr || !recv_cnt || !recv_peer)))
    return failf(RAMSES_AMD_EINVAL, "bad argument");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  NCHK(R.GroupStart(), "ncclGroupStart");
  for (int i = 0; i < nrecv; i++) {
    if (recv_peer[i] < 0 || recv_peer[i] >= R.nranks) { R.GroupEnd(); return failf(RAMSES_AMD_EINVAL, "bad peer rank %d", recv_peer[i]); }
    NCHK_GROUP(R.Recv(recv_ptr[i], (size_t)recv_cnt[i], ncclDouble, recv_peer[i], R.comm, s), "ncclRecv");
  }
  for (int i = 0; i < nsend; i++) {
    if (send_peer[i] < 0 || send_peer[i] >= R.nranks) { R.GroupEnd(); return failf(RAMSES_AMD_EINVAL, "bad peer rank %d", send_peer[i]); }
    NCHK_GROUP(R.Send(send_ptr[i], (size_t)send_cnt[i], ncclDouble, send_peer[i], R.comm, s), "ncclSend");
  }
  NCHK(R.GroupEnd(), "ncclGroupEnd");
  return 0;
}

// in-place all-reduce of n doubles on the device: op 0 sum, 1 min, 2 max (the scalar reductions of
// courant_fine hydro/courant_fine.f90:133-140, of the multigrid norms and of the CG dot products)
int ramses_amd_rccl_allreduce(double *d_buf, int n, int op, void *stream) {
  Rccl &R = g_rccl;
  if (!R.comm) return failf(RAMSES_AMD_EINVAL, "RCCL communicator not initialised (ramses_amd_rccl_init)");
  if (!d_buf || n < 1 || op < 0 || op > 2) return failf(RAMSES_AMD_EINVAL, "bad argument");
  const ncclRedOp_t o = op == 0 ? ncclSum : (op == 1 ? ncclMin : ncclMax);
  NCHK(R.AllReduce(d_buf, d_buf, (size_t)n, ncclDouble, o, R.comm, reinterpret_cast<hipStream_t>(stream)), "ncclAllReduce");
  return 0;
}

// d_recv[r*count .. (r+1)*count) = d_send of rank r (the replicated coarse levels of the distributed multigrid)
int ramses_amd_rccl_allgather(const double *d_send, int64_t count, double *d_recv, void *stream) {
  Rccl &R = g_rccl;
  if (!R.comm) return failf(RAMSES_AMD_EINVAL, "RCCL communicator not initialised (ramses_amd_rccl_init)");
  if (!d_send || !d_recv || count < 1) return failf(RAMSES_AMD_EINVAL, "bad argument");
  NCHK(R.AllGather(d_send, d_recv, (size_t)count, ncclDouble, R.comm, reinterpret_cast<hipStream_t>(stream)), "ncclAllGather");
  return 0;
}

}  // extern "C"

// ---------------------------------------------------------------------------
// oct-list kernels: messages are indexed by the position m in the concatenated list:
//   buf[(m*nvar + v)*8 + ind]  <->  brick[org[m] + (ind&1) + pitch_y*((ind>>1)&1) + pitch_z*(ind>>2) + v*pitch_var]
// (reference layout per message: u(i+(ind-1)*ngrid,1) per variable, virtual_boundaries.f90:454-464;
// here all variables travel together and both ends use the same indexing)
// ---------------------------------------------------------------------------
namespace {
struct OctListArgs {
  double *brick;
  double *buf;
  const int64_t *org;     // [n]
  const int *src;         // [n] message oct of list entry (nullptr: identity)
  int n, nvar;
  long pitch_y, pitch_z, pitch_var;
};
enum { OL_PACK = 0, OL_UNPACK = 1, OL_ACCUM = 2, OL_ADDZERO = 3 };

template <int MODE>
__global__ __launch_bounds__(256) void oct_list_kernel(OctListArgs A) {
  const long total = (long)A.n * 8;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    const int e = (int)(t >> 3), ind = (int)(t & 7);
    const long m = A.src ? A.src[e] : e;
    const long b = A.org[e] + (ind & 1) + A.pitch_y * ((ind >> 1) & 1) + A.pitch_z * (ind >> 2);
    for (int v = 0; v < A.nvar; v++) {
      double *cell = A.brick + b + (long)v * A.pitch_var;
      double *msg = A.buf + (m * A.nvar + v) * 8 + ind;
      if (MODE == OL_PACK) *msg = *cell;
      else if (MODE == OL_UNPACK) *cell = *msg;
      else if (MODE == OL_ACCUM) *cell = *cell + *msg;
      else *cell = *cell + 0.0;
    }
  }
}

hipError_t launch_oct_list(const OctListArgs &A, int mode, hipStream_t s) {
  if (A.n <= 0) return hipSuccess;
  long grid = ((long)A.n * 8 + 255) / 256;
  if (grid > 8192) grid = 8192;
  switch (mode) {
    case OL_PACK: hipLaunchKernelGGL(oct_list_kernel<OL_PACK>, dim3((int)grid), dim3(256), 0, s, A); break;
    case OL_UNPACK: hipLaunchKernelGGL(oct_list_kernel<OL_UNPACK>, dim3((int)grid), dim3(256), 0, s, A); break;
    case OL_ACCUM: hipLaunchKernelGGL(oct_list_kernel<OL_ACCUM>, dim3((int)grid), dim3(256), 0, s, A); break;
    default: hipLaunchKernelGGL(oct_list_kernel<OL_ADDZERO>, dim3((int)grid), dim3(256), 0, s, A); break;
  }
  return hipGetLastError();
}

struct Buf {
  void *p = nullptr;
  size_t cap = 0;
  hipError_t ensure(size_t bytes) {
    if (bytes <= cap && p) return hipSuccess;
    if (p) { hipFree(p); p = nullptr; cap = 0; }
    if (bytes == 0) bytes = 8;
    hipError_t e = hipMalloc(&p, bytes);
    if (e == hipSuccess) cap = bytes;
    return e;
  }
  template <class T> T *as() { return reinterpret_cast<T *>(p); }
};
struct PinBuf {
  void *p = nullptr;
  size_t cap = 0;
  hipError_t ensure(size_t bytes) {
    if (bytes <= cap && p) return hipSuccess;
    if (p) { hipHostFree(p); p = nullptr; cap = 0; }
    if (bytes == 0) bytes = 8;
    hipError_t e = hipHostMalloc(&p, bytes, hipHostMallocDefault);
    if (e == hipSuccess) cap = bytes;
    return e;
  }
  template <class T> T *as() { return reinterpret_cast<T *>(p); }
};

struct MpiRes {
  bool valid = false, host_stale = false, new_ready = false;
  int level = 0, ngrid = 0, nvar = 0, ncpu = 0, myid = 0;
  long ncell = 0, ncoarse = 0, ngridmax = 0;
  const double *h_uold = nullptr, *h_unew = nullptr;
  HaloPlan plan;
  ramses_amd_brick brick;
  Buf bold, bnew, vec, load_igrid, load_org, em_org, rc_src, rc_org, sendbuf, recvbuf, red;
  PinBuf h_send, h_recv;
  std::vector<int> peers;                       // ranks (0-based) with a non-empty list
  std::vector<int64_t> send_off, send_cnt, recv_off, recv_cnt;   // per entry of peers, in doubles
  std::vector<int64_t> f_send_off, f_recv_off;  // [ncpu+1] for the Fortran shim (host-staged transport)
  long nexchanges = 0;
  // overlap: the shell sweep produces every cell the peers receive; the exchange of the NEW state then runs on a
  // second stream behind the interior sweep (the reference runs them back to back, amr/amr_step.f90:388-510)
  hipStream_t s_comp = nullptr, s_comm = nullptr;
  hipEvent_t ev_shell = nullptr, ev_comm = nullptr;
  bool overlap = false;         // configured (RAMSES_AMD_OVERLAP != 0 and every emission oct lies in the boundary layer)
  int prefetched = 0;           // 0 no; 1 packed and staged to the host buffer; 2 exchanged and unpacked (RCCL)
};
MpiRes g_mr;

int self_fill(MpiRes &M, double *d_u, hipStream_t s) {
  // periodic copy of the rank's own interior into the ghost layers of the directions it spans completely
  const HaloPlan &P = M.plan;
  const int n[3] = {P.nx, P.ny, P.nz};
  const long pitch[3] = {1, P.pitch_y, P.pitch_z};
  for (int d = 0; d < 3; d++) {
    if (!(P.self_axes >> d & 1)) continue;
    for (int hi = 0; hi < 2; hi++) {
      int org_s[3], org_d[3], ext[3];
      for (int e = 0; e < 3; e++) {
        const bool full = !(P.self_axes >> e & 1) || e < d;   // shared with peers (filled by the unpack) or already done
        if (e == d) {
          ext[e] = 2;
          org_d[e] = hi ? n[e] + 2 : 0;          // ghost cells beyond the face
          org_s[e] = hi ? 2 : n[e];              // interior cells next to the opposite face
        } else {
          ext[e] = full ? n[e] + 4 : n[e];
          org_s[e] = org_d[e] = full ? 0 : 2;
        }
      }
      BoxCopyArgs A;
      A.src = d_u; A.dst = d_u;
      A.ex = ext[0]; A.ey = ext[1]; A.ez = ext[2]; A.nvar = M.nvar;
      A.s_off = org_s[0] * pitch[0] + org_s[1] * pitch[1] + org_s[2] * pitch[2];
      A.d_off = org_d[0] * pitch[0] + org_d[1] * pitch[1] + org_d[2] * pitch[2];
      A.s_py = A.d_py = P.pitch_y; A.s_pz = A.d_pz = P.pitch_z; A.s_pv = A.d_pv = P.pitch_var;
      HCHK(launch_box_copy(A, s), "periodic self-fill launch");
    }
  }
  return 0;
}

// Whatever the communication stream still has in flight (pack / exchange / unpack of the prefetched halo, which use
// sendbuf, recvbuf, h_send and the ghost layer of bnew) must be over before the compute stream touches the same
// buffers in an order other than amr_step's: every consumer that does not itself consume the prefetch joins first.
int join_comm(MpiRes &M) {
  if (M.prefetched) HCHK(hipStreamWaitEvent(M.s_comp, M.ev_comm, 0), "stream wait");
  return 0;
}

OctListArgs list_args(MpiRes &M, double *brick, double *buf, const int64_t *org, const int *src, int n) {
  OctListArgs A;
  A.brick = brick; A.buf = buf; A.org = org; A.src = src; A.n = n; A.nvar = M.nvar;
  A.pitch_y = M.plan.pitch_y; A.pitch_z = M.plan.pitch_z; A.pitch_var = M.plan.pitch_var;
  return A;
}
}  // namespace

extern "C" {

// Host-side plan only (no device): the box of the rank's octs and the brick offsets of the list
// entries; what ramses_amd_mpires_setup uploads.  Exposed for the CPU tests of the multi-rank path.
// out_box = {olo x,y,z, odim x,y,z, self_axes, n_rc_used}; act_org[ngrid], em_org[sum em_ngrid],
// rc_src / rc_org [capacity rc_cap >= n_rc_used] may be NULL.
int ramses_amd_halo_plan(int ilevel, int ngrid, const int *igrid, const double *xg, int64_t ngridmax, int ncpu,
                         const int *em_ngrid, const int *em_igrid, const int *rc_ngrid, const int *rc_igrid,
                         int *out_box, int64_t *act_org, int64_t *em_org, int *rc_src, int64_t *rc_org, int64_t rc_cap) {
  if (!igrid || !xg || !em_ngrid || !rc_ngrid || !out_box || ncpu < 1) return failf(RAMSES_AMD_EINVAL, "NULL argument");
  HaloPlan P;
  if (!build_halo_plan(ilevel, ngrid, igrid, xg, ngridmax, ncpu, em_ngrid, em_igrid, rc_ngrid, rc_igrid, P))
    return failf(RAMSES_AMD_EUNSUPPORTED, "%s", P.error.c_str());
  for (int d = 0; d < 3; d++) { out_box[d] = P.olo[d]; out_box[3 + d] = P.odim[d]; }
  out_box[6] = P.self_axes;
  out_box[7] = (int)P.rc_src.size();
  if (act_org) std::memcpy(act_org, P.act_org.data(), sizeof(int64_t) * P.act_org.size());
  if (em_org) std::memcpy(em_org, P.em_org.data(), sizeof(int64_t) * P.em_org.size());
  if (rc_src && rc_org) {
    if ((int64_t)P.rc_src.size() > rc_cap) return failf(RAMSES_AMD_EINVAL, "rc_cap too small");
    std::memcpy(rc_src, P.rc_src.data(), sizeof(int) * P.rc_src.size());
    std::memcpy(rc_org, P.rc_org.data(), sizeof(int64_t) * P.rc_org.size());
  }
  return 0;
}

int ramses_amd_mpires_active(void) { return g_mr.valid ? 1 : 0; }

// (Re)build the device image of the level: box, ghost brick, communicator lists; load the state from the
// host arrays (active octs AND ghost octs: the host has just exchanged them itself).
int ramses_amd_mpires_setup(const ramses_amd_hydro_params *p, int ilevel, int ngrid, const int *igrid, const double *xg,
                            int64_t ngridmax, int64_t ncoarse, int nx_loc, const double *uold, const double *unew,
                            int ncpu, int myid, const int *em_ngrid, const int *em_igrid, const int *rc_ngrid,
                            const int *rc_igrid) {
  if (!p || !igrid || !xg || !uold || !unew || !em_ngrid || !rc_ngrid) return failf(RAMSES_AMD_EINVAL, "NULL argument");
  if (p->ndim != 3 || p->nvar < 5 || p->nvar > 7) return failf(RAMSES_AMD_EUNSUPPORTED, "device path implements NDIM=3, NVAR=5..7");
  if (nx_loc != 1) return failf(RAMSES_AMD_EUNSUPPORTED, "device path needs a periodic box with nx=ny=nz=1 (got nx_loc=%d)", nx_loc);
  if (ncpu < 1 || myid < 1 || myid > ncpu) return failf(RAMSES_AMD_EINVAL, "bad ncpu/myid");
  MpiRes &M = g_mr;
  if (M.valid && M.host_stale) return failf(RAMSES_AMD_EINVAL, "mpires_setup: the host array is stale; sync first");
  M.valid = false;
  if (!build_halo_plan(ilevel, ngrid, igrid, xg, ngridmax, ncpu, em_ngrid, em_igrid, rc_ngrid, rc_igrid, M.plan))
    return failf(RAMSES_AMD_EUNSUPPORTED, "%s", M.plan.error.c_str());
  HaloPlan &P = M.plan;
  M.level = ilevel; M.ngrid = ngrid; M.nvar = p->nvar; M.ncpu = ncpu; M.myid = myid;
  M.ncoarse = ncoarse; M.ngridmax = ngridmax; M.ncell = ncoarse + 8 * ngridmax;
  M.h_uold = uold; M.h_unew = unew;
  M.brick.nx = P.nx; M.brick.ny = P.ny; M.brick.nz = P.nz; M.brick.ng = 2;
  M.brick.pitch_y = P.pitch_y; M.brick.pitch_z = P.pitch_z; M.brick.pitch_var = P.pitch_var;
  const int nvar = M.nvar;
  const int nem = P.em_first[ncpu], nrcm = P.rc_first[ncpu], nrc = (int)P.rc_src.size();
  hipStream_t s = nullptr;
  const size_t bb = sizeof(double) * nvar * (size_t)P.pitch_var;
  HCHK(M.bold.ensure(bb), "hipMalloc brick"); HCHK(M.bnew.ensure(bb), "hipMalloc brick");
  HCHK(hipMemsetAsync(M.bold.p, 0, bb, s), "memset"); HCHK(hipMemsetAsync(M.bnew.p, 0, bb, s), "memset");
  HCHK(M.vec.ensure(sizeof(double) * nvar * (size_t)M.ncell), "hipMalloc uold");
  HCHK(M.red.ensure(sizeof(double) * 4), "hipMalloc");
  // lists
  std::vector<int> ligrid((size_t)ngrid + nrc);
  std::vector<int64_t> lorg((size_t)ngrid + nrc);
  for (int g = 0; g < ngrid; g++) { ligrid[g] = igrid[g]; lorg[g] = P.act_org[g]; }
  for (int r = 0; r < nrc; r++) { ligrid[ngrid + r] = rc_igrid[P.rc_src[r]]; lorg[ngrid + r] = P.rc_org[r]; }
  HCHK(M.load_igrid.ensure(sizeof(int) * ligrid.size()), "hipMalloc"); HCHK(M.load_org.ensure(sizeof(int64_t) * lorg.size()), "hipMalloc");
  HCHK(hipMemcpyAsync(M.load_igrid.p, ligrid.data(), sizeof(int) * ligrid.size(), hipMemcpyHostToDevice, s), "H2D");
  HCHK(hipMemcpyAsync(M.load_org.p, lorg.data(), sizeof(int64_t) * lorg.size(), hipMemcpyHostToDevice, s), "H2D");
  HCHK(M.em_org.ensure(sizeof(int64_t) * (size_t)nem), "hipMalloc"); HCHK(M.rc_src.ensure(sizeof(int) * (size_t)nrc), "hipMalloc");
  HCHK(M.rc_org.ensure(sizeof(int64_t) * (size_t)nrc), "hipMalloc");
  if (nem) HCHK(hipMemcpyAsync(M.em_org.p, P.em_org.data(), sizeof(int64_t) * nem, hipMemcpyHostToDevice, s), "H2D");
  if (nrc) {
    HCHK(hipMemcpyAsync(M.rc_src.p, P.rc_src.data(), sizeof(int) * nrc, hipMemcpyHostToDevice, s), "H2D");
    HCHK(hipMemcpyAsync(M.rc_org.p, P.rc_org.data(), sizeof(int64_t) * nrc, hipMemcpyHostToDevice, s), "H2D");
  }
  HCHK(M.sendbuf.ensure(sizeof(double) * 8 * nvar * (size_t)nem), "hipMalloc"); HCHK(M.recvbuf.ensure(sizeof(double) * 8 * nvar * (size_t)nrcm), "hipMalloc");
  HCHK(M.h_send.ensure(sizeof(double) * 8 * nvar * (size_t)nem), "hipHostMalloc"); HCHK(M.h_recv.ensure(sizeof(double) * 8 * nvar * (size_t)nrcm), "hipHostMalloc");
  M.peers.clear(); M.send_off.clear(); M.send_cnt.clear(); M.recv_off.clear(); M.recv_cnt.clear();
  M.f_send_off.assign(ncpu + 1, 0); M.f_recv_off.assign(ncpu + 1, 0);
  const int64_t per = 8 * (int64_t)nvar;
  for (int c = 0; c < ncpu; c++) {
    M.f_send_off[c + 1] = (int64_t)P.em_first[c + 1] * per;
    M.f_recv_off[c + 1] = (int64_t)P.rc_first[c + 1] * per;
    if (em_ngrid[c] > 0 || rc_ngrid[c] > 0) {
      if (c == myid - 1) return failf(RAMSES_AMD_EINVAL, "a rank cannot be its own peer");
      M.peers.push_back(c);
      M.send_off.push_back((int64_t)P.em_first[c] * per); M.send_cnt.push_back((int64_t)em_ngrid[c] * per);
      M.recv_off.push_back((int64_t)P.rc_first[c] * per); M.recv_cnt.push_back((int64_t)rc_ngrid[c] * per);
    }
  }
  // state: host cell vectors -> brick (interior + ghost octs)
  HCHK(hipMemcpyAsync(M.vec.p, uold, sizeof(double) * nvar * (size_t)M.ncell, hipMemcpyHostToDevice, s), "H2D uold");
  PackArgs A;
  A.igrid = M.load_igrid.as<int>(); A.octorg = reinterpret_cast<const long *>(M.load_org.p);
  A.ngrid = ngrid + nrc; A.n = 0; A.nvar = nvar;
  A.ncoarse = ncoarse; A.ngridmax = ngridmax; A.ncell = M.ncell; A.pitch_var = P.pitch_var;
  A.pitch_y = P.pitch_y; A.pitch_z = P.pitch_z;
  A.brick = M.bold.as<double>(); A.cellvec = M.vec.as<double>();
  HCHK(launch_oct_copy(A, true, s), "gather launch");
  if (int rc = self_fill(M, M.bold.as<double>(), s)) return rc;
  HCHK(hipStreamSynchronize(s), "sync");
  if (!M.s_comp) {
    HCHK(hipStreamCreateWithFlags(&M.s_comp, hipStreamNonBlocking), "hipStreamCreate");
    HCHK(hipStreamCreateWithFlags(&M.s_comm, hipStreamNonBlocking), "hipStreamCreate");
    HCHK(hipEventCreateWithFlags(&M.ev_shell, hipEventDisableTiming), "hipEventCreate");
    HCHK(hipEventCreateWithFlags(&M.ev_comm, hipEventDisableTiming), "hipEventCreate");
  }
  {
    const char *e = getenv("RAMSES_AMD_OVERLAP");
    M.overlap = !(e && e[0] == '0');
    // every emission oct must be produced by the shell launch: within one oct of a face shared with a peer
    for (int m = 0; m < nem && M.overlap; m++) {
      bool edge = false;
      for (int d = 0; d < 3; d++) {
        if (P.self_axes >> d & 1) continue;
        const int r = oct_coord(xg, ngridmax, em_igrid[m], d, P.no) - P.olo[d];
        if (r == 0 || r == P.odim[d] - 1) edge = true;
      }
      if (!edge) M.overlap = false;
    }
  }
  M.prefetched = 0;
  M.valid = true; M.host_stale = false; M.new_ready = false; M.nexchanges = 0;
  return 0;
}

// which column of the level's host arrays is xx?  +ivar: uold(1,ivar); -ivar: unew(1,ivar); 0: neither
int ramses_amd_mpires_which(const double *xx) {
  const MpiRes &M = g_mr;
  if (!M.valid || !xx) return 0;
  for (int v = 0; v < M.nvar; v++) {
    if (xx == M.h_uold + (size_t)v * M.ncell) return v + 1;
    if (xx == M.h_unew + (size_t)v * M.ncell) return -(v + 1);
  }
  return 0;
}

#define NEED_VALID(who) do { if (!g_mr.valid) return failf(RAMSES_AMD_EINVAL, "%s: no resident level (ramses_amd_mpires_setup)", who); } while (0)

// courant_fine on the rank's brick: out4 = {dt_loc (min with dt_in), mass_loc, sum(E*vol), eint_loc};
// the shim reduces over the ranks as the reference does (hydro/courant_fine.f90:133-140)
int ramses_amd_mpires_courant(const ramses_amd_hydro_params *p, double dx, double dt_in, double *out4) {
  NEED_VALID("courant_fine");
  if (!p || !out4) return failf(RAMSES_AMD_EINVAL, "NULL argument");
  MpiRes &M = g_mr;
  hipStream_t s = M.s_comp;
  if (int rc = join_comm(M)) return rc;
  if (int rc = ramses_amd_courant_init(p, dx, M.red.as<double>(), s)) return rc;
  if (int rc = ramses_amd_courant_brick(p, &M.brick, M.bold.as<double>(), nullptr, dx, M.red.as<double>(), s)) return rc;
  HCHK(hipMemcpyAsync(out4, M.red.p, sizeof(double) * 4, hipMemcpyDeviceToHost, s), "D2H courant");
  HCHK(hipStreamSynchronize(s), "sync");
  if (dt_in < out4[0]) out4[0] = dt_in;
  return 0;
}

// set_unew + godunov_fine: bold (ghosts current) -> bnew interior.  With the overlap on, the launch is split:
// shell (every cell within one oct of a face) on the compute stream; then, on the communication stream, the
// + 0.0 of the reverse exchange, the pack of the NEW state and its way to the peers (RCCL: the whole exchange
// and the unpack into the ghost layer of bnew; host transport: the copy into the pinned send buffer) run while
// the compute stream sweeps the interior.  Same values as the serial order.
int ramses_amd_mpires_godunov(const ramses_amd_hydro_params *p, double dx, double dt) {
  NEED_VALID("godunov_fine");
  MpiRes &M = g_mr;
  if (int rc = join_comm(M)) return rc;     // a repeated call: the previous prefetch may still be writing bnew's ghosts
  M.prefetched = 0;
  if (!M.overlap) {
    if (int rc = ramses_amd_godunov_brick(p, &M.brick, M.bold.as<double>(), nullptr, M.bnew.as<double>(), dx, dt, M.s_comp)) return rc;
    M.new_ready = true;
    return 0;
  }
  if (int rc = ramses_amd_godunov_brick_shell(p, &M.brick, M.bold.as<double>(), nullptr, M.bnew.as<double>(), dx, dt, M.s_comp)) return rc;
  HCHK(hipEventRecord(M.ev_shell, M.s_comp), "event record");
  HCHK(hipStreamWaitEvent(M.s_comm, M.ev_shell, 0), "stream wait");
  {
    hipStream_t s = M.s_comm;
    const int nem = M.plan.em_first[M.ncpu];
    OctListArgs Z = list_args(M, M.bnew.as<double>(), nullptr, M.em_org.as<int64_t>(), nullptr, nem);
    HCHK(launch_oct_list(Z, OL_ADDZERO, s), "reverse launch");
    OctListArgs A = list_args(M, M.bnew.as<double>(), M.sendbuf.as<double>(), M.em_org.as<int64_t>(), nullptr, nem);
    HCHK(launch_oct_list(A, OL_PACK, s), "halo pack launch");
    if (ramses_amd_rccl_ready()) {
      if (int rc = ramses_amd_rccl_exchange((int)M.peers.size(), M.peers.data(), M.sendbuf.as<double>(), M.send_off.data(), M.send_cnt.data(),
                                            M.recvbuf.as<double>(), M.recv_off.data(), M.recv_cnt.data(), s)) return rc;
      OctListArgs B = list_args(M, M.bnew.as<double>(), M.recvbuf.as<double>(), M.rc_org.as<int64_t>(), M.rc_src.as<int>(), (int)M.plan.rc_src.size());
      HCHK(launch_oct_list(B, OL_UNPACK, s), "halo unpack launch");
      if (int rc = self_fill(M, M.bnew.as<double>(), s)) return rc;
      M.prefetched = 2;
    } else {
      if (nem) HCHK(hipMemcpyAsync(M.h_send.p, M.sendbuf.p, sizeof(double) * 8 * M.nvar * (size_t)nem, hipMemcpyDeviceToHost, s), "D2H halo");
      M.prefetched = 1;
    }
    HCHK(hipEventRecord(M.ev_comm, s), "event record");
  }
  if (int rc = ramses_amd_godunov_brick_interior(p, &M.brick, M.bold.as<double>(), nullptr, M.bnew.as<double>(), dx, dt, M.s_comp)) return rc;
  M.new_ready = true;
  return 0;
}

// make_virtual_reverse_dp(unew(1,ivar),ilevel) on a fully refined level: the reception cells of unew are
// zero (set_unew, hydro/godunov_fine.f90:92-104; no finer level has added corrections), so every emission
// cell receives + 0.0 once per peer (virtual_boundaries.f90:857-867): kept, because -0.0 + 0.0 = +0.0.
int ramses_amd_mpires_reverse_unew(void) {
  NEED_VALID("make_virtual_reverse_dp");
  MpiRes &M = g_mr;
  if (!M.new_ready) return failf(RAMSES_AMD_EINVAL, "make_virtual_reverse_dp(unew): no godunov_fine result pending");
  if (M.prefetched) return 0;     // done on the communication stream before the pack (ramses_amd_mpires_godunov)
  OctListArgs A = list_args(M, M.bnew.as<double>(), nullptr, M.em_org.as<int64_t>(), nullptr, M.plan.em_first[M.ncpu]);
  HCHK(launch_oct_list(A, OL_ADDZERO, M.s_comp), "reverse launch");
  return 0;
}

int ramses_amd_mpires_set_uold(void) {
  NEED_VALID("set_uold");
  MpiRes &M = g_mr;
  if (!M.new_ready) return failf(RAMSES_AMD_EINVAL, "set_uold: no godunov_fine result pending");
  Buf t = M.bold; M.bold = M.bnew; M.bnew = t;
  M.new_ready = false; M.host_stale = true;
  return 0;
}

// make_virtual_fine_dp(uold(1,1:nvar),ilevel) over RCCL: pack -> one grouped send/recv -> unpack -> periodic self-fill
int ramses_amd_mpires_halo_forward(void) {
  NEED_VALID("make_virtual_fine_dp");
  MpiRes &M = g_mr;
  hipStream_t s = M.s_comp;
  if (M.prefetched == 2 && !M.new_ready) {
    // the exchange of this state ran behind the interior sweep: the compute stream only has to wait for it
    HCHK(hipStreamWaitEvent(s, M.ev_comm, 0), "stream wait");
    M.prefetched = 0;
    M.nexchanges++;
    return 0;
  }
  if (int rc = join_comm(M)) return rc;
  M.prefetched = 0;
  OctListArgs A = list_args(M, M.bold.as<double>(), M.sendbuf.as<double>(), M.em_org.as<int64_t>(), nullptr, M.plan.em_first[M.ncpu]);
  HCHK(launch_oct_list(A, OL_PACK, s), "halo pack launch");
  if (int rc = ramses_amd_rccl_exchange((int)M.peers.size(), M.peers.data(), M.sendbuf.as<double>(), M.send_off.data(), M.send_cnt.data(),
                                        M.recvbuf.as<double>(), M.recv_off.data(), M.recv_cnt.data(), s)) return rc;
  OctListArgs B = list_args(M, M.bold.as<double>(), M.recvbuf.as<double>(), M.rc_org.as<int64_t>(), M.rc_src.as<int>(), (int)M.plan.rc_src.size());
  HCHK(launch_oct_list(B, OL_UNPACK, s), "halo unpack launch");
  if (int rc = self_fill(M, M.bold.as<double>(), s)) return rc;
  M.nexchanges++;
  return 0;
}

// The same with the shim's own MPI as transport (several ranks on one GPU, or no RCCL): stage_out packs on the
// device and hands pinned host buffers to the caller -- message of peer icpu at h_send + send_off[icpu-1], of
// length send_off[icpu]-send_off[icpu-1] doubles, likewise h_recv/recv_off -- stage_in unpacks what arrived.
int ramses_amd_mpires_halo_stage_out(double **h_send, const int64_t **send_off, double **h_recv, const int64_t **recv_off) {
  NEED_VALID("make_virtual_fine_dp");
  if (!h_send || !send_off || !h_recv || !recv_off) return failf(RAMSES_AMD_EINVAL, "NULL argument");
  MpiRes &M = g_mr;
  hipStream_t s = M.s_comp;
  const int nem = M.plan.em_first[M.ncpu];
  if (M.prefetched == 1 && !M.new_ready) {
    // packed and copied to the pinned buffer behind the interior sweep
    HCHK(hipEventSynchronize(M.ev_comm), "event sync");
    M.prefetched = 0;
  } else {
    if (int rc = join_comm(M)) return rc;
    M.prefetched = 0;
    OctListArgs A = list_args(M, M.bold.as<double>(), M.sendbuf.as<double>(), M.em_org.as<int64_t>(), nullptr, nem);
    HCHK(launch_oct_list(A, OL_PACK, s), "halo pack launch");
    if (nem) HCHK(hipMemcpyAsync(M.h_send.p, M.sendbuf.p, sizeof(double) * 8 * M.nvar * (size_t)nem, hipMemcpyDeviceToHost, s), "D2H halo");
    HCHK(hipStreamSynchronize(s), "sync");
  }
  *h_send = M.h_send.as<double>(); *h_recv = M.h_recv.as<double>();
  *send_off = M.f_send_off.data(); *recv_off = M.f_recv_off.data();
  return 0;
}
int ramses_amd_mpires_halo_stage_in(void) {
  NEED_VALID("make_virtual_fine_dp");
  MpiRes &M = g_mr;
  hipStream_t s = M.s_comp;
  const int nrcm = M.plan.rc_first[M.ncpu];
  if (nrcm) HCHK(hipMemcpyAsync(M.recvbuf.p, M.h_recv.p, sizeof(double) * 8 * M.nvar * (size_t)nrcm, hipMemcpyHostToDevice, s), "H2D halo");
  OctListArgs B = list_args(M, M.bold.as<double>(), M.recvbuf.as<double>(), M.rc_org.as<int64_t>(), M.rc_src.as<int>(), (int)M.plan.rc_src.size());
  HCHK(launch_oct_list(B, OL_UNPACK, s), "halo unpack launch");
  if (int rc = self_fill(M, M.bold.as<double>(), s)) return rc;
  M.nexchanges++;
  return 0;
}
// Fortran-friendly accessors of the staged buffers (addresses as integers: the shim maps them with c_f_pointer)
int ramses_amd_mpires_halo_stage_out_f90(int64_t *h_send_addr, int64_t *h_recv_addr, int64_t *send_off, int64_t *recv_off, int ncpu) {
  double *hs, *hr;
  const int64_t *so, *ro;
  if (!h_send_addr || !h_recv_addr || !send_off || !recv_off) return failf(RAMSES_AMD_EINVAL, "NULL argument");
  if (ncpu != g_mr.ncpu) return failf(RAMSES_AMD_EINVAL, "ncpu mismatch");
  if (int rc = ramses_amd_mpires_halo_stage_out(&hs, &so, &hr, &ro)) return rc;
  *h_send_addr = (int64_t)(intptr_t)hs; *h_recv_addr = (int64_t)(intptr_t)hr;
  for (int c = 0; c <= ncpu; c++) { send_off[c] = so[c]; recv_off[c] = ro[c]; }
  return 0;
}

// active octs of the level back into the host array (no-op when it is current)
int ramses_amd_mpires_sync_host(double *uold) {
  MpiRes &M = g_mr;
  if (!M.valid || !M.host_stale) return 0;
  if (uold != M.h_uold) return failf(RAMSES_AMD_EINVAL, "sync_host: not the array the level was loaded from");
  hipStream_t s = M.s_comp;
  HCHK(hipStreamSynchronize(M.s_comm), "sync");
  PackArgs A;
  A.igrid = M.load_igrid.as<int>(); A.octorg = reinterpret_cast<const long *>(M.load_org.p);
  A.ngrid = M.ngrid + (int)M.plan.rc_src.size();   // the ghost octs too: the host's reception cells stay what an exchange would leave
  A.n = 0; A.nvar = M.nvar;
  A.ncoarse = M.ncoarse; A.ngridmax = M.ngridmax; A.ncell = M.ncell; A.pitch_var = M.plan.pitch_var;
  A.pitch_y = M.plan.pitch_y; A.pitch_z = M.plan.pitch_z;
  A.brick = M.bold.as<double>(); A.cellvec = M.vec.as<double>();
  // the device copy of the cell vector dates from the setup: cells outside the level (coarser levels, free slots) are
  // taken from the host as they are NOW, so that the copy back changes the level's octs and nothing else
  HCHK(hipMemcpyAsync(M.vec.p, uold, sizeof(double) * M.nvar * (size_t)M.ncell, hipMemcpyHostToDevice, s), "H2D uold");
  HCHK(launch_oct_copy(A, false, s), "scatter launch");
  HCHK(hipMemcpyAsync(uold, M.vec.p, sizeof(double) * M.nvar * (size_t)M.ncell, hipMemcpyDeviceToHost, s), "D2H uold");
  HCHK(hipStreamSynchronize(s), "sync");
  M.host_stale = false;
  return 0;
}

int ramses_amd_mpires_invalidate(void) {
  MpiRes &M = g_mr;
  if (M.valid && M.host_stale) return failf(RAMSES_AMD_EINVAL, "invalidate: the host array is stale; sync first");
  M.valid = false;
  return 0;
}

}  // extern "C"

#include "warm.hpp"
RAMSES_AMD_TU_WARM(capi_mpi)
