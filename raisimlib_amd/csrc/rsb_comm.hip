// rsb_comm.hip — multi-GPU without Python: the obs all-gather over RCCL / xGMI (rsb_comm_*, rsb_allgather_obs) and the
// peer-mapped obs exchange without a collective (rsb_obs_peer_*).  One process per GPU, envs sharded contiguously; nothing inside
// integrate() communicates.  Upstream has no counterpart (RaiSim is single-process).
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cstring>
#include <string>
#include <type_traits>
#include <vector>

#include "rsb_world.h"

// RCCL is bound by dlopen + hand-written prototypes (a single-GPU host never loads it).  Where its header is installed, the constants this file
// copies from it are checked against it AT BUILD TIME - the header is included for these assertions only, no symbol of it is used.
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
#define RSB_HAVE_RCCL_HEADER 1
#endif

using namespace rsbw;

// ---- RCCL (loaded at run time: a single-GPU host never needs it) -------------------------------------------------
namespace {
struct Rccl {
  struct UniqueId { char internal[RSB_COMM_ID_BYTES]; };     // ncclUniqueId (rccl.h: 128 opaque bytes, passed by value)
  int (*GetUniqueId)(UniqueId*) = nullptr;
  int (*CommInitRank)(void**, int, UniqueId, int) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  int (*GetVersion)(int*) = nullptr;
  void* handle = nullptr;
  std::string error;
};
Rccl* rccl() {
  static Rccl r;
  static bool tried = false;
  if (tried) return r.handle ? &r : nullptr;
  tried = true;
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (const char* n : names) { r.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL); if (r.handle) break; }
  if (!r.handle) { r.error = std::string("cannot load librccl.so.1: ") + dlerror(); return nullptr; }
  r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(r.handle, "ncclGetUniqueId"));
  r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(r.handle, "ncclCommInitRank"));
  r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(r.handle, "ncclCommDestroy"));
  r.AllGather = reinterpret_cast<decltype(r.AllGather)>(dlsym(r.handle, "ncclAllGather"));
  r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(r.handle, "ncclGetErrorString"));
  r.GetVersion = reinterpret_cast<decltype(r.GetVersion)>(dlsym(r.handle, "ncclGetVersion"));      // (optional: rsb_comm_rccl_version)
  if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllGather || !r.GetErrorString) {
    r.error = "librccl.so.1 lacks an expected ncclXxx symbol"; dlclose(r.handle); r.handle = nullptr; return nullptr;
  }
  return &r;
}
Rccl* need_rccl() {
  Rccl* r = rccl();
  if (!r) rsb::set_error("RCCL unavailable (multi-GPU entry points need /opt/rocm/lib/librccl.so.1)");
  return r;
}
constexpr int kNcclFloat32 = 7;   // ncclDataType_t::ncclFloat32 (rccl.h)
#ifdef RSB_HAVE_RCCL_HEADER
static_assert((int)ncclFloat32 == kNcclFloat32, "rccl.h: ncclFloat32 is not the value rsb_comm.hip passes to ncclAllGather");
static_assert((int)ncclSuccess == 0, "rccl.h: ncclSuccess is not 0");
static_assert(sizeof(ncclUniqueId) == RSB_COMM_ID_BYTES, "rccl.h: ncclUniqueId is not RSB_COMM_ID_BYTES bytes");
static_assert(std::is_same<decltype(&ncclAllGather), ncclResult_t (*)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t)>::value, "rccl.h: ncclAllGather's signature changed");
static_assert(std::is_same<decltype(&ncclCommInitRank), ncclResult_t (*)(ncclComm_t*, int, ncclUniqueId, int)>::value, "rccl.h: ncclCommInitRank's signature changed");
constexpr int kRcclHeaderVersion = NCCL_VERSION_CODE;
#else
constexpr int kRcclHeaderVersion = 0;
#endif
#define NCCL_TRY(expr)                                                                                   \
  do {                                                                                                   \
    int r_ = (expr);                                                                                     \
    if (r_ != 0) { rsb::set_error(std::string(#expr) + ": " + R->GetErrorString(r_)); return RSB_E_HIP; } \
  } while (0)
}  // namespace

extern "C" {

int rsb_comm_rccl_version(int* runtime_version, int* header_version) {
  if (header_version) *header_version = kRcclHeaderVersion;
  if (runtime_version) {
    *runtime_version = 0;
    Rccl* R = need_rccl();
    if (!R) return RSB_E_UNSUPPORTED;
    if (R->GetVersion) { int v = 0; NCCL_TRY(R->GetVersion(&v)); *runtime_version = v; }
  }
  return RSB_OK;
}

int rsb_comm_get_unique_id(char id[RSB_COMM_ID_BYTES]) {
  if (!id) return RSB_E_INVALID;
  Rccl* R = need_rccl();
  if (!R) return RSB_E_UNSUPPORTED;
  Rccl::UniqueId u;
  NCCL_TRY(R->GetUniqueId(&u));
  std::memcpy(id, u.internal, RSB_COMM_ID_BYTES);
  return RSB_OK;
}

int rsb_comm_init(rsb_world* w, int n_ranks, int rank, const char id[RSB_COMM_ID_BYTES]) {
  if (!w || !id || n_ranks < 1 || rank < 0 || rank >= n_ranks) { rsb::set_error("rsb_comm_init: bad argument"); return RSB_E_INVALID; }
  if (w->comm) { rsb::set_error("rsb_comm_init: the world already has a communicator"); return RSB_E_STATE; }
  Rccl* R = need_rccl();
  if (!R) return RSB_E_UNSUPPORTED;
  HIP_TRY(hipSetDevice(w->device));
  Rccl::UniqueId u;
  std::memcpy(u.internal, id, RSB_COMM_ID_BYTES);
  NCCL_TRY(R->CommInitRank(&w->comm, n_ranks, u, rank));
  w->comm_ranks = n_ranks; w->comm_rank = rank;
  return RSB_OK;
}

int rsb_comm_destroy(rsb_world* w) {
  if (!w || !w->comm) return RSB_OK;
  Rccl* R = rccl();
  if (R) { (void)hipSetDevice(w->device); (void)hipStreamSynchronize(stream_of(w)); (void)R->CommDestroy(w->comm); }
  w->comm = nullptr; w->comm_ranks = 0;
  return RSB_OK;
}

// ---- peer-mapped obs exchange: no collective, no copy kernel (see rsb.h) -------------------------------------------------
__global__ void obs_peer_wait_kernel(const uint32_t* flags, int n, uint32_t step) {
  // fallback of rsb_obs_peer_wait when the stream cannot wait on a memory value: lane p spins until rank p's flag has the step
  const int p = threadIdx.x;
  if (p < n) while ((int32_t)(__hip_atomic_load(flags + p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - step) < 0) __builtin_amdgcn_s_sleep(8);
}

int rsb_obs_peer_create(rsb_world* w, int n_ranks, int rank, const int32_t* collision_indices, int n_force_slots, char handle[RSB_OBS_HANDLE_BYTES]) {
  if (!w || n_ranks < 1 || n_ranks > RSB_MAX_RANKS || rank < 0 || rank >= n_ranks || n_force_slots < 0 || n_force_slots > RSB_MAX_COLLISIONS) {
    rsb::set_error("rsb_obs_peer_create: bad argument (1 <= n_ranks <= RSB_MAX_RANKS)"); return RSB_E_INVALID;
  }
  if (w->peer.base) { rsb::set_error("rsb_obs_peer_create: the world already has an exchange (rsb_obs_peer_destroy first)"); return RSB_E_STATE; }
  HIP_TRY(hipSetDevice(w->device));
  rsb_world::Peer& P = w->peer;
  P.ranks = n_ranks; P.rank = rank; P.slots = n_force_slots; P.od = w->blob.nq + w->blob.nv + 3 * n_force_slots;
  P.idx.clear();
  if (collision_indices) {
    for (int i = 0; i < n_force_slots; ++i) {
      if (collision_indices[i] < 0 || collision_indices[i] >= w->blob.ncol) { rsb::set_error("rsb_obs_peer_create: collision index out of range"); return RSB_E_INVALID; }
      P.idx.push_back(collision_indices[i]);
    }
  }
  const size_t bufsz = (size_t)n_ranks * w->N * P.od;
  P.bytes = 2 * bufsz * sizeof(float) + (2 * RSB_MAX_RANKS + 4) * sizeof(uint32_t);
  // fine-grained memory: stores of OTHER devices' kernels (and their system-scope flag writes) become visible without a kernel
  // boundary on this device; plain hipMalloc is the fallback where the runtime refuses the flag
  // Coarse-grained memory gives NO such guarantee (a remote rank's write-through stores and the flag a consumer polls may sit in a
  // cache until a kernel boundary: a wait can hang), so without fine-grained memory the exchange is refused - RSB_OBS_PEER_COARSE=1
  // forces plain hipMalloc for single-device diagnostics.
  static const bool coarse = std::getenv("RSB_OBS_PEER_COARSE") != nullptr;
  if (coarse) HIP_TRY(hipMalloc(&P.base, P.bytes));
  else if (hipExtMallocWithFlags(&P.base, P.bytes, hipDeviceMallocFinegrained) != hipSuccess) {
    (void)hipGetLastError(); P.base = nullptr;
    rsb::set_error("rsb_obs_peer_create: no fine-grained device memory on this system (the exchange's visibility rests on it); use the RCCL all-gather");
    return RSB_E_UNSUPPORTED;
  }
  auto fail = [&](hipError_t e) {      // nothing half-created survives an error: a retry must not see "already has an exchange"
    rsb::set_error(std::string("rsb_obs_peer_create: ") + hipGetErrorString(e));
    (void)hipFree(P.base); P.base = nullptr;
    if (P.d_idx) { (void)hipFree(P.d_idx); P.d_idx = nullptr; }
    return RSB_E_HIP;
  };
  hipError_t e = hipMemsetAsync(P.base, 0, P.bytes, stream_of(w));
  if (e != hipSuccess) return fail(e);
  if (!P.idx.empty()) {
    if ((e = hipMalloc(&P.d_idx, P.idx.size() * sizeof(int32_t))) != hipSuccess) return fail(e);
    if ((e = hipMemcpyAsync(P.d_idx, P.idx.data(), P.idx.size() * sizeof(int32_t), hipMemcpyHostToDevice, stream_of(w))) != hipSuccess) return fail(e);
  }
  if ((e = hipStreamSynchronize(stream_of(w))) != hipSuccess) return fail(e);
  if (handle) {
    std::memset(handle, 0, RSB_OBS_HANDLE_BYTES);
    hipIpcMemHandle_t h;
    static_assert(sizeof(hipIpcMemHandle_t) <= RSB_OBS_HANDLE_BYTES, "IPC handle does not fit RSB_OBS_HANDLE_BYTES");
    if (hipIpcGetMemHandle(&h, P.base) == hipSuccess) std::memcpy(handle, &h, sizeof h);
    else (void)hipGetLastError();       // (no IPC on this system: rsb_obs_peer_connect_ptrs within one process still works)
  }
  return RSB_OK;
}

static int obs_peer_finish_connect(rsb_world* w) {
  // diagnostic: force the one-wave wait kernel instead of the stream's memory-wait packet
  w->peer.wait_by_kernel = std::getenv("RSB_OBS_PEER_WAIT_KERNEL") != nullptr;
  // a RE-connect starts the step numbers again at 0: the flag words and the arrival counter must not keep the numbers of the earlier
  // connection, or the first waits (>= 1) would pass on stale rows.  (The first connect finds them zeroed by rsb_obs_peer_create; the ranks
  // of a reconnecting job must meet at a barrier between their connects and their first control step, like at start-up.)
  rsb_world::Peer& P = w->peer;
  if (P.step != 0) {
    const size_t bufsz = (size_t)P.ranks * w->N * P.od;
    HIP_TRY(hipSetDevice(w->device));
    HIP_TRY(hipMemsetAsync(static_cast<float*>(P.base) + 2 * bufsz, 0, (2 * RSB_MAX_RANKS + 4) * sizeof(uint32_t), stream_of(w)));
    HIP_TRY(hipStreamSynchronize(stream_of(w)));
  }
  w->peer.connected = true; w->peer.step = 0;
  return RSB_OK;
}

int rsb_obs_peer_connect(rsb_world* w, const char* handles) {
  if (!w || !handles) { rsb::set_error("rsb_obs_peer_connect: bad argument"); return RSB_E_INVALID; }
  if (!w->peer.base) { rsb::set_error("rsb_obs_peer_connect: call rsb_obs_peer_create first"); return RSB_E_STATE; }
  HIP_TRY(hipSetDevice(w->device));
  rsb_world::Peer& P = w->peer;
  for (int p = 0; p < P.ranks; ++p) {
    if (p == P.rank) { P.peer_base[p] = P.base; continue; }
    hipIpcMemHandle_t h;
    std::memcpy(&h, handles + (size_t)p * RSB_OBS_HANDLE_BYTES, sizeof h);
    void* ptr = nullptr;
    HIP_TRY(hipIpcOpenMemHandle(&ptr, h, hipIpcMemLazyEnablePeerAccess));
    P.peer_base[p] = ptr; P.imported[p] = true;
  }
  return obs_peer_finish_connect(w);
}

int rsb_obs_peer_connect_ptrs(rsb_world* w, void* const* bases) {
  if (!w || !bases) { rsb::set_error("rsb_obs_peer_connect_ptrs: bad argument"); return RSB_E_INVALID; }
  if (!w->peer.base) { rsb::set_error("rsb_obs_peer_connect_ptrs: call rsb_obs_peer_create first"); return RSB_E_STATE; }
  rsb_world::Peer& P = w->peer;
  for (int p = 0; p < P.ranks; ++p) {
    if (p != P.rank && !bases[p]) { rsb::set_error("rsb_obs_peer_connect_ptrs: null base pointer"); return RSB_E_INVALID; }
    P.peer_base[p] = p == P.rank ? P.base : bases[p];
  }
  return obs_peer_finish_connect(w);
}

void* rsb_obs_peer_base(rsb_world* w) { return w ? w->peer.base : nullptr; }

int rsb_obs_peer_wait(rsb_world* w, float** gathered) {
  if (!w) return RSB_E_INVALID;
  rsb_world::Peer& P = w->peer;
  if (!P.connected || P.step == 0) { rsb::set_error("rsb_obs_peer_wait: no control step has been issued with the exchange"); return RSB_E_STATE; }
  HIP_TRY(hipSetDevice(w->device));
  const size_t bufsz = (size_t)P.ranks * w->N * P.od;
  const int par = (int)(P.step & 1u);
  uint32_t* flags = reinterpret_cast<uint32_t*>(static_cast<float*>(P.base) + 2 * bufsz) + (size_t)par * RSB_MAX_RANKS;
  if (!P.wait_by_kernel) {
    // the command processor polls the flag words: no kernel, no CU
    for (int p = 0; p < P.ranks && !P.wait_by_kernel; ++p)
      if (hipStreamWaitValue32(stream_of(w), flags + p, P.step, hipStreamWaitValueGte, 0xffffffffu) != hipSuccess) { (void)hipGetLastError(); P.wait_by_kernel = true; }
  }
  if (P.wait_by_kernel) {
    hipLaunchKernelGGL(obs_peer_wait_kernel, dim3(1), dim3(64), 0, stream_of(w), flags, P.ranks, P.step);
    HIP_TRY(hipGetLastError());
  }
  if (gathered) *gathered = static_cast<float*>(P.base) + (size_t)par * bufsz;
  return RSB_OK;
}

int rsb_obs_peer_destroy(rsb_world* w) {
  if (!w || !w->peer.base) return RSB_OK;
  (void)hipSetDevice(w->device);
  (void)hipStreamSynchronize(stream_of(w));
  rsb_world::Peer& P = w->peer;
  for (int p = 0; p < P.ranks; ++p) if (P.imported[p]) (void)hipIpcCloseMemHandle(P.peer_base[p]);
  (void)hipFree(P.base);
  if (P.d_idx) (void)hipFree(P.d_idx);
  w->peer = rsb_world::Peer();
  return RSB_OK;
}

int rsb_allgather_obs(rsb_world* w, const int32_t* collision_indices, int n_force_slots, float* out, int space) {
  if (!w || !out || n_force_slots < 0 || n_force_slots > RSB_MAX_COLLISIONS) { rsb::set_error("rsb_allgather_obs: bad argument"); return RSB_E_INVALID; }
  if (!w->comm) { rsb::set_error("rsb_allgather_obs: call rsb_comm_init first"); return RSB_E_STATE; }
  Rccl* R = need_rccl();
  if (!R) return RSB_E_UNSUPPORTED;
  HIP_TRY(hipSetDevice(w->device));
  const size_t local = (size_t)w->N * (w->blob.nq + w->blob.nv + 3 * n_force_slots), all = local * w->comm_ranks;
  if (w->obs_local_cap < local) {
    if (w->d_obs_local) HIP_TRY(hipFree(w->d_obs_local));
    w->d_obs_local = nullptr; w->obs_local_cap = 0;
    HIP_TRY(hipMalloc(&w->d_obs_local, local * sizeof(float)));
    w->obs_local_cap = local;
  }
  float* dall = out;
  if (space == RSB_HOST) {
    if (w->obs_all_cap < all) {
      if (w->d_obs_all) HIP_TRY(hipFree(w->d_obs_all));
      w->d_obs_all = nullptr; w->obs_all_cap = 0;
      HIP_TRY(hipMalloc(&w->d_obs_all, all * sizeof(float)));
      w->obs_all_cap = all;
    }
    dall = w->d_obs_all;
  }
  int st = rsb_gather_obs(w, w->d_obs_local, collision_indices, n_force_slots, RSB_DEVICE);
  if (st != RSB_OK) return st;
  NCCL_TRY(R->AllGather(w->d_obs_local, dall, local, kNcclFloat32, w->comm, stream_of(w)));
  if (space == RSB_HOST) return copy_out(w, out, dall, all * sizeof(float), RSB_HOST);
  return RSB_OK;
}

}  // extern "C"

