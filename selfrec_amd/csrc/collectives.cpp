// (e) The row-sharded layout's exchanges behind the C ABI: thin wrappers over RCCL.
//
// The reference has no collective to replace (model/graph/XSimGCL.py:24,73: one implicit device); what these entry points
// carry is the per-layer exchange that the row partition of XSimGCL_Encoder.forward (XSimGCL.py:83-101) needs on N > 1
// GPUs: every rank owns n rows of every (N, d) table, a propagation layer reads all of them.
//
// RCCL is NOT linked: a process must hold exactly one copy of it, and which one is the host application's decision
// (PyTorch-ROCm ships its own librccl.so beside libtorch; a C++ trainer links /opt/rocm/lib/librccl.so.1).  The symbols are
// looked up at first use among what the process has already loaded (RTLD_DEFAULT), then by dlopen("librccl.so.1") /
// ("librccl.so") -- so a communicator the caller made with ITS RCCL (an ncclComm_t passed as void*) is used by the same copy.
#include <dlfcn.h>

#include <cstring>
#include <mutex>

#include "common.h"

namespace {

struct UniqueId { char internal[128]; };          // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES = 128)
static_assert(sizeof(UniqueId) == SRH_COMM_ID_BYTES, "unique id size");
constexpr int kNcclFloat32 = 7, kNcclSum = 0;     // ncclDataType_t::ncclFloat32, ncclRedOp_t::ncclSum (rccl.h)

struct Rccl {
  int (*GetUniqueId)(UniqueId*) = nullptr;
  int (*CommInitRank)(void**, int, UniqueId, int) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  int (*CommCount)(void*, int*) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
  int (*ReduceScatter)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  bool ok = false;
  char why[256] = {0};
};

Rccl& rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    void* h = RTLD_DEFAULT;
    if (!dlsym(h, "ncclAllGather")) {
      h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
      if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
      if (!h) {
        snprintf(r.why, sizeof(r.why), "RCCL is not loaded in this process and librccl.so[.1] was not found (%s)", dlerror());
        return;
      }
    }
    auto need = [&](const char* name) -> void* {
      void* p = dlsym(h, name);
      if (!p && !r.why[0]) snprintf(r.why, sizeof(r.why), "RCCL symbol %s not found", name);
      return p;
    };
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(need("ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(need("ncclCommInitRank"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(need("ncclCommDestroy"));
    r.CommCount = reinterpret_cast<decltype(r.CommCount)>(need("ncclCommCount"));
    r.AllGather = reinterpret_cast<decltype(r.AllGather)>(need("ncclAllGather"));
    r.ReduceScatter = reinterpret_cast<decltype(r.ReduceScatter)>(need("ncclReduceScatter"));
    r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(need("ncclAllReduce"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(need("ncclGetErrorString"));
    r.ok = !r.why[0];
  });
  return r;
}

#define SRH_RCCL_LOADED(r)                                     \
  do {                                                         \
    if (!(r).ok) {                                             \
      ::srh::set_error("collectives: %s", (r).why);            \
      return SRH_ERR_UNSUPPORTED;                              \
    }                                                          \
  } while (0)

#define SRH_RCCL(r, call)                                                                          \
  do {                                                                                             \
    const int e_ = (call);                                                                         \
    if (e_ != 0) {                                                                                 \
      ::srh::set_error("%s failed: %s", #call, (r).GetErrorString ? (r).GetErrorString(e_) : "?"); \
      return SRH_ERR_HIP;                                                                          \
    }                                                                                              \
  } while (0)

}  // namespace

extern "C" {

srh_status_t srh_comm_unique_id(uint8_t* h_id) {
  SRH_REQUIRE(h_id, "comm_unique_id: null argument");
  Rccl& r = rccl();
  SRH_RCCL_LOADED(r);
  UniqueId id;
  SRH_RCCL(r, r.GetUniqueId(&id));
  std::memcpy(h_id, id.internal, sizeof(id.internal));
  return SRH_OK;
}

srh_status_t srh_comm_init_rank(void** out_comm, int32_t world, int32_t rank, const uint8_t* h_id) {
  SRH_REQUIRE(out_comm && h_id, "comm_init_rank: null argument");
  SRH_REQUIRE(world >= 1 && rank >= 0 && rank < world, "comm_init_rank: rank %d of %d", rank, world);
  Rccl& r = rccl();
  SRH_RCCL_LOADED(r);
  UniqueId id;
  std::memcpy(id.internal, h_id, sizeof(id.internal));
  void* comm = nullptr;
  SRH_RCCL(r, r.CommInitRank(&comm, world, id, rank));
  *out_comm = comm;
  return SRH_OK;
}

srh_status_t srh_comm_destroy(void* comm) {
  if (!comm) return SRH_OK;
  Rccl& r = rccl();
  SRH_RCCL_LOADED(r);
  SRH_RCCL(r, r.CommDestroy(comm));
  return SRH_OK;
}

srh_status_t srh_comm_world(void* comm, int32_t* out_world) {
  SRH_REQUIRE(comm && out_world, "comm_world: null argument");
  Rccl& r = rccl();
  SRH_RCCL_LOADED(r);
  int n = 0;
  SRH_RCCL(r, r.CommCount(comm, &n));
  *out_world = n;
  return SRH_OK;
}

srh_status_t srh_allgather_rows(const float* d_rows, float* d_table, int64_t n_rows, int32_t d, void* comm, void* stream) {
  SRH_REQUIRE(d_rows && d_table && comm, "allgather_rows: null argument");
  SRH_REQUIRE(n_rows > 0 && d > 0, "allgather_rows: bad shape (%lld, %d)", (long long)n_rows, d);
  Rccl& r = rccl();
  SRH_RCCL_LOADED(r);
  SRH_RCCL(r, r.AllGather(d_rows, d_table, (size_t)n_rows * (size_t)d, kNcclFloat32, comm, srh::as_stream(stream)));
  return SRH_OK;
}

srh_status_t srh_reducescatter_rows(const float* d_table, float* d_rows, int64_t n_rows, int32_t d, void* comm, void* stream) {
  SRH_REQUIRE(d_rows && d_table && comm, "reducescatter_rows: null argument");
  SRH_REQUIRE(n_rows > 0 && d > 0, "reducescatter_rows: bad shape (%lld, %d)", (long long)n_rows, d);
  Rccl& r = rccl();
  SRH_RCCL_LOADED(r);
  SRH_RCCL(r, r.ReduceScatter(d_table, d_rows, (size_t)n_rows * (size_t)d, kNcclFloat32, kNcclSum, comm, srh::as_stream(stream)));
  return SRH_OK;
}

srh_status_t srh_allreduce_sum_f32(float* d_buf, int64_t n_elem, void* comm, void* stream) {
  SRH_REQUIRE(d_buf && comm, "allreduce_sum_f32: null argument");
  SRH_REQUIRE(n_elem > 0, "allreduce_sum_f32: bad size %lld", (long long)n_elem);
  Rccl& r = rccl();
  SRH_RCCL_LOADED(r);
  SRH_RCCL(r, r.AllReduce(d_buf, d_buf, (size_t)n_elem, kNcclFloat32, kNcclSum, comm, srh::as_stream(stream)));
  return SRH_OK;
}

}  // extern "C"
