// The one collective of an evaluation behind the C ABI (saa_comm_init / exchange / destroy).
//
// The reference is a single process (drone_risk.py:18 pins jax to one CPU device); what has to be served across
// ranks once the sample axis is sharded is the sample mean of drone_risk.py:294-296 and the Monte-Carlo
// statistics of :663-695 / drone_main_plot.py:640-652.  Both ride on ONE RCCL all-gather of each rank's record
// [fp64 partial sums | fp32 Z row] (4 MB at M = 1e6: latency-bound, an all-gather drives all seven xGMI links of a
// GPU at once, a ring all-reduce would be per-link bound), followed by rato_unpack_records on the same stream.
//
// RCCL is bound at run time (dlopen, RTLD_NOLOAD first): a torch process already holds torch's own librccl.so and
// a second copy of the library in one process would own none of its state; a plain C/C++ host gets /opt/rocm's.
// Nothing here links against torch.
#include <dlfcn.h>
#include <string.h>
#include <rccl/rccl.h>

#include <mutex>

#include "rato_common.h"

namespace {

struct RcclApi {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  bool ok = false;
};

RcclApi& rccl() {
  static RcclApi api;
  static std::once_flag once;
  std::call_once(once, [] {
    const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {   // the copy already in the process (torch's) wins
      api.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
      if (api.handle) break;
    }
    for (int i = 0; !api.handle && i < 3; ++i) api.handle = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
    if (!api.handle) return;
    api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(dlsym(api.handle, "ncclGetUniqueId"));
    api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(dlsym(api.handle, "ncclCommInitRank"));
    api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(dlsym(api.handle, "ncclCommDestroy"));
    api.AllGather = reinterpret_cast<decltype(api.AllGather)>(dlsym(api.handle, "ncclAllGather"));
    api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(dlsym(api.handle, "ncclGetErrorString"));
    api.ok = api.GetUniqueId && api.CommInitRank && api.CommDestroy && api.AllGather;
  });
  return api;
}

}  // namespace

struct rato_comm {
  ncclComm_t comm;
  int rank, world;
};

extern "C" int rato_comm_available(void) { return rccl().ok ? RATO_OK : RATO_ENOCOMM; }

extern "C" int rato_comm_unique_id(void* id_out) {
  if (!id_out) return RATO_EINVAL;
  RcclApi& a = rccl();
  if (!a.ok) return RATO_ENOCOMM;
  ncclUniqueId id;
  const ncclResult_t r = a.GetUniqueId(&id);
  if (r != ncclSuccess) return RATO_ERCCL - (int)r;
  static_assert(sizeof(ncclUniqueId) == RATO_COMM_ID_BYTES, "unique id size");
  ::memcpy(id_out, &id, sizeof(id));
  return RATO_OK;
}

extern "C" int rato_comm_init(rato_comm** out, const void* id_bytes, int32_t rank, int32_t world) {
  if (!out || !id_bytes || world < 1 || rank < 0 || rank >= world) return RATO_EINVAL;
  RcclApi& a = rccl();
  if (!a.ok) return RATO_ENOCOMM;
  ncclUniqueId id;
  ::memcpy(&id, id_bytes, sizeof(id));
  ncclComm_t c;
  const ncclResult_t r = a.CommInitRank(&c, world, id, rank);   // collective: every rank calls it, on its own device
  if (r != ncclSuccess) return RATO_ERCCL - (int)r;
  *out = new rato_comm{c, rank, world};
  return RATO_OK;
}

extern "C" int rato_comm_world(const rato_comm* c) { return c ? c->world : RATO_EINVAL; }
extern "C" int rato_comm_rank(const rato_comm* c) { return c ? c->rank : RATO_EINVAL; }

extern "C" int rato_comm_destroy(rato_comm* c) {
  if (!c) return RATO_EINVAL;
  const ncclResult_t r = rccl().CommDestroy(c->comm);
  delete c;
  return r == ncclSuccess ? RATO_OK : RATO_ERCCL - (int)r;
}

extern "C" int rato_comm_allgather(rato_comm* c, const void* send, void* recv_all, int64_t bytes, void* stream) {
  if (!c || !send || !recv_all || bytes <= 0) return RATO_EINVAL;
  const ncclResult_t r = rccl().AllGather(send, recv_all, (size_t)bytes, ncclUint8, c->comm, rato::as_stream(stream));
  return r == ncclSuccess ? RATO_OK : RATO_ERCCL - (int)r;
}

extern "C" int rato_comm_exchange(rato_comm* c, const void* record, void* all, int64_t rec_bytes, int32_t n_sums,
                                  int64_t M_local, double* total, float* Z_all, void* stream) {
  if (!c || !record || !all || rec_bytes < 8L * n_sums + 4 * M_local || rec_bytes % 8 != 0) return RATO_EINVAL;
  const int s = rato_comm_allgather(c, record, all, rec_bytes, stream);
  if (s != RATO_OK) return s;
  return rato_unpack_records(all, c->world, n_sums, M_local, rec_bytes, total, Z_all, stream);
}
