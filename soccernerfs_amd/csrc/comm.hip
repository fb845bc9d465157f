// Gradient exchange behind the C ABI: one RCCL communicator per process and the per-step all-reduce of the flat gradient buffer.
//
// Reference: DistributedDataParallel around the model (NS/pipelines/base_pipeline.py:244-246) -- NCCL bucketed all-reduce, mean over ranks.
// Here the whole gradient is ONE flat fp32 buffer, so the exchange is a single ncclAllReduce(SUM) on the caller's stream; the 1 / world
// mean is folded into the optimiser (snerf_adam_step's grad_scale).  "nccl" on ROCm is RCCL: collectives over xGMI inside a node.
//
// libsnerf does not link RCCL: the five entry points it needs are resolved with dlopen("librccl.so.1") at the first call, which returns
// the copy the host process has already loaded (PyTorch ships one with that SONAME) and keeps the library loadable on machines without a
// GPU runtime (the CPU-side build / ABI checks).
#include <dlfcn.h>
#include <stdio.h>
#include <rccl/rccl.h>

#include "common.hpp"

namespace snerf {

struct RcclApi {
  ncclResult_t (*GetUniqueId)(ncclUniqueId*);
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int);
  ncclResult_t (*CommDestroy)(ncclComm_t);
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t);
  const char* (*GetErrorString)(ncclResult_t);
  bool ok;
  char why[256];  // dlopen / dlsym diagnosis, captured once (dlerror() clears itself when read)
};

static RcclApi* rccl() {
  static RcclApi api = [] {
    RcclApi a = {};
    void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) {
      const char* e = dlerror();
      snprintf(a.why, sizeof(a.why), "%s", e ? e : "dlopen failed");
      return a;
    }
    a.GetUniqueId = reinterpret_cast<decltype(a.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
    a.CommInitRank = reinterpret_cast<decltype(a.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
    a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
    a.AllReduce = reinterpret_cast<decltype(a.AllReduce)>(dlsym(h, "ncclAllReduce"));
    a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
    a.ok = a.GetUniqueId && a.CommInitRank && a.CommDestroy && a.AllReduce && a.GetErrorString;
    if (!a.ok) snprintf(a.why, sizeof(a.why), "librccl is missing one of ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclAllReduce / ncclGetErrorString");
    return a;
  }();
  return &api;
}

static int check_rccl(RcclApi* r, ncclResult_t e, const char* what) {
  if (e == ncclSuccess) return 0;
  set_error("%s: %s", what, r->GetErrorString(e));
  return 1000 + (int)e;
}

struct Comm {
  ncclComm_t comm;
  int world, rank;
};

}  // namespace snerf

using namespace snerf;

#define SNERF_NEED_RCCL(r)                                                                              \
  RcclApi* r = rccl();                                                                                  \
  SNERF_REQUIRE(r->ok, "RCCL (librccl.so.1) could not be loaded: %s", r->why)

extern "C" int snerf_comm_unique_id(void* id128) {
  SNERF_REQUIRE(id128, "comm_unique_id: null buffer");
  SNERF_NEED_RCCL(r);
  static_assert(sizeof(ncclUniqueId) == SNERF_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
  return check_rccl(r, r->GetUniqueId(reinterpret_cast<ncclUniqueId*>(id128)), "ncclGetUniqueId");
}

extern "C" int snerf_comm_create(int32_t world, int32_t rank, const void* id128, void** comm_out) {
  SNERF_REQUIRE(world >= 1 && rank >= 0 && rank < world && id128 && comm_out, "comm_create: world=%d rank=%d", world, rank);
  SNERF_NEED_RCCL(r);
  ncclUniqueId id;
  __builtin_memcpy(&id, id128, sizeof(id));
  Comm* c = new Comm{nullptr, world, rank};
  int rc = check_rccl(r, r->CommInitRank(&c->comm, world, id, rank), "ncclCommInitRank");
  if (rc) { delete c; return rc; }
  *comm_out = c;
  return 0;
}

extern "C" int snerf_comm_destroy(void* comm) {
  if (!comm) return 0;
  SNERF_NEED_RCCL(r);
  Comm* c = static_cast<Comm*>(comm);
  int rc = check_rccl(r, r->CommDestroy(c->comm), "ncclCommDestroy");
  delete c;
  return rc;
}

extern "C" int snerf_allreduce_grads(void* comm, float* grads, int64_t n, snerf_stream_t stream) {
  SNERF_REQUIRE(comm && n >= 0, "allreduce_grads: null communicator or n=%lld", (long long)n);
  if (n == 0) return 0;
  SNERF_REQUIRE(grads, "allreduce_grads: null buffer");
  SNERF_NEED_RCCL(r);
  Comm* c = static_cast<Comm*>(comm);
  return check_rccl(r, r->AllReduce(grads, grads, (size_t)n, ncclFloat32, ncclSum, c->comm, (hipStream_t)stream), "ncclAllReduce");
}
