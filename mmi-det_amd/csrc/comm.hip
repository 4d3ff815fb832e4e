// Data-parallel communication of the training step: the gradient all-reduce and the initial parameter broadcast on RCCL,
// enqueued on a HIP stream the caller chooses (SURVEY.md §8b "comm: mmi_comm_init, mmi_allreduce_bucket"; replaces what
// DistributedDataParallel does for train.py:683-686 of the reference).
//
// RCCL is bound at run time (dlopen/dlsym) and the copy ALREADY in the process is preferred (PyTorch-ROCm ships one with the
// soname librccl.so.1): one RCCL per process, no link-time dependency, and the library still loads -- and every other entry
// point still works -- on a host without RCCL (the CPU test tier).  One communicator per process (one process per GPU).
// Unlike torch.distributed's ProcessGroupNCCL there is no watchdog thread polling events, so the collectives may be captured
// into a hipGraph together with the kernels around them, and nothing binds the communicator to the device at init.
#include <dlfcn.h>
#include <string.h>

#include "common.h"

namespace {

typedef struct { char internal[128]; } UniqueId;           // ncclUniqueId (NCCL_UNIQUE_ID_BYTES = 128)
typedef void* Comm;                                        // ncclComm_t
constexpr int kFloat32 = 7, kUint8 = 1, kSum = 0, kAvg = 4;  // ncclFloat32, ncclUint8, ncclSum, ncclAvg (rccl.h)

struct Api {
  void* handle = nullptr;
  int (*GetUniqueId)(UniqueId*) = nullptr;
  int (*CommInitRank)(Comm*, int, UniqueId, int) = nullptr;
  int (*CommDestroy)(Comm) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, Comm, hipStream_t) = nullptr;
  int (*Broadcast)(const void*, void*, size_t, int, int, Comm, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  int (*GetVersion)(int*) = nullptr;
} g_api;
Comm g_comm = nullptr;
int g_rank = -1, g_world = 0;

int load_api() {
  if (g_api.handle != nullptr) return MMI_OK;
  void* h = nullptr;
  for (const char* name : {"librccl.so.1", "librccl.so"}) {   // the copy torch has already loaded, if any
    h = dlopen(name, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
    if (h) break;
  }
  for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
    if (h) break;
    h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
  }
  if (h == nullptr) {
    mmi_set_error("mmi_comm: RCCL (librccl.so.1) is not available: %s", dlerror());
    return MMI_ERR_LAUNCH;
  }
#define SYM(field, sym)                                              \
  g_api.field = (decltype(g_api.field))dlsym(h, sym);                \
  if (g_api.field == nullptr) {                                      \
    mmi_set_error("mmi_comm: symbol %s missing from RCCL", sym);     \
    return MMI_ERR_LAUNCH;                                           \
  }
  SYM(GetUniqueId, "ncclGetUniqueId")
  SYM(CommInitRank, "ncclCommInitRank")
  SYM(CommDestroy, "ncclCommDestroy")
  SYM(AllReduce, "ncclAllReduce")
  SYM(Broadcast, "ncclBroadcast")
  SYM(GetErrorString, "ncclGetErrorString")
  SYM(GetVersion, "ncclGetVersion")
#undef SYM
  g_api.handle = h;
  return MMI_OK;
}

int check(int rc, const char* what) {
  if (rc == 0) return MMI_OK;
  mmi_set_error("%s: RCCL error %d (%s)", what, rc, g_api.GetErrorString ? g_api.GetErrorString(rc) : "?");
  return MMI_ERR_LAUNCH;
}

}  // namespace

extern "C" int mmi_comm_available(void) { return load_api() == MMI_OK ? 1 : 0; }

// rank 0 creates the 128-byte rendezvous id (HOST memory); the caller ships it to the other ranks by whatever side channel
// it has (the Python side uses the torch.distributed store / a gloo broadcast; MPI, a file or a socket work as well).
extern "C" int mmi_comm_unique_id(void* id128_host) {
  MMI_CHECK_ARG(id128_host != nullptr, "mmi_comm_unique_id: null pointer");
  if (int e = load_api()) return e;
  return check(g_api.GetUniqueId((UniqueId*)id128_host), "mmi_comm_unique_id");
}

// Collective over all ranks: every rank calls it with the same id, its own rank, after hipSetDevice() to its GPU.
extern "C" int mmi_comm_init(int rank, int world, const void* id128_host) {
  MMI_CHECK_ARG(id128_host != nullptr && world > 0 && rank >= 0 && rank < world, "mmi_comm_init: bad arguments");
  MMI_CHECK_ARG(g_comm == nullptr, "mmi_comm_init: a communicator already exists (one per process; mmi_comm_destroy first)");
  if (int e = load_api()) return e;
  UniqueId id;
  memcpy(&id, id128_host, sizeof(id));
  if (int e = check(g_api.CommInitRank(&g_comm, world, id, rank), "mmi_comm_init")) {
    g_comm = nullptr;
    return e;
  }
  g_rank = rank;
  g_world = world;
  return MMI_OK;
}

extern "C" int mmi_comm_world(void) { return g_comm ? g_world : 0; }
extern "C" int mmi_comm_rank(void) { return g_comm ? g_rank : -1; }

// In-place all-reduce of `count` fp32 elements of a flat gradient bucket, enqueued on `stream`: average = 1 gives every rank
// the mean over ranks (what DDP leaves in .grad), 0 the sum.
extern "C" int mmi_allreduce_bucket(float* bucket, int64_t count, int average, void* stream) {
  MMI_CHECK_ARG(bucket != nullptr && count > 0, "mmi_allreduce_bucket: bad arguments");
  MMI_CHECK_ARG(g_comm != nullptr, "mmi_allreduce_bucket: call mmi_comm_init first");
  return check(g_api.AllReduce(bucket, bucket, (size_t)count, kFloat32, average ? kAvg : kSum, g_comm, (hipStream_t)stream),
               "mmi_allreduce_bucket");
}

// In-place broadcast of `bytes` bytes from rank `root` (initial weights and buffers, once).
extern "C" int mmi_broadcast_bytes(void* buf, int64_t bytes, int root, void* stream) {
  MMI_CHECK_ARG(buf != nullptr && bytes > 0 && root >= 0, "mmi_broadcast_bytes: bad arguments");
  MMI_CHECK_ARG(g_comm != nullptr && root < g_world, "mmi_broadcast_bytes: call mmi_comm_init first");
  return check(g_api.Broadcast(buf, buf, (size_t)bytes, kUint8, root, g_comm, (hipStream_t)stream), "mmi_broadcast_bytes");
}

extern "C" int mmi_comm_destroy(void) {
  if (g_comm == nullptr) return MMI_OK;
  const int rc = g_api.CommDestroy(g_comm);
  g_comm = nullptr;
  g_rank = -1;
  g_world = 0;
  return check(rc, "mmi_comm_destroy");
}
