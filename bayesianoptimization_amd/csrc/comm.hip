// Multi-GPU arg-best exchange over RCCL (xGMI), one process per GPU.
// The candidate shards are independent; the only exchange is an all-gather of each rank's
// (value, global index) records — 16 bytes each — followed by an identical host-side merge on every
// rank.  librccl is dlopen'ed on first use so that single-GPU users never load it.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include "gpbo_internal.h"

namespace gpbo {

struct RcclApi {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

static RcclApi g_rccl;

static int load_rccl(gpbo_ctx* ctx) {
  if (g_rccl.handle) return GPBO_OK;
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  void* h = nullptr;
  for (const char* n : names) {
    h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (h) break;
  }
  if (!h) GPBO_FAIL(ctx, GPBO_ERR_COMM, std::string("dlopen(librccl) failed: ") + dlerror());
  g_rccl.GetUniqueId = (decltype(g_rccl.GetUniqueId))dlsym(h, "ncclGetUniqueId");
  g_rccl.CommInitRank = (decltype(g_rccl.CommInitRank))dlsym(h, "ncclCommInitRank");
  g_rccl.AllGather = (decltype(g_rccl.AllGather))dlsym(h, "ncclAllGather");
  g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))dlsym(h, "ncclCommDestroy");
  g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))dlsym(h, "ncclGetErrorString");
  if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.AllGather || !g_rccl.CommDestroy)
    GPBO_FAIL(ctx, GPBO_ERR_COMM, "librccl is missing a required symbol");
  g_rccl.handle = h;
  return GPBO_OK;
}

#define GPBO_NCCL(ctx, expr)                                                               \
  do {                                                                                     \
    ncclResult_t _r = (expr);                                                              \
    if (_r != ncclSuccess) {                                                               \
      std::string _m = std::string(#expr) + " failed: " +                                  \
                       (g_rccl.GetErrorString ? g_rccl.GetErrorString(_r) : "?");          \
      GPBO_FAIL(ctx, GPBO_ERR_COMM, _m);                                                   \
    }                                                                                      \
  } while (0)

}  // namespace gpbo

using namespace gpbo;

extern "C" int gpbo_comm_unique_id(char id[128]) {
  int rc = load_rccl(nullptr);
  if (rc) return rc;
  ncclUniqueId uid;
  GPBO_NCCL((gpbo_ctx*)nullptr, g_rccl.GetUniqueId(&uid));
  static_assert(sizeof(uid) == 128, "ncclUniqueId is 128 bytes");
  memcpy(id, &uid, 128);
  return GPBO_OK;
}

extern "C" int gpbo_comm_init(gpbo_ctx* ctx, const char id[128], int world_size, int rank) {
  if (!ctx || !id || world_size < 1 || rank < 0 || rank >= world_size)
    GPBO_FAIL(ctx, GPBO_ERR_INVALID, "comm_init: bad arguments");
  int rc = load_rccl(ctx);
  if (rc) return rc;
  GPBO_HIP(ctx, hipSetDevice(ctx->device));
  ncclUniqueId uid;
  memcpy(&uid, id, 128);
  ncclComm_t comm = nullptr;
  GPBO_NCCL(ctx, g_rccl.CommInitRank(&comm, world_size, uid, rank));
  ctx->comm = comm;
  ctx->world = world_size;
  ctx->rank = rank;
  return GPBO_OK;
}

extern "C" int gpbo_comm_allgather_best(gpbo_ctx* ctx, const double* vals, const int64_t* idxs,
                                        int n_records, double* all_vals, int64_t* all_idxs) {
  if (!ctx || !ctx->comm) GPBO_FAIL(ctx, GPBO_ERR_STATE, "comm_allgather_best: communicator not initialised");
  if (n_records < 1 || n_records > 4096) GPBO_FAIL(ctx, GPBO_ERR_INVALID, "comm_allgather_best: bad n_records");
  GPBO_HIP(ctx, hipSetDevice(ctx->device));
  const size_t rec = 16, send_bytes = rec * n_records, recv_bytes = send_bytes * ctx->world;
  int rc;
  {
    char* p = (char*)ctx->comm_buf;
    int64_t cap = ctx->cap_comm_buf;
    if ((rc = ensure(ctx, &p, &cap, (int64_t)(send_bytes + recv_bytes)))) return rc;
    ctx->comm_buf = p;
    ctx->cap_comm_buf = cap;
  }
  std::string host(send_bytes + recv_bytes, '\0');
  for (int t = 0; t < n_records; ++t) {
    memcpy(&host[t * rec], &vals[t], 8);
    memcpy(&host[t * rec + 8], &idxs[t], 8);
  }
  char* dsend = (char*)ctx->comm_buf;
  char* drecv = dsend + send_bytes;
  GPBO_HIP(ctx, hipMemcpyAsync(dsend, host.data(), send_bytes, hipMemcpyHostToDevice, ctx->stream));
  GPBO_NCCL(ctx, g_rccl.AllGather(dsend, drecv, send_bytes, ncclChar, (ncclComm_t)ctx->comm, ctx->stream));
  GPBO_HIP(ctx, hipMemcpyAsync(&host[send_bytes], drecv, recv_bytes, hipMemcpyDeviceToHost, ctx->stream));
  GPBO_HIP(ctx, hipStreamSynchronize(ctx->stream));
  for (int t = 0; t < n_records * ctx->world; ++t) {
    memcpy(&all_vals[t], &host[send_bytes + t * rec], 8);
    memcpy(&all_idxs[t], &host[send_bytes + t * rec + 8], 8);
  }
  return GPBO_OK;
}

extern "C" int gpbo_comm_destroy(gpbo_ctx* ctx) {
  if (!ctx) return GPBO_ERR_INVALID;
  if (ctx->comm && g_rccl.CommDestroy) g_rccl.CommDestroy((ncclComm_t)ctx->comm);
  ctx->comm = nullptr;
  ctx->world = 1;
  ctx->rank = 0;
  return GPBO_OK;
}
