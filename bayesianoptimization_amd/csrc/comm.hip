// Multi-GPU arg-best exchange over RCCL (xGMI) — SURVEY.md §8e / §8b-B3.
//
// The candidate shards are independent (sklearn _gpr.py:443-494 is row-wise; bayes_opt/acquisition.py:312-317 needs
// only a global argmin and the k best), so the ONLY exchange is an all-gather of each shard's 1 + k (value, global
// index) records — 16 bytes each — followed by an identical host merge on every rank.  The records are produced on
// the device (pack_records_kernel) and go straight into ncclAllGather on the context's stream; nothing bounces
// through the host before the collective.
//
// Two ways to own the GPUs, one exchange:
//   * one process per GPU  : gpbo_comm_unique_id / gpbo_comm_init (ncclCommInitRank), then gpbo_comm_acq_argbest;
//   * one process, G GPUs  : gpbo_group_create (ncclCommInitAll, one context + one host thread per device), then
//                            gpbo_group_* — this is what sits behind BayesianOptimization.suggest(), which is one
//                            Python process (bayes_opt/bayesian_optimization.py:323-333).
// librccl is dlopen'ed on first use so that single-GPU users never load it.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <functional>
#include <limits>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "gpbo_internal.h"

namespace gpbo {

struct RcclApi {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;      // optional: the failure path works without it, it only cannot unblock a peer
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;   // optional: only reported (gpbo_device_info)
};

static RcclApi g_rccl;
static std::mutex g_rccl_mu;

static int load_rccl(gpbo_ctx* ctx) {
  std::lock_guard<std::mutex> lk(g_rccl_mu);
  if (g_rccl.handle) return GPBO_OK;
  // a copy that is already in the process (e.g. pulled in by another library) wins: two RCCL instances in one
  // process must never talk to the same GPUs
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  void* h = nullptr;
  for (const char* n : names) {
    h = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
    if (h) break;
  }
  if (!h)
    for (const char* n : names) {
      h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
      if (h) break;
    }
  if (!h) GPBO_FAIL(ctx, GPBO_ERR_COMM, std::string("dlopen(librccl) failed: ") + dlerror());
  g_rccl.GetUniqueId = (decltype(g_rccl.GetUniqueId))dlsym(h, "ncclGetUniqueId");
  g_rccl.CommInitRank = (decltype(g_rccl.CommInitRank))dlsym(h, "ncclCommInitRank");
  g_rccl.CommInitAll = (decltype(g_rccl.CommInitAll))dlsym(h, "ncclCommInitAll");
  g_rccl.AllGather = (decltype(g_rccl.AllGather))dlsym(h, "ncclAllGather");
  g_rccl.AllReduce = (decltype(g_rccl.AllReduce))dlsym(h, "ncclAllReduce");
  g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))dlsym(h, "ncclCommDestroy");
  g_rccl.CommAbort = (decltype(g_rccl.CommAbort))dlsym(h, "ncclCommAbort");
  g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))dlsym(h, "ncclGetErrorString");
  g_rccl.CommCount = (decltype(g_rccl.CommCount))dlsym(h, "ncclCommCount");
  if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.CommInitAll || !g_rccl.AllGather || !g_rccl.AllReduce ||
      !g_rccl.CommDestroy)
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

// device [send | recv | 8-byte word] and pinned host [recv | word] buffers for n_records per rank
static int comm_buffers(gpbo_ctx* ctx, int n_records, char** dsend, char** drecv, char** hrecv) {
  const size_t send_bytes = sizeof(BestRecord) * (size_t)n_records, recv_bytes = send_bytes * (size_t)ctx->world;
  int rc;
  {
    char* p = (char*)ctx->comm_buf;
    int64_t cap = ctx->cap_comm_buf;
    if ((rc = ensure(ctx, &p, &cap, (int64_t)(send_bytes + recv_bytes + 64)))) return rc;
    ctx->comm_buf = p;
    ctx->cap_comm_buf = cap;
  }
  if ((int64_t)(recv_bytes + 64) > ctx->cap_comm_host) {
    if (ctx->comm_host) GPBO_HIP(ctx, hipHostFree(ctx->comm_host));
    ctx->comm_host = nullptr;
    ctx->cap_comm_host = 0;
    GPBO_HIP(ctx, hipHostMalloc(&ctx->comm_host, recv_bytes + 64, hipHostMallocDefault));
    ctx->cap_comm_host = (int64_t)(recv_bytes + 64);
  }
  *dsend = (char*)ctx->comm_buf;
  *drecv = *dsend + send_bytes;
  *hrecv = (char*)ctx->comm_host;
  return GPBO_OK;
}

// ---- failure path -------------------------------------------------------------------------------------------------
// A collective only completes if every rank enters it.  Two rules keep one rank's trouble from becoming everybody's hang:
//  (1) a rank whose LOCAL work failed still enters the all-gather, with a poisoned arg-best record (index INT64_MIN):
//      every rank finishes the exchange, sees the poison and returns GPBO_ERR_COMM (the failing rank: its own code);
//  (2) nobody waits for a collective without a deadline (GPBO_COMM_TIMEOUT_S, default 120): when it passes, the rank
//      aborts its communicator (ncclCommAbort: the stuck kernel is released) and returns GPBO_ERR_COMM; the communicator
//      is gone after that and every later collective call says so.
constexpr int64_t POISON_INDEX = std::numeric_limits<int64_t>::min();

static double env_seconds(const char* name, double dflt) {
  const char* e = getenv(name);
  if (!e) return dflt;
  const double v = atof(e);
  return v > 0.0 ? v : dflt;
}

// The group's main thread (deadline of a job) and the context's own worker (deadline of a collective) may both decide to
// abort: whoever exchanges the pointer out owns the one ncclCommAbort call; the other finds NULL and does nothing.
static void abort_comm(gpbo_ctx* ctx) {
  if (!ctx) return;
  void* c = __atomic_exchange_n(&ctx->comm, (void*)nullptr, __ATOMIC_ACQ_REL);   // never used again (not destroyed: the abort releases it)
  if (!c) return;
  __atomic_store_n(&ctx->comm_lost, true, __ATOMIC_RELEASE);
  if (g_rccl.CommAbort) (void)g_rccl.CommAbort((ncclComm_t)c);
}
static void* comm_of(gpbo_ctx* ctx) { return __atomic_load_n(&ctx->comm, __ATOMIC_ACQUIRE); }

// How many ranks RCCL itself says the context's communicator has (ncclCommCount): 0 = no communicator, -1 = not answered.
// Reported by gpbo_device_info so that a multi-GPU bench line says what RCCL saw, not what the launcher asked for.
int comm_nranks(gpbo_ctx* ctx) {
  void* c = comm_of(ctx);
  if (!c) return 0;
  int n = -1;
  if (!g_rccl.CommCount || g_rccl.CommCount((ncclComm_t)c, &n) != ncclSuccess) return -1;
  return n;
}

// hipStreamSynchronize with a deadline; on expiry the communicator is aborted
static int wait_collective(gpbo_ctx* ctx, const char* what) {
  const double limit = env_seconds("GPBO_COMM_TIMEOUT_S", 120.0);
  const auto t0 = std::chrono::steady_clock::now();
  for (unsigned spin = 0;; ++spin) {
    const hipError_t e = hipStreamQuery(ctx->stream);
    if (e == hipSuccess) return GPBO_OK;
    if (e != hipErrorNotReady) {
      abort_comm(ctx);
      GPBO_HIP(ctx, e);
    }
    const double waited = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (waited > limit) {
      abort_comm(ctx);
      GPBO_FAIL(ctx, GPBO_ERR_COMM, std::string(what) + ": the collective did not complete within " + std::to_string((int)limit) +
                                        " s (GPBO_COMM_TIMEOUT_S) — a peer never entered it; communicator aborted");
    }
    if (spin < 2000) std::this_thread::yield();
    else std::this_thread::sleep_for(std::chrono::microseconds(50));
  }
}

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
  ctx->comm_lost = false;
  ctx->world = world_size;
  ctx->rank = rank;
  return GPBO_OK;
}

extern "C" int gpbo_comm_allgather_best(gpbo_ctx* ctx, const double* vals, const int64_t* idxs,
                                        int n_records, double* all_vals, int64_t* all_idxs) {
  if (!ctx || !comm_of(ctx)) GPBO_FAIL(ctx, GPBO_ERR_STATE, "comm_allgather_best: communicator not initialised");
  void* const comm = comm_of(ctx);
  if (n_records < 1 || n_records > 4096) GPBO_FAIL(ctx, GPBO_ERR_INVALID, "comm_allgather_best: bad n_records");
  if (!vals || !idxs || !all_vals || !all_idxs) GPBO_FAIL(ctx, GPBO_ERR_INVALID, "comm_allgather_best: NULL argument");
  GPBO_HIP(ctx, hipSetDevice(ctx->device));
  char *dsend, *drecv, *hrecv;
  int rc = comm_buffers(ctx, n_records, &dsend, &drecv, &hrecv);
  if (rc) return rc;
  const size_t send_bytes = sizeof(BestRecord) * (size_t)n_records, recv_bytes = send_bytes * (size_t)ctx->world;
  std::vector<BestRecord> host((size_t)n_records);
  for (int t = 0; t < n_records; ++t) host[t] = BestRecord{vals[t], idxs[t]};
  GPBO_HIP(ctx, hipMemcpyAsync(dsend, host.data(), send_bytes, hipMemcpyHostToDevice, ctx->stream));
  GPBO_NCCL(ctx, g_rccl.AllGather(dsend, drecv, send_bytes, ncclChar, (ncclComm_t)comm, ctx->stream));
  GPBO_HIP(ctx, hipMemcpyAsync(hrecv, drecv, recv_bytes, hipMemcpyDeviceToHost, ctx->stream));
  if ((rc = wait_collective(ctx, "comm_allgather_best"))) return rc;
  const BestRecord* all = (const BestRecord*)hrecv;
  for (int t = 0; t < n_records * ctx->world; ++t) { all_vals[t] = all[t].v; all_idxs[t] = all[t].i; }
  return GPBO_OK;
}

extern "C" int gpbo_comm_acq_argbest(gpbo_ctx* ctx, int acq, double acq_param, double y_max, int n_constraints,
                                     const double* lb, const double* ub, int k_seeds, int64_t index_offset,
                                     int64_t* best_idx, double* best_val, int64_t* seed_idx, double* seed_val,
                                     double* ys_out) {
  AcqArgs a{};
  int rc = build_acq_args(ctx, "comm_acq_argbest", acq, acq_param, y_max, n_constraints, lb, ub, k_seeds, best_idx,
                          best_val, seed_idx, seed_val, &a);
  if (rc) return rc;
  if (ctx->comm_lost) GPBO_FAIL(ctx, GPBO_ERR_COMM, "comm_acq_argbest: the communicator was aborted after a failed collective");
  void* const comm = comm_of(ctx);     // read once: the group's deadline may abort (and clear) it from another thread
  if (ctx->world > 1 && !comm) GPBO_FAIL(ctx, GPBO_ERR_STATE, "comm_acq_argbest: communicator not initialised");
  char *dsend, *drecv, *hrecv;
  const int n_records = 1 + k_seeds;
  if ((rc = comm_buffers(ctx, n_records, &dsend, &drecv, &hrecv))) {
    abort_comm(ctx);       // no buffers to enter the exchange with: the peers' deadline ends their wait
    return rc;
  }
  const size_t send_bytes = sizeof(BestRecord) * (size_t)n_records, recv_bytes = send_bytes * (size_t)ctx->world;
  ev_begin(ctx, T_ACQ);
  int rc_local = launch_acq_records(ctx, a, ctx->M, k_seeds, index_offset, (BestRecord*)dsend);
#ifdef GPBO_DEBUG
  if (ctx->debug_fail_next_acq && rc_local == GPBO_OK) {     // gpbo_debug_fail_next_acq: this rank pretends its local pass failed
    ctx->debug_fail_next_acq = false;
    ctx->err = "injected local failure (gpbo_debug_fail_next_acq)";
    rc_local = GPBO_ERR_HIP;
  }
#endif
  const std::string local_err = ctx->err;
  if (rc_local != GPBO_OK && comm) {
    // rule (1): enter the exchange anyway, with a poisoned record
    std::vector<BestRecord> poison((size_t)n_records, BestRecord{std::numeric_limits<double>::quiet_NaN(), -1});
    poison[0].i = POISON_INDEX;
    if (hipMemcpyAsync(dsend, poison.data(), send_bytes, hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
        hipStreamSynchronize(ctx->stream) != hipSuccess) {     // the staging vector dies with this scope
      (void)hipGetLastError();
      ev_end(ctx, T_ACQ);
      abort_comm(ctx);
      ctx->err = local_err;
      return rc_local;
    }
  } else if (rc_local != GPBO_OK) {
    ev_end(ctx, T_ACQ);
    return rc_local;
  }
  const char* gathered = dsend;    // a single shard is its own union
  if (comm) {
    const ncclResult_t nr = g_rccl.AllGather(dsend, drecv, send_bytes, ncclChar, (ncclComm_t)comm, ctx->stream);
    if (nr != ncclSuccess) {
      ev_end(ctx, T_ACQ);
      abort_comm(ctx);
      GPBO_FAIL(ctx, GPBO_ERR_COMM, std::string("ncclAllGather failed: ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(nr) : "?"));
    }
    gathered = drecv;
  }
  ev_end(ctx, T_ACQ);
  GPBO_HIP(ctx, hipMemcpyAsync(hrecv, gathered, comm ? recv_bytes : send_bytes, hipMemcpyDeviceToHost, ctx->stream));
  if (ys_out && rc_local == GPBO_OK)
    GPBO_HIP(ctx, hipMemcpyAsync(ys_out, ctx->ys, (size_t)ctx->M * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  if (comm) {
    if ((rc = wait_collective(ctx, "comm_acq_argbest"))) return rc;
  } else {
    GPBO_HIP(ctx, hipStreamSynchronize(ctx->stream));
  }
  const int world = comm ? ctx->world : 1;
  const BestRecord* all = (const BestRecord*)hrecv;
  for (int r = 0; r < world; ++r)
    if (all[(size_t)r * n_records].i == POISON_INDEX) {
      if (rc_local != GPBO_OK) { ctx->err = local_err; set_global_error(local_err); return rc_local; }
      // the exchange itself completed and the communicator is intact: a distinct code, so that callers (and
      // gpbo_group::run) can tell "a peer's step failed" from "the communicator is gone"
      GPBO_FAIL(ctx, GPBO_ERR_PEER, "comm_acq_argbest: rank " + std::to_string(r) + " reported a local failure; no result for this step");
    }
  merge_records(all, world, k_seeds, best_idx, best_val, seed_idx, seed_val);
  return GPBO_OK;
}

#ifdef GPBO_DEBUG
extern "C" int gpbo_debug_fail_next_acq(gpbo_ctx* ctx) {
  if (!ctx) return GPBO_ERR_INVALID;
  ctx->debug_fail_next_acq = true;
  return GPBO_OK;
}
#endif

extern "C" int gpbo_comm_allreduce_max(gpbo_ctx* ctx, double* value) {
  if (!ctx || !value) return GPBO_ERR_INVALID;
  GPBO_HIP(ctx, hipSetDevice(ctx->device));
  GPBO_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (ctx->comm_lost) GPBO_FAIL(ctx, GPBO_ERR_COMM, "comm_allreduce_max: the communicator was aborted after a failed collective");
  void* const comm = comm_of(ctx);
  if (!comm) return GPBO_OK;    // one rank: the maximum is the value itself
  char *dsend, *drecv, *hrecv;
  int rc = comm_buffers(ctx, 1, &dsend, &drecv, &hrecv);
  if (rc) return rc;
  double* dword = (double*)(drecv + sizeof(BestRecord) * (size_t)ctx->world);
  double* hword = (double*)(hrecv + sizeof(BestRecord) * (size_t)ctx->world);
  *hword = *value;
  GPBO_HIP(ctx, hipMemcpyAsync(dword, hword, sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  GPBO_NCCL(ctx, g_rccl.AllReduce(dword, dword, 1, ncclDouble, ncclMax, (ncclComm_t)comm, ctx->stream));
  GPBO_HIP(ctx, hipMemcpyAsync(hword, dword, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  if ((rc = wait_collective(ctx, "comm_allreduce_max"))) return rc;
  *value = *hword;
  return GPBO_OK;
}

extern "C" int gpbo_comm_destroy(gpbo_ctx* ctx) {
  if (!ctx) return GPBO_ERR_INVALID;
  void* c = __atomic_exchange_n(&ctx->comm, (void*)nullptr, __ATOMIC_ACQ_REL);
  if (c && g_rccl.CommDestroy) g_rccl.CommDestroy((ncclComm_t)c);
  ctx->comm_lost = false;
  ctx->world = 1;
  ctx->rank = 0;
  return GPBO_OK;
}

// ================================================================================================================
// Single-process device group
// ================================================================================================================
struct gpbo_group {
  std::vector<gpbo_ctx*> ctx;
  std::vector<int> devices;
  std::vector<std::thread> workers;
  std::mutex mu;
  std::condition_variable cv_job, cv_done;
  std::function<int(int)> job;
  uint64_t generation = 0;
  int pending = 0;
  bool stop = false;
  std::vector<int> rcs;
  bool virtual_ranks = false;      // duplicate devices (tests on one GPU): records merged on the host, no RCCL
  std::string collective = "none";
  std::string err;
  int64_t M = 0;                   // candidates resident across the group
  int d = 0;
  std::vector<int64_t> row0;       // device r owns rows [row0[r], row0[r + 1])

  bool broken = false;             // a job missed its deadline: communicators aborted, workers possibly stuck, nothing runs any more
  bool no_device = false;          // self-test group without contexts (gpbo_group_debug_create)

  // Every job has a deadline (GPBO_GROUP_TIMEOUT_S, default 300): a worker that never comes back — a collective a peer
  // did not enter, a wedged device — turns into GPBO_ERR_COMM for the caller instead of a hung suggest().  The
  // communicators are aborted (which releases workers blocked inside RCCL), the group is marked broken and every later
  // call fails at once; gpbo_group_destroy then detaches whatever is still stuck.
  // Lifetime rule: a job closure captures BY VALUE — scalars, the caller's array pointers, and a shared_ptr to whatever
  // per-job state the workers write (info words, records, returned RNG state).  Every worker holds its own copy of the
  // closure while it runs, so a worker that is released long after run() gave up writes into state it keeps alive
  // itself, never into a dead stack frame; the results are copied out to the caller only when run() succeeded.
  // (The caller's own arrays stay the caller's: after GPBO_ERR_COMM from a timed-out call they must outlive the group —
  // include/gpbo.h; GroupEngine keeps them referenced.)
  int run(std::function<int(int)> f) {
    if (broken) {
      err = "device group is broken (an earlier call missed its deadline or lost its communicator); create a new group";
      set_global_error(err);
      return GPBO_ERR_COMM;
    }
    {
      std::lock_guard<std::mutex> lk(mu);
      job = std::move(f);
      pending = (int)ctx.size();
      std::fill(done_flag.begin(), done_flag.end(), 0);     // here, not in the worker: a worker that never picks the job up is late too
      ++generation;
    }
    cv_job.notify_all();
    std::unique_lock<std::mutex> lk(mu);
    double limit = env_seconds("GPBO_GROUP_TIMEOUT_S", 300.0);
    if (!no_device && !virtual_ranks) {
      // a worker's own collective deadline has to fire first (it names the collective and aborts ITS communicator from
      // its own thread); the job deadline is the backstop for everything that is not a collective
      const double comm_limit = env_seconds("GPBO_COMM_TIMEOUT_S", 120.0);
      if (limit < comm_limit + 15.0) limit = comm_limit + 15.0;
    }
    if (!cv_done.wait_for(lk, std::chrono::duration<double>(limit), [&] { return pending == 0; })) {
      broken = true;
      std::string late;
      for (size_t r = 0; r < ctx.size(); ++r)
        if (!done_flag[r]) late += (late.empty() ? "" : ", ") + std::to_string(r);
      lk.unlock();                               // abort_comm may block inside RCCL: not under the group's mutex
      for (gpbo_ctx* c : ctx) abort_comm(c);     // (serialised per context against the worker's own abort)
      lk.lock();
      // released by the abort, the workers usually come back with an error within moments
      cv_done.wait_for(lk, std::chrono::seconds(5), [&] { return pending == 0; });
      err = "device group: rank(s) " + late + " did not finish within " + std::to_string((int)limit) +
            " s (GPBO_GROUP_TIMEOUT_S); communicators aborted, the group is unusable";
      set_global_error(err);
      return GPBO_ERR_COMM;
    }
    // the root cause first: a rank that only relays "a peer failed" (GPBO_ERR_PEER) is reported when nobody says more
    int first = -1;
    for (size_t r = 0; r < ctx.size(); ++r)
      if (rcs[r] != GPBO_OK && (first < 0 || (rcs[(size_t)first] == GPBO_ERR_PEER && rcs[r] != GPBO_ERR_PEER))) first = (int)r;
    if (first >= 0) {
      const size_t r = (size_t)first;
      err = "device " + std::to_string(no_device ? (int)r : devices[r]) + " (rank " + std::to_string(r) + "): " +
            (ctx[r] ? ctx[r]->err : std::string("injected failure"));
      set_global_error(err);
      // only a LOST communicator breaks the group: a rank's local failure relayed through the (completed) exchange leaves
      // every communicator intact and the next call may succeed
      for (size_t q = 0; q < ctx.size(); ++q)
        if (rcs[q] == GPBO_ERR_COMM || (ctx[q] && ctx[q]->comm_lost)) broken = true;
      return rcs[r];
    }
    return GPBO_OK;
  }
  std::vector<char> done_flag;
};

static void group_worker(gpbo_group* g, int rank) {
  if (!g->no_device) (void)hipSetDevice(g->devices[rank]);
  uint64_t seen = 0;
  for (;;) {
    std::function<int(int)> f;
    {
      std::unique_lock<std::mutex> lk(g->mu);
      g->cv_job.wait(lk, [&] { return g->stop || g->generation != seen; });
      if (g->stop) return;
      seen = g->generation;
      f = g->job;
    }
    const int rc = f(rank);
    {
      std::lock_guard<std::mutex> lk(g->mu);
      g->rcs[rank] = rc;
      g->done_flag[rank] = 1;
      --g->pending;
    }
    g->cv_done.notify_all();
  }
}

static int group_fail(gpbo_group* g, int code, const std::string& msg) {
  if (g) g->err = msg;
  set_global_error(msg);
  return code;
}

extern "C" int gpbo_group_create(int n_dev, const int* devices, gpbo_group** out) {
  if (!out) return GPBO_ERR_INVALID;
  *out = nullptr;
  if (n_dev < 1 || n_dev > 64 || !devices) return group_fail(nullptr, GPBO_ERR_INVALID, "group_create: bad arguments");
  gpbo_group* g = new gpbo_group();
  for (int r = 0; r < n_dev; ++r)
    for (int q = 0; q < r; ++q)
      if (devices[q] == devices[r]) g->virtual_ranks = true;
  if (const char* e = getenv("GPBO_GROUP_HOST_MERGE"))
    if (e[0] == '1') g->virtual_ranks = true;
  g->devices.assign(devices, devices + n_dev);
  g->rcs.assign((size_t)n_dev, GPBO_OK);
  g->done_flag.assign((size_t)n_dev, 1);
  auto cleanup = [&](int rc) {
    for (gpbo_ctx* c : g->ctx) gpbo_destroy(c);
    delete g;
    return rc;
  };
  for (int r = 0; r < n_dev; ++r) {
    gpbo_ctx* c = nullptr;
    int rc = gpbo_create(devices[r], &c);
    if (rc) return cleanup(rc);
    g->ctx.push_back(c);
  }
  if (g->virtual_ranks) {
    g->collective = "host-merge(virtual ranks)";
  } else {
    int rc = load_rccl(nullptr);
    if (rc) return cleanup(rc);
    std::vector<ncclComm_t> comms((size_t)n_dev, nullptr);
    ncclResult_t nr = g_rccl.CommInitAll(comms.data(), n_dev, devices);
    if (nr != ncclSuccess) {
      set_global_error(std::string("ncclCommInitAll failed: ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(nr) : "?"));
      return cleanup(GPBO_ERR_COMM);
    }
    for (int r = 0; r < n_dev; ++r) { g->ctx[r]->comm = comms[r]; g->ctx[r]->world = n_dev; g->ctx[r]->rank = r; }
    g->collective = "rccl-allgather";
  }
  for (int r = 0; r < n_dev; ++r) g->workers.emplace_back(group_worker, g, r);
  *out = g;
  return GPBO_OK;
}

extern "C" int gpbo_group_destroy(gpbo_group* g) {
  if (!g) return GPBO_OK;
  {
    std::lock_guard<std::mutex> lk(g->mu);
    g->stop = true;
  }
  g->cv_job.notify_all();
  bool stuck = false;
  {
    std::lock_guard<std::mutex> lk(g->mu);
    stuck = g->broken && g->pending > 0;
  }
  if (stuck) {
    // a worker is still inside a call that never returns: it cannot be joined and its context cannot be torn down under
    // it — leave both to the process (this group is only ever destroyed on the way out of a failed run)
    for (auto& t : g->workers) t.detach();
    return GPBO_OK;
  }
  for (auto& t : g->workers) t.join();
  for (gpbo_ctx* c : g->ctx) if (c) gpbo_destroy(c);
  delete g;
  return GPBO_OK;
}

#ifdef GPBO_DEBUG
// Self-test seam (no device needed): a group of `n_ranks` workers without contexts, and a job in which rank `fail_rank`
// returns `fail_code` and rank `hang_rank` sleeps `hang_ms` before returning — what a failed / wedged device looks like
// to gpbo_group::run.  Either rank may be -1.
extern "C" int gpbo_group_debug_create(int n_ranks, gpbo_group** out) {
  if (!out || n_ranks < 1 || n_ranks > 64) return GPBO_ERR_INVALID;
  gpbo_group* g = new gpbo_group();
  g->no_device = true;
  g->virtual_ranks = true;
  g->collective = "none(self-test)";
  g->ctx.assign((size_t)n_ranks, nullptr);
  g->rcs.assign((size_t)n_ranks, GPBO_OK);
  g->done_flag.assign((size_t)n_ranks, 1);
  for (int r = 0; r < n_ranks; ++r) g->workers.emplace_back(group_worker, g, r);
  *out = g;
  return GPBO_OK;
}

extern "C" int gpbo_group_debug_run(gpbo_group* g, int fail_rank, int fail_code, int hang_rank, int hang_ms) {
  if (!g) return GPBO_ERR_INVALID;
  return g->run([=](int r) {
    if (r == hang_rank) std::this_thread::sleep_for(std::chrono::milliseconds(hang_ms));
    return r == fail_rank ? fail_code : (int)GPBO_OK;
  });
}
#endif  // GPBO_DEBUG

extern "C" int gpbo_group_size(const gpbo_group* g) { return g ? (int)g->ctx.size() : 0; }
extern "C" gpbo_ctx* gpbo_group_ctx(gpbo_group* g, int rank) {
  return (g && rank >= 0 && rank < (int)g->ctx.size()) ? g->ctx[rank] : nullptr;
}
extern "C" const char* gpbo_group_collective(const gpbo_group* g) { return g ? g->collective.c_str() : "none"; }
extern "C" const char* gpbo_group_last_error(const gpbo_group* g) { return g ? g->err.c_str() : gpbo_last_error(nullptr); }

extern "C" int gpbo_group_synchronize(gpbo_group* g) {
  if (!g) return GPBO_ERR_INVALID;
  return g->run([g](int r) { return gpbo_synchronize(g->ctx[r]); });
}

extern "C" int gpbo_group_fit(gpbo_group* g, int slot, const double* X, const double* y_norm, int64_t N, int d, int kernel,
                              const double* length_scale, int n_ls, double noise, int precision, int* info) {
  if (!g) return GPBO_ERR_INVALID;
  if (info) *info = 0;
  auto infos = std::make_shared<std::vector<int>>(g->ctx.size(), 0);      // per-job state: owned by the closure copies
  // replicated: every device factorises the same (deterministic) model — cheaper than shipping N^2 doubles
  int rc = g->run([=](int r) {
    return gpbo_fit(g->ctx[r], slot, X, y_norm, N, d, kernel, length_scale, n_ls, noise, precision, &(*infos)[r]);
  });
  if (info && rc != GPBO_ERR_COMM) *info = (*infos)[0];
  return rc;
}

extern "C" int gpbo_group_fit_append(gpbo_group* g, int slot, const double* x_new, int64_t n_new, int d, const double* y_norm,
                                     int64_t n_total, int* info) {
  if (!g) return GPBO_ERR_INVALID;
  if (info) *info = 0;
  auto infos = std::make_shared<std::vector<int>>(g->ctx.size(), 0);
  int rc = g->run([=](int r) { return gpbo_fit_append(g->ctx[r], slot, x_new, n_new, d, y_norm, n_total, &(*infos)[r]); });
  if (info && rc != GPBO_ERR_COMM) *info = (*infos)[0];
  return rc;
}

// The theta search across the group (sklearn _gpr.py:296-338: 1 + n_restarts_optimizer independent L-BFGS-B runs over the
// log-marginal likelihood; HipGPR advances them in lockstep and asks for every live run's next evaluation at once).
// gpbo_lml_batch puts the lanes of one call side by side on ONE device, where from N ~ 2048 on they queue behind each
// other's GEMMs (six lanes 13.0 ms, one 3.2 ms at N = 4096); the lanes are independent, so here lane i runs on device
// i mod G — each device evaluates its lanes with gpbo_lml_batch, i.e. every value and gradient is bitwise gpbo_lml's on any
// device.  X / y_norm non-NULL: made resident on EVERY device first (a later call may hand a lane to a device that had
// none in this one); NULL: the inputs of the previous call are reused.
extern "C" int gpbo_group_lml_batch(gpbo_group* g, int n_theta, const double* X, const double* y_norm, int64_t N, int d, int kernel,
                                    const double* length_scales, int n_ls, double noise, int eval_gradient, double* lml,
                                    double* grad, int* info, int* lane_device) {
  if (!g) return GPBO_ERR_INVALID;
  if (n_theta < 1 || n_theta > 64 || !length_scales || !lml || (eval_gradient && !grad) || (!X) != (!y_norm) || (n_ls != 1 && n_ls != d))
    return group_fail(g, GPBO_ERR_INVALID, "group_lml_batch: bad arguments");
  const int G = (int)g->ctx.size();
  struct State {
    std::vector<double> ls, lml, grad;
    std::vector<int> info;
  };
  auto st = std::make_shared<State>();          // per-job state: the workers write here, copied out on success
  st->ls.assign(length_scales, length_scales + (size_t)n_theta * n_ls);
  st->lml.assign((size_t)n_theta, 0.0);
  st->grad.assign((size_t)n_theta * n_ls, 0.0);
  st->info.assign((size_t)n_theta, 0);
  int rc = g->run([=](int r) {
    gpbo_ctx* c = g->ctx[r];
    if (X) {
      int rcu = lml_upload_inputs(c, X, y_norm, N, d);
      if (rcu) return rcu;
    }
    std::vector<int> mine;
    for (int i = r; i < n_theta; i += G) mine.push_back(i);
    for (size_t b = 0; b < mine.size(); b += GPBO_LML_BATCH_MAX) {      // (more lanes than one call takes: a second call)
      const int nb = (int)std::min<size_t>(GPBO_LML_BATCH_MAX, mine.size() - b);
      std::vector<double> ls((size_t)nb * n_ls), v((size_t)nb), gr((size_t)nb * n_ls);
      std::vector<int> inf((size_t)nb, 0);
      for (int t = 0; t < nb; ++t)
        for (int q = 0; q < n_ls; ++q) ls[(size_t)t * n_ls + q] = st->ls[(size_t)mine[b + t] * n_ls + q];
      int rcl = gpbo_lml_batch(c, nb, nullptr, nullptr, N, d, kernel, ls.data(), n_ls, noise, eval_gradient, v.data(),
                               eval_gradient ? gr.data() : nullptr, inf.data());
      if (rcl) return rcl;
      for (int t = 0; t < nb; ++t) {
        st->lml[(size_t)mine[b + t]] = v[t];
        st->info[(size_t)mine[b + t]] = inf[t];
        if (eval_gradient)
          for (int q = 0; q < n_ls; ++q) st->grad[(size_t)mine[b + t] * n_ls + q] = gr[(size_t)t * n_ls + q];
      }
    }
    return (int)GPBO_OK;
  });
  if (rc) return rc;
  for (int i = 0; i < n_theta; ++i) {
    lml[i] = st->lml[i];
    if (info) info[i] = st->info[i];
    if (lane_device) lane_device[i] = g->devices.empty() ? i % G : g->devices[(size_t)(i % G)];
    if (eval_gradient)
      for (int q = 0; q < n_ls; ++q) grad[(size_t)i * n_ls + q] = st->grad[(size_t)i * n_ls + q];
  }
  return GPBO_OK;
}

// contiguous block partition in index order: rank r owns rows [r M / G, (r + 1) M / G), so that global index =
// offset + local index keeps the reference's first-minimum tie-break
static void group_partition(gpbo_group* g, int64_t M, int d) {
  const int64_t G = (int64_t)g->ctx.size();
  g->row0.assign((size_t)G + 1, 0);
  for (int64_t r = 0; r <= G; ++r) g->row0[(size_t)r] = r * M / G;
  g->M = M;
  g->d = d;
}

extern "C" int gpbo_group_set_candidates(gpbo_group* g, const double* Xc, int64_t M, int d) {
  if (!g) return GPBO_ERR_INVALID;
  if (!Xc || d < 1 || d > GPBO_MAX_DIM || M < (int64_t)g->ctx.size())
    return group_fail(g, GPBO_ERR_INVALID, "group_set_candidates: need at least one candidate per device");
  group_partition(g, M, d);
  return g->run([=](int r) {
    const int64_t a = g->row0[r], b = g->row0[r + 1];
    return gpbo_set_candidates(g->ctx[r], Xc + a * d, b - a, d);
  });
}

extern "C" int gpbo_generate_candidate_rows_mt19937(gpbo_ctx* ctx, int64_t M, int d, int64_t row_begin, int64_t row_end,
                                                    const double* lo, const double* hi, const uint32_t* key, int pos,
                                                    uint32_t* key_out, int* pos_out);

extern "C" int gpbo_group_generate_candidates_mt19937(gpbo_group* g, int64_t M, int d, const double* lo, const double* hi,
                                                      uint32_t* key, int* pos) {
  if (!g) return GPBO_ERR_INVALID;
  if (!lo || !hi || !key || !pos || d < 1 || d > GPBO_MAX_DIM || M < (int64_t)g->ctx.size() || *pos < 0 || *pos > 624)
    return group_fail(g, GPBO_ERR_INVALID, "group_generate_candidates_mt19937: bad arguments (need >= one candidate per device)");
  group_partition(g, M, d);
  struct State { uint32_t key_in[624]; uint32_t key_new[624]; int pos_in; int pos_new; };
  auto st = std::make_shared<State>();            // per-job state (inputs copied: the workers read them too)
  memcpy(st->key_in, key, sizeof(st->key_in));
  st->pos_in = st->pos_new = *pos;
  // every device draws ITS rows of every column straight from the caller's stream (jump-ahead sub-streams): no host
  // sampling, no upload; the device that owns the last row hands the advanced state back
  int rc = g->run([=](int r) {
    const bool last = (r + 1 == (int)g->ctx.size());
    return gpbo_generate_candidate_rows_mt19937(g->ctx[r], M, d, g->row0[r], g->row0[r + 1], lo, hi, st->key_in, st->pos_in,
                                                last ? st->key_new : nullptr, last ? &st->pos_new : nullptr);
  });
  if (rc) return rc;
  memcpy(key, st->key_new, sizeof(st->key_new));
  *pos = st->pos_new;
  return GPBO_OK;
}

extern "C" int gpbo_group_shard(const gpbo_group* g, int rank, int64_t* row_begin, int64_t* row_end) {
  if (!g || rank < 0 || rank >= (int)g->ctx.size() || g->row0.empty()) return GPBO_ERR_INVALID;
  if (row_begin) *row_begin = g->row0[rank];
  if (row_end) *row_end = g->row0[rank + 1];
  return GPBO_OK;
}

extern "C" int gpbo_group_posterior(gpbo_group* g, int slot, double y_mean, double y_std, double* mu, double* sd) {
  if (!g) return GPBO_ERR_INVALID;
  if (g->M < 1) return group_fail(g, GPBO_ERR_STATE, "group_posterior: no candidates resident (call gpbo_group_set_candidates)");
  return g->run([=](int r) {
    const int64_t a = g->row0[r];
    return gpbo_posterior(g->ctx[r], slot, y_mean, y_std, mu ? mu + a : nullptr, sd ? sd + a : nullptr);
  });
}

extern "C" int gpbo_group_acq_argbest(gpbo_group* g, int acq, double acq_param, double y_max, int n_constraints,
                                      const double* lb, const double* ub, int k_seeds, int64_t* best_idx, double* best_val,
                                      int64_t* seed_idx, double* seed_val, double* ys_out) {
  if (!g) return GPBO_ERR_INVALID;
  if (g->M < 1) return group_fail(g, GPBO_ERR_STATE, "group_acq_argbest: no candidates resident");
  if (k_seeds < 0 || k_seeds > GPBO_MAX_SEEDS || !best_idx || !best_val || (k_seeds > 0 && (!seed_idx || !seed_val)))
    return group_fail(g, GPBO_ERR_INVALID, "group_acq_argbest: bad arguments");
  const int G = (int)g->ctx.size(), stride = 1 + k_seeds;
  auto recs = std::make_shared<std::vector<BestRecord>>((size_t)G * stride);     // per rank: [best, seeds...]; per-job state
  auto local = [=](int r, bool exchange) {
    int64_t bi = -1; double bv = 0.0;
    std::vector<int64_t> si((size_t)std::max(k_seeds, 1), -1);
    std::vector<double> sv((size_t)std::max(k_seeds, 1), 0.0);
    double* ys = ys_out ? ys_out + g->row0[r] : nullptr;
    int rc = exchange
      ? gpbo_comm_acq_argbest(g->ctx[r], acq, acq_param, y_max, n_constraints, lb, ub, k_seeds, g->row0[r], &bi, &bv,
                              si.data(), sv.data(), ys)
      : gpbo_acq_argbest(g->ctx[r], acq, acq_param, y_max, n_constraints, lb, ub, k_seeds, g->row0[r], &bi, &bv, si.data(),
                         sv.data(), ys);
    if (rc) return rc;
    std::vector<BestRecord>& rec = *recs;
    rec[(size_t)r * stride] = BestRecord{bv, bi};
    for (int t = 0; t < k_seeds; ++t) rec[(size_t)r * stride + 1 + t] = BestRecord{sv[t], si[t]};
    return (int)GPBO_OK;
  };
  if (g->virtual_ranks) {
    // shards of one GPU (tests): every rank's local records, merged here exactly as the ranks merge the gathered ones
    int rc = g->run([=](int r) { return local(r, false); });
    if (rc) return rc;
    merge_records(recs->data(), G, k_seeds, best_idx, best_val, seed_idx, seed_val);
    return GPBO_OK;
  }
  // every device: records packed on the device -> ncclAllGather on its stream -> the same merge; all ranks must agree
  int rc = g->run([=](int r) { return local(r, true); });
  if (rc) return rc;
  const std::vector<BestRecord>& rec = *recs;
  for (int r = 1; r < G; ++r)
    for (int t = 0; t < stride; ++t) {
      const BestRecord &a = rec[t], &b = rec[(size_t)r * stride + t];
      const bool same_v = (a.v == b.v) || (a.v != a.v && b.v != b.v);
      if (a.i != b.i || !same_v) return group_fail(g, GPBO_ERR_COMM, "group_acq_argbest: ranks disagree after the all-gather");
    }
  *best_idx = rec[0].i;
  *best_val = rec[0].v;
  for (int t = 0; t < k_seeds; ++t) { seed_idx[t] = rec[1 + t].i; seed_val[t] = rec[1 + t].v; }
  return GPBO_OK;
}

extern "C" int gpbo_group_get_candidate_rows(gpbo_group* g, const int64_t* idx, int n, double* out) {
  if (!g) return GPBO_ERR_INVALID;
  if (!idx || !out || n < 1 || n > 4096) return group_fail(g, GPBO_ERR_INVALID, "group_get_candidate_rows: bad arguments");
  if (g->M < 1) return group_fail(g, GPBO_ERR_STATE, "group_get_candidate_rows: no candidates resident");
  const int d = g->d;
  for (int t = 0; t < n; ++t)
    if (idx[t] < 0 || idx[t] >= g->M)
      for (int c = 0; c < d; ++c) out[(size_t)t * d + c] = std::numeric_limits<double>::quiet_NaN();
  const int64_t* const row0 = g->row0.data();      // (group-owned)
  return g->run([=](int r) {
    std::vector<int64_t> loc;
    std::vector<int> pos;
    for (int t = 0; t < n; ++t)
      if (idx[t] >= row0[r] && idx[t] < row0[r + 1]) { loc.push_back(idx[t] - row0[r]); pos.push_back(t); }
    if (loc.empty()) return (int)GPBO_OK;
    std::vector<double> rows(loc.size() * (size_t)d);
    int rc = gpbo_get_candidate_rows(g->ctx[r], loc.data(), (int)loc.size(), rows.data());
    if (rc) return rc;
    for (size_t q = 0; q < loc.size(); ++q) memcpy(out + (size_t)pos[q] * d, rows.data() + q * d, (size_t)d * sizeof(double));
    return (int)GPBO_OK;
  });
}
