// C-ABI entry points of libgpbo (see include/gpbo.h for the contract and the reference call each
// function replaces).  Host-side orchestration only: every numeric step is a HIP kernel launched
// on the context's stream.
#include <cmath>
#include <mutex>
#include <shared_mutex>
#include <vector>

#include "gpbo_internal.h"

namespace gpbo {

static std::mutex g_err_mu;
// hipGraph captures are taken one at a time, process-wide, and not while another context of the process allocates or uploads
// theta-search inputs (shared side): see gpbo_lml_batch.
static std::shared_mutex g_capture_mu;
#ifndef GPBO_CAPTURE_MODE
#define GPBO_CAPTURE_MODE hipStreamCaptureModeThreadLocal
#endif
#ifdef GPBO_CAPTURE_NOLOCK           // experiment builds (scripts/archive/r04_capture_stress.sh)
#define CAPTURE_LOCK
#else
#define CAPTURE_LOCK std::unique_lock<std::shared_mutex> capture_lock(g_capture_mu)
#endif
#ifdef GPBO_CAPTURE_TRACE
#define CAPTURE_TRACE(...) fprintf(stderr, __VA_ARGS__)
#else
#define CAPTURE_TRACE(...) ((void)0)
#endif
static std::string g_err;

void set_global_error(const std::string& s) {
  std::lock_guard<std::mutex> lk(g_err_mu);
  g_err = s;
}

static int check_slot(gpbo_ctx* ctx, int slot) {
  if (!ctx) return GPBO_ERR_INVALID;
  if (slot < 0 || slot >= GPBO_MAX_MODELS) GPBO_FAIL(ctx, GPBO_ERR_INVALID, "slot out of range");
  return GPBO_OK;
}

static void free_model(Model& m) {
  double** ptrs[] = {&m.ls, &m.Xs, &m.K, &m.L, &m.W, &m.Wp, &m.dinv, &m.tmp, &m.yn, &m.tvec, &m.alpha, &m.mu, &m.sd};
  for (double** p : ptrs) {
    if (*p) (void)hipFree(*p);
    *p = nullptr;
  }
  if (m.Wp32) (void)hipFree(m.Wp32);
  m = Model();
}

static int alloc_model(gpbo_ctx* ctx, Model& m, int64_t NP, int DP) {
  if (NP <= m.cap_NP && DP <= m.cap_DP && m.K) return GPBO_OK;
  double* mu = m.mu; double* sd = m.sd; int64_t cap_M = m.cap_M;   // keep posterior buffers
  m.mu = m.sd = nullptr;
  free_model(m);
  m.mu = mu; m.sd = sd; m.cap_M = cap_M;
  const size_t sq = (size_t)NP * NP * sizeof(double);
  const int64_t tmp_elems = std::max<int64_t>(NP * NP / 2, NP * (int64_t)GPBO_MAX_DIM);
  GPBO_HIP(ctx, hipMalloc((void**)&m.ls, GPBO_MAX_DIM * sizeof(double)));
  GPBO_HIP(ctx, hipMalloc((void**)&m.Xs, (size_t)NP * DP * sizeof(double)));
  GPBO_HIP(ctx, hipMalloc((void**)&m.K, sq));
  GPBO_HIP(ctx, hipMalloc((void**)&m.L, sq));
  GPBO_HIP(ctx, hipMalloc((void**)&m.W, sq));
  GPBO_HIP(ctx, hipMalloc((void**)&m.Wp, sq));
  GPBO_HIP(ctx, hipMalloc((void**)&m.dinv, (size_t)(NP / NB) * NB * NB * sizeof(double)));
  GPBO_HIP(ctx, hipMalloc((void**)&m.tmp, (size_t)tmp_elems * sizeof(double)));
  GPBO_HIP(ctx, hipMalloc((void**)&m.yn, (size_t)NP * sizeof(double)));
  GPBO_HIP(ctx, hipMalloc((void**)&m.tvec, (size_t)NP * sizeof(double)));
  GPBO_HIP(ctx, hipMalloc((void**)&m.alpha, (size_t)NP * sizeof(double)));
  m.cap_NP = NP;
  m.cap_DP = DP;
  return GPBO_OK;
}

// Blocked Cholesky of m.L (lower, in place) + the inverted 64x64 diagonal blocks (chol_kernels.hip): 128-column steps —
// one workgroup factors and inverts the diagonal block while the rest of the launch applies the previous step's update —
// inside `outer`-column panels; the matrix right of a panel gets ONE rank-`outer` update per panel, so the trailing
// matrix is read and written N / outer times instead of N / 128 times.  (Round 2's 64-column schedules — potrf_diag_kernel /
// chol_step_kernel — were retired in round 4.)
static int chol_outer_width(int64_t NP) {
  // Outer panel width by size (scripts/archive/r03_chol_probe.py, round-3 schedule): up to NP = 2048 one panel — the rank-128
  // updates of the steps reach the whole trailing matrix, whose traffic is still small, and no latency-bound
  // rank-`outer` GEMM stands between the steps (NP = 1024: 0.312 -> 0.283 ms, 2048: 0.683 -> 0.618); 1024 up to NP = 4096
  // (1.70 -> 1.66-1.68); 512 beyond (8192: 6.0 against 6.36 with 1024), where the trailing matrix no longer fits the
  // caches and every pass over it counts.
  int outer = NP <= 2048 ? (int)round_up(NP, 2 * NB) : (NP <= 4096 ? 1024 : 512);
  if (const char* e = dbg_env("GPBO_CHOL_OUTER")) outer = atoi(e);
  if (outer < 2 * NB || outer % (2 * NB)) outer = 2 * NB;
  return outer;
}

static int cholesky(gpbo_ctx* ctx, Model& m) { return launch_cholesky128(ctx, m, chol_outer_width(m.NP), nullptr); }

// W = L^-1 by recursive doubling from the inverted 64x64 diagonal blocks:
//   [[A,0],[C,B]]^-1 = [[A^-1,0],[-B^-1 C A^-1, B^-1]]  — two batched GEMMs per level.
// `zero_above`: W's strict upper block triangle is zero-filled first (a fit: the posterior pack and gpbo_get_Linv read W as a
// square).  An LML evaluation reads W by tiles at and below the diagonal only — these products, t = W y, alpha = W^T t, W^T W with
// its k-loop cut at the tiles' diagonal — and skips the 8 NP^2-byte fill (18 us + a 6 us gap at N = 4096).
static int trtri(gpbo_ctx* ctx, Model& m, bool zero_above = true) {
  int rc;
  if ((rc = launch_fill_w_diag(ctx, m, zero_above))) return rc;
  const int64_t NP = m.NP;
  for (int64_t b = NB; b < NP; b *= 2) {
    const int64_t full = NP / (2 * b);               // pairs with a full-size second block
    const int64_t rag = NP - full * 2 * b;           // leftover rows; a ragged pair exists if rag > b
    for (int part = 0; part < 2; ++part) {
      int64_t npairs, b2, r0;
      if (part == 0) { npairs = full; b2 = b; r0 = 0; }
      else { npairs = (rag > b) ? 1 : 0; b2 = rag - b; r0 = full * 2 * b; }
      if (npairs == 0) continue;
      GemmArgs t{};   // T = L21 * W11
      t.m = (int)b2; t.n = (int)b; t.k = (int)b; t.alpha = 1.0; t.beta = 0.0;
      t.A = m.L + (r0 + b) * NP + r0; t.lda = NP; t.strideA = 2 * b * NP + 2 * b;
      t.B = m.W + r0 * NP + r0; t.ldb = NP; t.strideB = 2 * b * NP + 2 * b; t.b_lower = 1;
      t.C = m.tmp; t.ldc = b; t.strideC = b * b; t.batch = (int)npairs;
      if ((rc = launch_gemm(ctx, t))) return rc;
      GemmArgs w{};   // W21 = -W22 * T
      w.m = (int)b2; w.n = (int)b; w.k = (int)b2; w.alpha = -1.0; w.beta = 0.0;
      w.A = m.W + (r0 + b) * NP + (r0 + b); w.lda = NP; w.strideA = 2 * b * NP + 2 * b; w.a_lower = 1;
      w.B = m.tmp; w.ldb = b; w.strideB = b * b;
      w.C = m.W + (r0 + b) * NP + r0; w.ldc = NP; w.strideC = 2 * b * NP + 2 * b; w.batch = (int)npairs;
      if ((rc = launch_gemm(ctx, w))) return rc;
    }
  }
  return GPBO_OK;
}

static int copy_square(gpbo_ctx* ctx, const Model& m, const double* dev, double* out, int mode) {
  // mode 0: symmetric from lower; 1: lower with zero upper
  std::vector<double> h((size_t)m.NP * m.NP);
  GPBO_HIP(ctx, hipMemcpyAsync(h.data(), dev, h.size() * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  GPBO_HIP(ctx, hipStreamSynchronize(ctx->stream));
  for (int64_t i = 0; i < m.N; ++i)
    for (int64_t j = 0; j < m.N; ++j) {
      double v;
      if (j <= i) v = h[i * m.NP + j];
      else v = mode == 0 ? h[j * m.NP + i] : 0.0;
      out[i * m.N + j] = v;
    }
  return GPBO_OK;
}

int build_acq_args(gpbo_ctx* ctx, const char* who, int acq, double acq_param, double y_max, int n_constraints,
                   const double* lb, const double* ub, int k_seeds, const void* best_idx, const void* best_val,
                   const void* seed_idx, const void* seed_val, AcqArgs* out) {
  if (!ctx) return GPBO_ERR_INVALID;
  const std::string w(who);
  if (acq < GPBO_ACQ_UCB || acq > GPBO_ACQ_POI) GPBO_FAIL(ctx, GPBO_ERR_INVALID, w + ": unknown acquisition");
  if (n_constraints < 0 || n_constraints >= GPBO_MAX_MODELS) GPBO_FAIL(ctx, GPBO_ERR_INVALID, w + ": bad n_constraints");
  if (n_constraints > 0 && (!lb || !ub)) GPBO_FAIL(ctx, GPBO_ERR_INVALID, w + ": lb/ub required");
  if (k_seeds < 0 || k_seeds > GPBO_MAX_SEEDS) GPBO_FAIL(ctx, GPBO_ERR_INVALID, w + ": k_seeds out of range [0, 64]");
  if (!best_idx || !best_val || (k_seeds > 0 && (!seed_idx || !seed_val))) GPBO_FAIL(ctx, GPBO_ERR_INVALID, w + ": NULL output");
  GPBO_HIP(ctx, hipSetDevice(ctx->device));
  AcqArgs a{};
  a.acq = acq; a.param = acq_param; a.y_max = y_max; a.n_constraints = n_constraints;
  for (int j = 0; j <= n_constraints; ++j) {
    Model& m = ctx->models[j];
    if (!m.fitted || m.M_post != ctx->M || ctx->M < 1)
      GPBO_FAIL(ctx, GPBO_ERR_STATE, w + ": run gpbo_posterior for slots 0..n_constraints first");
    a.mu[j] = m.mu;
    a.sd[j] = m.sd;
  }
  for (int j = 0; j < n_constraints; ++j) { a.lb[j] = lb[j]; a.ub[j] = ub[j]; }
  *out = a;
  return GPBO_OK;
}

}  // namespace gpbo

using namespace gpbo;

extern "C" {

int gpbo_abi_version(void) { return GPBO_ABI_VERSION; }

const char* gpbo_last_error(const gpbo_ctx* ctx) {
  if (ctx) return ctx->err.c_str();
  std::lock_guard<std::mutex> lk(g_err_mu);
  static thread_local std::string copy;
  copy = g_err;
  return copy.c_str();
}

int gpbo_device_count(int* count) {
  if (!count) return GPBO_ERR_INVALID;
  *count = 0;
  hipError_t e = hipGetDeviceCount(count);
  if (e != hipSuccess) {
    *count = 0;
    set_global_error(std::string("hipGetDeviceCount failed: ") + hipGetErrorString(e));
    return GPBO_ERR_HIP;
  }
  return GPBO_OK;
}

int gpbo_create(int device, gpbo_ctx** out) {
  if (!out) return GPBO_ERR_INVALID;
  *out = nullptr;
  int n = 0;
  int rc = gpbo_device_count(&n);
  if (rc) return rc;
  if (device < 0 || device >= n) GPBO_FAIL((gpbo_ctx*)nullptr, GPBO_ERR_INVALID, "gpbo_create: no such device");
  gpbo_ctx* ctx = new gpbo_ctx();
  ctx->device = device;
  hipError_t e = hipSetDevice(device);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipMalloc((void**)&ctx->info_dev, sizeof(int));
  if (e == hipSuccess) e = hipHostMalloc(&ctx->pinned, PIN_WINDOWS * PIN_WINDOW, hipHostMallocDefault);
  if (e == hipSuccess) ctx->pinned_aux = (char*)ctx->pinned + (PIN_WINDOWS - 1) * PIN_WINDOW;
  if (e == hipSuccess) {
    ctx->pinned_base = ctx->pinned;
    e = hipHostGetDevicePointer((void**)&ctx->pinned_base_dev, ctx->pinned_base, 0);
  }
  if (e == hipSuccess) e = hipHostMalloc(&ctx->small_pinned, SMALL_PIN_BYTES, hipHostMallocDefault);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->small_ev, hipEventDisableTiming);
  if (e == hipSuccess) {
    int* host_flag = (int*)((char*)ctx->pinned_aux + PIN_AUX_NEGVAR);
    *host_flag = 0;
    e = hipHostGetDevicePointer((void**)&ctx->negvar, host_flag, 0);
  }
  if (e != hipSuccess) {
    set_global_error(std::string("gpbo_create: ") + hipGetErrorString(e));
    delete ctx;
    return GPBO_ERR_HIP;
  }
  *out = ctx;
  return GPBO_OK;
}

int gpbo_destroy(gpbo_ctx* ctx) {
  if (!ctx) return GPBO_OK;
  (void)hipSetDevice(ctx->device);
  if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
  gpbo_comm_destroy(ctx);
  for (auto& m : ctx->models) free_model(m);
  for (auto& la : ctx->lookahead) {
    if (la.bulk) { (void)hipStreamSynchronize(la.bulk); (void)hipStreamDestroy(la.bulk); }
    for (auto ev : la.ev) (void)hipEventDestroy(ev);
  }
  ctx->lookahead.clear();
  for (auto& st : ctx->lml_stream) if (st) (void)hipStreamDestroy(st);
  for (auto& st : ctx->slot_stream) if (st) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); }
  if (ctx->small_ev) (void)hipEventDestroy(ctx->small_ev);
  if (ctx->small_pinned) (void)hipHostFree(ctx->small_pinned);
  if (ctx->polish_pinned) (void)hipHostFree(ctx->polish_pinned);
  if (ctx->fused_stage) (void)hipHostFree(ctx->fused_stage);
  if (ctx->info_slots) (void)hipFree(ctx->info_slots);
  for (auto& ln : ctx->lml_lane) if (ln.exec) (void)hipGraphExecDestroy(ln.exec);
  if (ctx->lml_slab) (void)hipFree(ctx->lml_slab);
  if (ctx->lml_X) (void)hipFree(ctx->lml_X);
  if (ctx->lml_y) (void)hipFree(ctx->lml_y);
  void* ptrs[] = {ctx->Xc, ctx->Xc_raw, ctx->Xcs, ctx->part, ctx->mu_part, ctx->ys, ctx->red, ctx->info_dev, ctx->comm_buf, ctx->kst, ctx->stage};
  if (ctx->comm_host) (void)hipHostFree(ctx->comm_host);
  if (ctx->mt_work) (void)hipFree(ctx->mt_work);
  if (ctx->mt_bits) (void)hipFree(ctx->mt_bits);
  if (ctx->mt_offset) (void)hipFree(ctx->mt_offset);
  if (ctx->mt_desc) (void)hipFree(ctx->mt_desc);

  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  if (ctx->pinned_base) (void)hipHostFree(ctx->pinned_base);
  for (auto& e : ctx->ev) {
    if (e.a) (void)hipEventDestroy(e.a);
    if (e.b) (void)hipEventDestroy(e.b);
  }
  if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
  return GPBO_OK;
}

int gpbo_synchronize(gpbo_ctx* ctx) {
  if (!ctx) return GPBO_ERR_INVALID;
  GPBO_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return GPBO_OK;
}

int gpbo_device_info(gpbo_ctx* ctx, char* buf, int buflen) {
  if (!ctx || !buf || buflen < 64) return GPBO_ERR_INVALID;
  hipDeviceProp_t p;
  GPBO_HIP(ctx, hipGetDeviceProperties(&p, ctx->device));
  char pci[32] = "?";
  if (hipDeviceGetPCIBusId(pci, (int)sizeof(pci), ctx->device) != hipSuccess) { (void)hipGetLastError(); snprintf(pci, sizeof(pci), "?"); }
  snprintf(buf, buflen,
           "{\"name\": \"%s\", \"arch\": \"%s\", \"compute_units\": %d, \"clock_mhz\": %d, "
           "\"memory_clock_mhz\": %d, \"hbm_gib\": %.1f, \"l2_mib\": %.1f, \"lds_per_block_kib\": %.0f, "
           "\"device\": %d, \"pci_bus_id\": \"%s\", \"rank\": %d, \"world\": %d, \"rccl_nranks\": %d}",
           p.name, p.gcnArchName, p.multiProcessorCount, p.clockRate / 1000, p.memoryClockRate / 1000,
           (double)p.totalGlobalMem / (1024.0 * 1024.0 * 1024.0), (double)p.l2CacheSize / (1024.0 * 1024.0),
           (double)p.sharedMemPerBlock / 1024.0, ctx->device, pci, ctx->rank, ctx->world, comm_nranks(ctx));
  return GPBO_OK;
}

// Shared by gpbo_fit and gpbo_lml: validate, upload, K, Cholesky, W = L^-1, alpha — all queued on the
// stream; the potrf info word is copied to pinned memory (valid after the next stream sync).
// Copies/fills that act on one buffer of EVERY lane (lane mode: the buffers of lane l sit l * lane_stride doubles
// behind lane 0's; host staging areas are arrays with `host_pitch` bytes per lane).
// The look-ahead side stream + events that belong to `main` (a stream about to be destroyed).
static void drop_lookahead(gpbo_ctx* ctx, hipStream_t main) {
  for (size_t i = 0; i < ctx->lookahead.size();) {
    if (ctx->lookahead[i].main != main) { ++i; continue; }
    for (auto ev : ctx->lookahead[i].ev) (void)hipEventDestroy(ev);
    if (ctx->lookahead[i].bulk) (void)hipStreamDestroy(ctx->lookahead[i].bulk);
    ctx->lookahead.erase(ctx->lookahead.begin() + (long)i);
  }
}

static hipError_t lane_memset(gpbo_ctx* ctx, void* p, size_t bytes) {
  if (ctx->lanes == 1) return hipMemsetAsync(p, 0, bytes, ctx->stream);
  return hipMemset2DAsync(p, (size_t)ctx->lane_stride * sizeof(double), 0, bytes, (size_t)ctx->lanes, ctx->stream);
}
static hipError_t lane_h2d(gpbo_ctx* ctx, void* dst, const void* src_host, size_t host_pitch, size_t bytes) {
  if (ctx->lanes == 1) return hipMemcpyAsync(dst, src_host, bytes, hipMemcpyHostToDevice, ctx->stream);
  return hipMemcpy2DAsync(dst, (size_t)ctx->lane_stride * sizeof(double), src_host, host_pitch, bytes, (size_t)ctx->lanes,
                          hipMemcpyHostToDevice, ctx->stream);
}
static hipError_t lane_d2h(gpbo_ctx* ctx, void* dst_host, size_t host_pitch, const void* src, size_t bytes) {
  if (ctx->lanes == 1) return hipMemcpyAsync(dst_host, src, bytes, hipMemcpyDeviceToHost, ctx->stream);
  return hipMemcpy2DAsync(dst_host, host_pitch, src, (size_t)ctx->lane_stride * sizeof(double), bytes, (size_t)ctx->lanes,
                          hipMemcpyDeviceToHost, ctx->stream);
}

// Host staging: single lane = window 0 of the pinned allocation (PIN_LS / PIN_INFO / PIN_LML_OUT, gpbo_internal.h);
// lane mode = one window per group with per-lane pitches: length scales at +0, info words at PIN_LANE_INFO, LML scalars
// at PIN_LANE_OUT.
constexpr size_t PIN_LS_PITCH = PIN_LS_BYTES;
constexpr size_t PIN_INFO_PITCH = 8;
constexpr size_t PIN_OUT_PITCH = PIN_LML_OUT_BYTES;
constexpr size_t PIN_LANE_WINDOW = PIN_WINDOW, PIN_LANE_INFO = 4096, PIN_LANE_OUT = 8192;   // lane-mode layout inside a group's window
static_assert(GPBO_LML_BATCH_MAX * PIN_LS_PITCH <= PIN_LANE_INFO, "lane length scales overlap the lane info words");
static_assert(PIN_LANE_INFO + GPBO_LML_BATCH_MAX * PIN_INFO_PITCH <= PIN_LANE_OUT, "lane info words overlap the lane LML scalars");
static_assert(PIN_LANE_OUT + GPBO_LML_BATCH_MAX * PIN_OUT_PITCH <= PIN_WINDOW, "lane LML scalars leave the window");

// Small problems (NP <= fused_max_np()): the whole fit — or LML evaluation — as ONE launch of one workgroup per model
// (fused_small.hip; bitwise the multi-launch sequence below).  mode 0: fit incl. the packed W; 1 / 2: LML value / value + gradient.
// src 0: raw inputs (host arrays staged through pinned memory the kernel reads directly, or device arrays X_dev / y_dev) and the
// length scales of the pinned window; src 1: the model's resident Xs / yn / ls.  The pivot word and the LML scalars land in the
// pinned words *info_host / *out_host (valid after the stream has drained) without copy nodes.
static bool use_fused(const Model& m) { return m.NP <= fused_max_np(); }
// fused_max_np() < NP <= mid_max_np(): the strip path (mid_fit.hip)
static bool use_mid(const Model& m) { return !use_fused(m) && m.NP <= mid_max_np(); }

// the address the device sees a word of the pinned window allocation at
static char* pinned_dev(gpbo_ctx* ctx, void* host) { return ctx->pinned_base_dev + ((char*)host - (char*)ctx->pinned_base); }

// X (N, d) | y (N) of a small host-side fit copied into the pinned staging window that belongs to ctx->pinned's window (0: the
// context's own stream, every such call ends with a stream synchronisation; 1 + slot: a gpbo_fit_begin in flight); the first
// kernel of the fit reads them there.
static int stage_small_inputs(gpbo_ctx* ctx, const Model& m, const double* X, const double* y_norm, const double** Xd, const double** yd) {
  if (!ctx->fused_stage) {
    GPBO_HIP(ctx, hipHostMalloc(&ctx->fused_stage, PIN_WINDOWS * FUSED_STAGE_BYTES, hipHostMallocDefault));
    GPBO_HIP(ctx, hipHostGetDevicePointer((void**)&ctx->fused_stage_dev, ctx->fused_stage, 0));
  }
  const size_t w = (size_t)((char*)ctx->pinned - (char*)ctx->pinned_base) / PIN_WINDOW;
  double* h = (double*)((char*)ctx->fused_stage + w * FUSED_STAGE_BYTES);
  memcpy(h, X, (size_t)m.N * m.d * sizeof(double));
  memcpy(h + (size_t)STAGE_NP_CAP * GPBO_MAX_DIM, y_norm, (size_t)m.N * sizeof(double));
  *Xd = (const double*)(ctx->fused_stage_dev + w * FUSED_STAGE_BYTES);
  *yd = *Xd + (size_t)STAGE_NP_CAP * GPBO_MAX_DIM;
  return GPBO_OK;
}

static int enqueue_fused(gpbo_ctx* ctx, Model& m, const double* X, const double* y_norm, const double* X_dev, const double* y_dev,
                         double noise, int mode, int n_ls, int src, int** info_host, double** out_host) {
  int rc;
  m.noise = noise;
  const double *Xd = X_dev, *yd = y_dev;
  if (src == 0 && !X_dev && (rc = stage_small_inputs(ctx, m, X, y_norm, &Xd, &yd))) return rc;
  double* scal = m.tmp;
  if (mode != 0) {
    char* p = (char*)ctx->red;
    int64_t cap = ctx->cap_red;
    if ((rc = ensure(ctx, &p, &cap, (int64_t)(8 + GPBO_MAX_DIM) * 8))) return rc;
    ctx->red = p;
    ctx->cap_red = cap;
    scal = (double*)ctx->red;
  }
  int* info_h = (int*)((char*)ctx->pinned + (ctx->lanes == 1 ? PIN_INFO : PIN_LANE_INFO));
  double* out_h = (double*)((char*)ctx->pinned + (ctx->lanes == 1 ? PIN_LML_OUT : PIN_LANE_OUT));
  ev_begin(ctx, T_FIT);
  if (!ctx->no_timing) ctx->ev[T_KMAT].used = ctx->ev[T_CHOL].used = ctx->ev[T_TRTRI].used = false;   // one kernel: no phase events
  if ((rc = launch_fused_small(ctx, m, mode, src, n_ls, Xd, yd, (const double*)pinned_dev(ctx, ctx->pinned), scal,
                               (int*)pinned_dev(ctx, info_h), (int64_t)(PIN_INFO_PITCH / sizeof(int)),
                               mode ? (double*)pinned_dev(ctx, out_h) : nullptr, (int64_t)(PIN_OUT_PITCH / sizeof(double)))))
    return rc;
  if (mode == 0) m.wp_packed = true;
  else ev_end(ctx, T_FIT);
  *info_host = info_h;
  if (out_host) *out_host = out_h;
  return GPBO_OK;
}

// The strip path behind its inputs: K (quarter tiles) -> Cholesky -> W by column strips (+ the packed W of a fit) -> alpha; the
// pivot word reaches its pinned word from the last kernel.
static int factor_mid(gpbo_ctx* ctx, Model& m, double noise, bool pack, int** info_host) {
  int rc;
  m.noise = noise;
  if (!ctx->no_timing) ctx->ev[T_KMAT].used = ctx->ev[T_CHOL].used = ctx->ev[T_TRTRI].used = false;   // no phase events on this path
  if ((rc = launch_kmat_q(ctx, m, noise, m.L))) return rc;
  if ((rc = cholesky(ctx, m))) return rc;
  if ((rc = launch_w_strip(ctx, m, pack))) return rc;
  int* info_h = (int*)((char*)ctx->pinned + (ctx->lanes == 1 ? PIN_INFO : PIN_LANE_INFO));
  if ((rc = launch_alpha_strip(ctx, m, (int*)pinned_dev(ctx, info_h), (int64_t)(PIN_INFO_PITCH / sizeof(int))))) return rc;
  if (pack && m.Wp) m.wp_packed = true;
  *info_host = info_h;
  return GPBO_OK;
}

// K, L, W = L^-1 and alpha from the device-resident scaled inputs m.Xs / targets m.yn (m.N, m.NP, m.kernel set).
static int factor_resident(gpbo_ctx* ctx, Model& m, double noise, int** info_host, bool pack = true) {
  int rc;
  if (use_fused(m)) return enqueue_fused(ctx, m, nullptr, nullptr, nullptr, nullptr, noise, 0, 0, 1, info_host, nullptr);   // (gpbo_fit_append's rebuild)
  m.noise = noise;
  GPBO_HIP(ctx, lane_memset(ctx, ctx->info_dev, sizeof(int)));
  if (use_mid(m)) return factor_mid(ctx, m, noise, pack, info_host);
  ev_begin(ctx, T_KMAT);
  if ((rc = launch_kmat(ctx, m, noise, m.L))) return rc;   // straight into the buffer the Cholesky factorises in place
  ev_end(ctx, T_KMAT);
  ev_begin(ctx, T_CHOL);
  if ((rc = cholesky(ctx, m))) return rc;
  ev_end(ctx, T_CHOL);
  int* info_h = (int*)((char*)ctx->pinned + (ctx->lanes == 1 ? PIN_INFO : PIN_LANE_INFO));   // lane mode: one word per PIN_INFO_PITCH
  GPBO_HIP(ctx, lane_d2h(ctx, info_h, PIN_INFO_PITCH, ctx->info_dev, sizeof(int)));
  // W and alpha are issued before the info check resolves (harmless on failure)
  ev_begin(ctx, T_TRTRI);
  if ((rc = trtri(ctx, m, pack))) return rc;      // (pack == false: an LML evaluation)
  ev_end(ctx, T_TRTRI);
  if ((rc = launch_trmv(ctx, m))) return rc;
  *info_host = info_h;
  return GPBO_OK;
}

// Argument checks, buffers and descriptor of a fit; the scaled length scales go to the pinned staging words.  Nothing
// is enqueued (so this part stays outside a stream capture).
static int prepare_model(gpbo_ctx* ctx, Model& m, const char* who, bool have_inputs, int64_t N, int d, int kernel,
                         const double* length_scale, int n_ls, double noise, int precision) {
  int rc;
  std::string w(who);
  if (!have_inputs || !length_scale) GPBO_FAIL(ctx, GPBO_ERR_INVALID, w + ": NULL input");
  if (N < 1 || N > (1 << 16)) GPBO_FAIL(ctx, GPBO_ERR_INVALID, w + ": N out of range [1, 65536]");
  if (d < 1 || d > GPBO_MAX_DIM) GPBO_FAIL(ctx, GPBO_ERR_UNSUPPORTED, w + ": d out of range [1, 64]");
  if (kernel != GPBO_KERNEL_RBF && kernel != GPBO_KERNEL_MATERN25)
    GPBO_FAIL(ctx, GPBO_ERR_UNSUPPORTED, w + ": kernel must be RBF or Matern(nu=2.5)");
  if (n_ls != 1 && n_ls != d) GPBO_FAIL(ctx, GPBO_ERR_INVALID, w + ": length_scale must have 1 or d entries");
  if (precision != GPBO_F64 && precision != GPBO_F32) GPBO_FAIL(ctx, GPBO_ERR_UNSUPPORTED, w + ": precision must be GPBO_F64 or GPBO_F32");
  for (int t = 0; t < n_ls; ++t)
    if (!(length_scale[t] > 0.0) || !std::isfinite(length_scale[t]))
      GPBO_FAIL(ctx, GPBO_ERR_INVALID, w + ": length_scale must be positive and finite");
  if (!(noise >= 0.0)) GPBO_FAIL(ctx, GPBO_ERR_INVALID, w + ": noise must be >= 0");
  GPBO_HIP(ctx, hipSetDevice(ctx->device));

  m.fitted = false;
  m.wp_packed = false;     // (set by the small fit paths at enqueue time; a failure before finish_enqueue must not leave it behind)
  m.wt_valid = false;      // (K no longer holds the transpose of this slot's W)
  m.M_post = -1;
  const int64_t NP = round_up(N, NB);
  const int DP = pad_dim(d);
  if ((rc = alloc_model(ctx, m, NP, DP))) return rc;
  m.N = N; m.NP = NP; m.d = d; m.DP = DP; m.kernel = kernel; m.precision = precision;
  double* ls_h = (double*)ctx->pinned;
  for (int t = 0; t < GPBO_MAX_DIM; ++t) ls_h[t] = (t < d) ? (n_ls == 1 ? length_scale[0] : length_scale[t]) : 1.0;
  return GPBO_OK;
}

// The fit enqueued on ctx->stream: inputs from the host (X, y_norm) or already on the device (X_dev raw (N, d), y_dev),
// then K, L, W, alpha.  With device inputs every operation is capturable into a hipGraph.
static int enqueue_factor(gpbo_ctx* ctx, Model& m, const double* X, const double* y_norm, const double* X_dev,
                          const double* y_dev, double noise, int** info_host, bool pack = true) {
  int rc;
  const int64_t N = m.N, NP = m.NP;
  if (use_fused(m)) return enqueue_fused(ctx, m, X, y_norm, X_dev, y_dev, noise, 0, 0, 0, info_host, nullptr);
  if (use_mid(m)) {      // inputs straight from pinned host memory (or the resident device copies): no copy / fill nodes
    const double *Xd = X_dev, *yd = y_dev;
    if (!X_dev && (rc = stage_small_inputs(ctx, m, X, y_norm, &Xd, &yd))) return rc;
    ev_begin(ctx, T_FIT);
    if ((rc = launch_mid_inputs(ctx, m, Xd, yd, (const double*)pinned_dev(ctx, ctx->pinned)))) return rc;
    return factor_mid(ctx, m, noise, pack, info_host);
  }
  ev_begin(ctx, T_FIT);
  GPBO_HIP(ctx, lane_h2d(ctx, m.ls, ctx->pinned, PIN_LS_PITCH, GPBO_MAX_DIM * sizeof(double)));
  GPBO_HIP(ctx, lane_memset(ctx, m.yn, (size_t)NP * sizeof(double)));
  if (X_dev) {
    for (int l = 0; l < ctx->lanes; ++l)     // every lane gets the same targets
      GPBO_HIP(ctx, hipMemcpyAsync(m.yn + (int64_t)l * ctx->lane_stride, y_dev, (size_t)N * sizeof(double),
                                   hipMemcpyDeviceToDevice, ctx->stream));
    if ((rc = launch_prescale(ctx, X_dev, N, m.d, m.DP, m.ls, m.Xs, NP))) return rc;
  } else {
    GPBO_HIP(ctx, hipMemcpyAsync(m.tmp, X, (size_t)N * m.d * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    GPBO_HIP(ctx, hipMemcpyAsync(m.yn, y_norm, (size_t)N * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    if ((rc = launch_prescale(ctx, m.tmp, N, m.d, m.DP, m.ls, m.Xs, NP))) return rc;
  }
  return factor_resident(ctx, m, noise, info_host, pack);
}

static int factorize(gpbo_ctx* ctx, Model& m, const char* who, const double* X, const double* y_norm, int64_t N,
                     int d, int kernel, const double* length_scale, int n_ls, double noise, int precision,
                     int** info_host) {
  int rc = prepare_model(ctx, m, who, X && y_norm, N, d, kernel, length_scale, n_ls, noise, precision);
  if (rc) return rc;
  return enqueue_factor(ctx, m, X, y_norm, nullptr, nullptr, noise, info_host);
}

// Tail shared by the fit entry points: pack W for the posterior kernels (enqueue), then wait and resolve the pivot check.
static int finish_enqueue(gpbo_ctx* ctx, Model& m) {
  int rc;
  if (!m.wp_packed && (rc = launch_pack_w(ctx, m))) return rc;
  m.wp_packed = false;
  if (m.precision == GPBO_F32) {   // fp32 posterior: W rounded to fp32 in f32-MFMA fragment order (fit itself is fp64)
    if ((rc = ensure(ctx, &m.Wp32, &m.cap_Wp32, m.NP * m.NP))) return rc;
    if ((rc = launch_pack_w32(ctx, m))) return rc;
  }
  ev_end(ctx, T_FIT);
  return GPBO_OK;
}

static int finish_wait(gpbo_ctx* ctx, Model& m, hipStream_t stream, int* info_h, int* info) {
  GPBO_HIP(ctx, hipStreamSynchronize(stream));
  if (*info_h != 0) {
    if (info) *info = *info_h;
    char b[160];
    snprintf(b, sizeof(b), "the kernel matrix is not positive definite: leading minor of order %d", *info_h);
    GPBO_FAIL(ctx, GPBO_ERR_NOT_PD, b);
  }
  m.fitted = true;
  return GPBO_OK;
}

static int finish_fit(gpbo_ctx* ctx, Model& m, int* info_h, int* info) {
  int rc = finish_enqueue(ctx, m);
  if (rc) return rc;
  return finish_wait(ctx, m, ctx->stream, info_h, info);
}

// a slot whose gpbo_fit_begin has not been waited for must not be refitted or read
static int no_pending_fit(gpbo_ctx* ctx, int slot, const char* who) {
  if (ctx->pending_info[slot])
    GPBO_FAIL(ctx, GPBO_ERR_STATE, std::string(who) + ": the slot has a pending gpbo_fit_begin (call gpbo_fit_wait first)");
  return GPBO_OK;
}

static int wait_all_pending_fits(gpbo_ctx* ctx) {
  for (int s = 0; s < GPBO_MAX_MODELS; ++s)
    if (ctx->pending_info[s]) {
      int rc = gpbo_fit_wait(ctx, s, nullptr);
      if (rc) return rc;
    }
  return GPBO_OK;
}

int gpbo_fit(gpbo_ctx* ctx, int slot, const double* X, const double* y_norm, int64_t N, int d,
             int kernel, const double* length_scale, int n_ls, double noise, int precision,
             int* info) {
  if (info) *info = 0;
  int* info_h = nullptr;
  int rc = check_slot(ctx, slot);
  if (rc) return rc;
  if ((rc = no_pending_fit(ctx, slot, "gpbo_fit"))) return rc;
  rc = factorize(ctx, ctx->models[slot], "gpbo_fit", X, y_norm, N, d, kernel, length_scale, n_ls, noise, precision, &info_h);
  if (rc) return rc;
  return finish_fit(ctx, ctx->models[slot], info_h, info);
}

int gpbo_fit_begin(gpbo_ctx* ctx, int slot, const double* X, const double* y_norm, int64_t N, int d, int kernel,
                   const double* length_scale, int n_ls, double noise, int precision) {
  int rc = check_slot(ctx, slot);
  if (rc) return rc;
  if ((rc = no_pending_fit(ctx, slot, "gpbo_fit_begin"))) return rc;
  GPBO_HIP(ctx, hipSetDevice(ctx->device));
  if (!ctx->slot_stream[slot]) GPBO_HIP(ctx, hipStreamCreateWithFlags(&ctx->slot_stream[slot], hipStreamNonBlocking));
  if (!ctx->info_slots) GPBO_HIP(ctx, hipMalloc((void**)&ctx->info_slots, GPBO_MAX_MODELS * sizeof(int)));
  // the slot's own stream, pinned window and pivot word for the duration of the enqueue (as a gpbo_lml_batch group)
  hipStream_t stream0 = ctx->stream;
  void* pinned0 = ctx->pinned;
  int* info0 = ctx->info_dev;
  const bool timing0 = ctx->no_timing;
  ctx->stream = ctx->slot_stream[slot];
  ctx->pinned = (char*)pinned0 + PIN_LANE_WINDOW * (size_t)(1 + slot);
  ctx->info_dev = ctx->info_slots + slot;
  ctx->no_timing = true;            // the timing events belong to the main stream's calls
  int* info_h = nullptr;
  Model& m = ctx->models[slot];
  rc = factorize(ctx, m, "gpbo_fit_begin", X, y_norm, N, d, kernel, length_scale, n_ls, noise, precision, &info_h);
  if (rc == GPBO_OK) rc = finish_enqueue(ctx, m);
  ctx->stream = stream0; ctx->pinned = pinned0; ctx->info_dev = info0; ctx->no_timing = timing0;
  if (rc) {
    (void)hipStreamSynchronize(ctx->slot_stream[slot]);     // nothing of a half-enqueued fit may still be running
    return rc;
  }
  ctx->pending_info[slot] = info_h;
  return GPBO_OK;
}

int gpbo_fit_wait(gpbo_ctx* ctx, int slot, int* info) {
  if (info) *info = 0;
  int rc = check_slot(ctx, slot);
  if (rc) return rc;
  if (!ctx->pending_info[slot]) GPBO_FAIL(ctx, GPBO_ERR_STATE, "gpbo_fit_wait: the slot has no pending gpbo_fit_begin");
  GPBO_HIP(ctx, hipSetDevice(ctx->device));
  int* info_h = ctx->pending_info[slot];
  ctx->pending_info[slot] = nullptr;
  return finish_wait(ctx, ctx->models[slot], ctx->slot_stream[slot], info_h, info);
}

int gpbo_fit_append(gpbo_ctx* ctx, int slot, const double* x_new, int64_t n_new, int d,
                    const double* y_norm, int64_t n_total, int* info) {
  if (info) *info = 0;
  int rc = check_slot(ctx, slot);
  if (rc) return rc;
  if ((rc = no_pending_fit(ctx, slot, "gpbo_fit_append"))) return rc;
  Model& m = ctx->models[slot];
  if (!m.fitted) GPBO_FAIL(ctx, GPBO_ERR_STATE, "gpbo_fit_append: slot has no fitted model (call gpbo_fit first)");
  if (n_new < 0 || !y_norm || (n_new > 0 && !x_new)) GPBO_FAIL(ctx, GPBO_ERR_INVALID, "gpbo_fit_append: NULL input or n_new < 0");
  if (d != m.d) GPBO_FAIL(ctx, GPBO_ERR_INVALID, "gpbo_fit_append: d differs from the fitted model's");
  if (n_total != m.N + n_new) GPBO_FAIL(ctx, GPBO_ERR_INVALID, "gpbo_fit_append: n_total must be N + n_new");
  if (n_total > (1 << 16)) GPBO_FAIL(ctx, GPBO_ERR_INVALID, "gpbo_fit_append: N out of range [1, 65536]");
  GPBO_HIP(ctx, hipSetDevice(ctx->device));
  m.fitted = false;
  m.wp_packed = false;
  m.wt_valid = false;
  m.M_post = -1;
  const int64_t N0 = m.N;
  const int64_t NP_new = round_up(n_total, NB);
  const bool rebuild = (NP_new != m.NP) || n_new > 16;
  ev_begin(ctx, T_FIT);
  if (NP_new > m.cap_NP) {
    // grow the slot (25% head-room so that a maximize() loop reallocates rarely); the scaled inputs survive
    double* keep = nullptr;
    GPBO_HIP(ctx, hipMalloc((void**)&keep, (size_t)N0 * m.DP * sizeof(double)));
    GPBO_HIP(ctx, hipMemcpyAsync(keep, m.Xs, (size_t)N0 * m.DP * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    double* ls_h = (double*)ctx->pinned;
    GPBO_HIP(ctx, hipMemcpyAsync(ls_h, m.ls, GPBO_MAX_DIM * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    GPBO_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const int DP = m.DP;
    const Model old = m;   // alloc_model resets the descriptor along with the buffers
    if ((rc = alloc_model(ctx, m, round_up(NP_new + NP_new / 4, NB), DP))) { (void)hipFree(keep); return rc; }
    m.N = old.N; m.NP = old.NP; m.d = old.d; m.DP = old.DP; m.kernel = old.kernel; m.precision = old.precision;
    m.noise = old.noise;
    GPBO_HIP(ctx, hipMemcpyAsync(m.ls, ls_h, GPBO_MAX_DIM * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    GPBO_HIP(ctx, hipMemcpyAsync(m.Xs, keep, (size_t)N0 * DP * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    GPBO_HIP(ctx, hipStreamSynchronize(ctx->stream));
    GPBO_HIP(ctx, hipFree(keep));
  }
  // new rows: scaled into Xs[N0 .. NP_new) (zero padded), staged through ctx->Xcs-independent scratch (m.tvec is
  // too small for d > 1, m.tmp is free between fits)
  if (n_new > 0)
    GPBO_HIP(ctx, hipMemcpyAsync(m.tmp, x_new, (size_t)n_new * d * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  if (NP_new > N0)
    if ((rc = launch_prescale(ctx, m.tmp, n_new, d, m.DP, m.ls, m.Xs + N0 * m.DP, NP_new - N0))) return rc;
  GPBO_HIP(ctx, hipMemsetAsync(m.yn, 0, (size_t)NP_new * sizeof(double), ctx->stream));
  GPBO_HIP(ctx, hipMemcpyAsync(m.yn, y_norm, (size_t)n_total * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  int* info_h = nullptr;
  if (rebuild) {
    m.N = n_total;
    m.NP = NP_new;
    if ((rc = factor_resident(ctx, m, m.noise, &info_h))) return rc;
  } else {
    GPBO_HIP(ctx, hipMemsetAsync(ctx->info_dev, 0, sizeof(int), ctx->stream));
    for (int64_t j = N0; j < n_total; ++j)
      if ((rc = launch_append_row(ctx, m, j))) return rc;
    m.N = n_total;
    info_h = (int*)((char*)ctx->pinned + PIN_INFO);
    GPBO_HIP(ctx, hipMemcpyAsync(info_h, ctx->info_dev, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    if ((rc = launch_trmv(ctx, m))) return rc;
  }
  return finish_fit(ctx, m, info_h, info);
}

// The part of a log-marginal-likelihood evaluation that follows the factorisation, enqueued on ctx->stream: the scalar
// terms, K^-1 = W^T W and the gradient reduction, and the copy of the results to the pinned words *out_host.
static int lml_tail(gpbo_ctx* ctx, Model& m, int n_ls, int eval_gradient, double** out_host) {
  int rc;
  // the kernels write the scalars into the pinned words themselves (lane mode: PIN_OUT_PITCH bytes per lane): no copy node
  double* out_h = (double*)((char*)ctx->pinned + (ctx->lanes == 1 ? PIN_LML_OUT : PIN_LANE_OUT));
  double* out_d = (double*)pinned_dev(ctx, out_h);
  const int64_t pitch = (int64_t)(PIN_OUT_PITCH / sizeof(double));
  if (!eval_gradient && (rc = launch_lml_terms(ctx, m, out_d, pitch))) return rc;
  if (eval_gradient) {
    if (m.NP <= mid_max_np()) {
      // small problems: K^-1 tile by tile inside the gradient launch (kinv_grad_kernel), the two LML terms in its final launch
      if ((rc = launch_lml_grad(ctx, m, n_ls, nullptr, m.tmp, out_d, pitch, true))) return rc;
    } else {
      // K^-1 = W^T W (lower tiles) into the K buffer, then the trace reduction; partials go to m.tmp
      GemmArgs g{};
      g.m = (int)m.NP; g.n = (int)m.NP; g.k = (int)m.NP; g.alpha = 1.0; g.beta = 0.0;
      g.A = m.W; g.lda = m.NP; g.a_trans = 1;
      g.B = m.W; g.ldb = m.NP;
      g.C = m.K; g.ldc = m.NP; g.batch = 1; g.lower_only = 1; g.k_from_tile = 1;
      if ((rc = launch_gemm(ctx, g))) return rc;
      if ((rc = launch_lml_grad(ctx, m, n_ls, m.K, m.tmp, out_d, pitch, true))) return rc;     // ... and the two LML terms
    }
  }
  ev_end(ctx, T_FIT);
  *out_host = out_h;
  return GPBO_OK;
}

// One log-marginal-likelihood evaluation enqueued on ctx->stream into model m; results land in the pinned words
// *out_host (yT alpha, sum log L_ii, gradient...) and *info_host once the stream has drained.
static int lml_enqueue(gpbo_ctx* ctx, Model& m, const double* X, const double* y_norm, int64_t N, int d, int kernel,
                       const double* length_scale, int n_ls, double noise, int eval_gradient, double** out_host,
                       int** info_host) {
  int rc = prepare_model(ctx, m, "gpbo_lml", X && y_norm, N, d, kernel, length_scale, n_ls, noise, GPBO_F64);
  if (rc) return rc;
  if (use_fused(m)) return enqueue_fused(ctx, m, X, y_norm, nullptr, nullptr, noise, eval_gradient ? 2 : 1, n_ls, 0, info_host, out_host);
  if ((rc = enqueue_factor(ctx, m, X, y_norm, nullptr, nullptr, noise, info_host, false))) return rc;
  return lml_tail(ctx, m, n_ls, eval_gradient, out_host);
}

static void lml_finish(const double* out_h, const int* info_h, int64_t N, int n_ls, int eval_gradient, double* lml,
                       double* grad, int* info) {
  if (*info_h != 0) {  // sklearn returns -inf and a zero gradient when K is not PD (_gpr.py:588-589)
    if (info) *info = *info_h;
    *lml = -INFINITY;
    if (eval_gradient) for (int t = 0; t < n_ls; ++t) grad[t] = 0.0;
    return;
  }
  *lml = -0.5 * out_h[0] - out_h[1] - 0.5 * (double)N * 1.83787706640934548356;  // log(2 pi)
  if (eval_gradient) for (int t = 0; t < n_ls; ++t) grad[t] = out_h[2 + t];
}

int gpbo_lml(gpbo_ctx* ctx, int slot, const double* X, const double* y_norm, int64_t N, int d,
             int kernel, const double* length_scale, int n_ls, double noise, int eval_gradient,
             double* lml, double* grad, int* info) {
  if (info) *info = 0;
  if (!lml || (eval_gradient && !grad)) GPBO_FAIL(ctx, GPBO_ERR_INVALID, "gpbo_lml: NULL output");
  int rc = check_slot(ctx, slot);
  if (rc) return rc;
  if ((rc = no_pending_fit(ctx, slot, "gpbo_lml"))) return rc;
  int* info_h = nullptr;
  double* out_h = nullptr;
  // the slot is left "unfitted": its W is not packed for the posterior kernel
  rc = lml_enqueue(ctx, ctx->models[slot], X, y_norm, N, d, kernel, length_scale, n_ls, noise, eval_gradient, &out_h, &info_h);
  if (rc) return rc;
  GPBO_HIP(ctx, hipStreamSynchronize(ctx->stream));
  lml_finish(out_h, info_h, N, n_ls, eval_gradient, lml, grad, info);
  return GPBO_OK;
}

}  // extern "C"

namespace gpbo {
// The raw inputs of a theta search, resident on the device for every later gpbo_lml_batch call with X == y_norm == NULL
// (shared with gpbo_group_lml_batch, which puts them on every device of the group before it hands out lanes).
int lml_upload_inputs(gpbo_ctx* ctx, const double* X, const double* y_norm, int64_t N, int d) {
  if (!ctx || !X || !y_norm || N < 1 || d < 1) GPBO_FAIL(ctx, GPBO_ERR_INVALID, "lml inputs: bad arguments");
  GPBO_HIP(ctx, hipSetDevice(ctx->device));
  int rc;
  ctx->lml_N = 0;
  // On the context's own (non-blocking) stream, never the legacy stream: a synchronous hipMemcpy on one thread while another
  // thread's context captures its evaluation graph invalidates that capture on this runtime ("would make the legacy stream
  // depend on a capturing blocking stream"; seen with the lanes of a device group, one thread per device).
#ifndef GPBO_CAPTURE_NOLOCK
  std::shared_lock<std::shared_mutex> not_while_capturing(g_capture_mu);
#endif
  if ((rc = ensure(ctx, &ctx->lml_X, &ctx->cap_lml_X, N * d))) return rc;
  if ((rc = ensure(ctx, &ctx->lml_y, &ctx->cap_lml_y, N))) return rc;
#ifdef GPBO_LML_SYNC_UPLOAD          // experiment builds (scripts/archive/r04_capture_stress.sh): the legacy-stream copies of rounds 2-3
  GPBO_HIP(ctx, hipMemcpy(ctx->lml_X, X, (size_t)N * d * sizeof(double), hipMemcpyHostToDevice));
  GPBO_HIP(ctx, hipMemcpy(ctx->lml_y, y_norm, (size_t)N * sizeof(double), hipMemcpyHostToDevice));
#else
  GPBO_HIP(ctx, hipMemcpyAsync(ctx->lml_X, X, (size_t)N * d * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  GPBO_HIP(ctx, hipMemcpyAsync(ctx->lml_y, y_norm, (size_t)N * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  GPBO_HIP(ctx, hipStreamSynchronize(ctx->stream));
#endif
  ctx->lml_N = N; ctx->lml_d = d;
  return GPBO_OK;
}
}  // namespace gpbo

namespace gpbo {
// One workgroup that spins for `ticks` of the 100 MHz wall clock: the probe of pick_lane_streams.
__global__ void spin_ticks_kernel(long long ticks) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) {}
}

// Do one-workgroup kernels on streams a and b run side by side?  (100 us each: ~120 us side by side, ~220 one behind the other.)
static bool streams_overlap(hipStream_t a, hipStream_t b) {
  double best = 1e30;
  for (int r = 0; r < 3; ++r) {
    (void)hipStreamSynchronize(a); (void)hipStreamSynchronize(b);
    const auto t0 = std::chrono::steady_clock::now();
    spin_ticks_kernel<<<1, 64, 0, a>>>(10000);
    spin_ticks_kernel<<<1, 64, 0, b>>>(10000);
    (void)hipStreamSynchronize(a); (void)hipStreamSynchronize(b);
    best = std::min(best, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
  }
  return best < 170.0;
}

// The streams of the lane groups of gpbo_lml_batch (n_groups > 1 from NP = 2048 on) must sit on DIFFERENT hardware queues, or the
// groups run one behind the other: the runtime deals its (four) queues to streams by load at creation time, and streams created at
// different moments of a process's life do collide — seen in round 6: the third group's stream, created when the first six-lane round
// of a search at N = 4096 came by, shared a queue with the first: 12.1 ms for the round against 10.0 with the streams created back
// to back (scripts/probes/stream_queues.hip: of eight streams created in a row the pairs (0,7), (1,6), (2,5), (3,4) serialise).
// So the streams are PICKED: a candidate is kept when a 100 us one-workgroup kernel on it runs side by side with one on every
// stream already chosen; rejected candidates are held until the picking is over (so that the next one lands elsewhere).  Once per
// context and group count, ~1 ms; the first four groups only (there are four queues).
static int pick_lane_streams(gpbo_ctx* ctx, int n_groups) {
  if (!ctx->lml_stream[0]) GPBO_HIP(ctx, hipStreamCreateWithFlags(&ctx->lml_stream[0], hipStreamNonBlocking));
  if (ctx->lane_streams_picked < 1) ctx->lane_streams_picked = 1;
  if (n_groups <= ctx->lane_streams_picked) return GPBO_OK;
  std::shared_lock<std::shared_mutex> not_while_capturing(g_capture_mu);
  std::vector<hipStream_t> rejected;
  for (int g = ctx->lane_streams_picked; g < n_groups; ++g) {
    if (ctx->lml_stream[g]) continue;
    hipStream_t cand = nullptr;
    for (int attempt = 0; attempt < 8; ++attempt) {
      GPBO_HIP(ctx, hipStreamCreateWithFlags(&cand, hipStreamNonBlocking));
      if (g >= 4) break;
      spin_ticks_kernel<<<1, 64, 0, cand>>>(100);          // (the first launch on a stream sets its queue up)
      bool ok = true;
      for (int c = 0; c < g && ok; ++c) ok = streams_overlap(ctx->lml_stream[c], cand);
      if (ok || attempt == 7) break;
      rejected.push_back(cand);
      cand = nullptr;
    }
    ctx->lml_stream[g] = cand;
  }
  for (auto st : rejected) (void)hipStreamDestroy(st);
  (void)hipGetLastError();
  ctx->lane_streams_picked = n_groups;
  return GPBO_OK;
}
}  // namespace gpbo

extern "C" {

int gpbo_lml_batch(gpbo_ctx* ctx, int n_theta, const double* X, const double* y_norm, int64_t N, int d, int kernel,
                   const double* length_scales, int n_ls, double noise, int eval_gradient, double* lml, double* grad,
                   int* info) {
  if (!ctx) return GPBO_ERR_INVALID;
  if (n_theta < 1 || n_theta > GPBO_LML_BATCH_MAX) GPBO_FAIL(ctx, GPBO_ERR_INVALID, "gpbo_lml_batch: n_theta out of range [1, 8]");
  if (!lml || !length_scales || (eval_gradient && !grad) || (!X) != (!y_norm))
    GPBO_FAIL(ctx, GPBO_ERR_INVALID, "gpbo_lml_batch: NULL argument");
  {
    int rcw = wait_all_pending_fits(ctx);   // their pinned windows are the ones the lane groups are about to use
    if (rcw) return rcw;
  }
  const bool reuse_inputs = !X;     // X == y_norm == NULL: the inputs of the previous call are still on the device
  if (reuse_inputs && (ctx->lml_N != N || ctx->lml_d != d))
    GPBO_FAIL(ctx, GPBO_ERR_STATE, "gpbo_lml_batch: no resident inputs of this shape (pass X and y_norm)");
  if (N < 1 || N > (1 << 16)) GPBO_FAIL(ctx, GPBO_ERR_INVALID, "gpbo_lml_batch: N out of range [1, 65536]");
  if (d < 1 || d > GPBO_MAX_DIM) GPBO_FAIL(ctx, GPBO_ERR_UNSUPPORTED, "gpbo_lml_batch: d out of range [1, 64]");
  if (kernel != GPBO_KERNEL_RBF && kernel != GPBO_KERNEL_MATERN25)
    GPBO_FAIL(ctx, GPBO_ERR_UNSUPPORTED, "gpbo_lml_batch: kernel must be RBF or Matern(nu=2.5)");
  if (n_ls != 1 && n_ls != d) GPBO_FAIL(ctx, GPBO_ERR_INVALID, "gpbo_lml_batch: length_scale must have 1 or d entries");
  if (!(noise >= 0.0)) GPBO_FAIL(ctx, GPBO_ERR_INVALID, "gpbo_lml_batch: noise must be >= 0");
  for (int64_t t = 0; t < (int64_t)n_theta * n_ls; ++t)
    if (!(length_scales[t] > 0.0) || !std::isfinite(length_scales[t]))
      GPBO_FAIL(ctx, GPBO_ERR_INVALID, "gpbo_lml_batch: length_scale must be positive and finite");
  GPBO_HIP(ctx, hipSetDevice(ctx->device));

  // One slab, one layout per lane: every kernel of the evaluation runs ONCE for all lanes (lane = a grid dimension,
  // lane l's buffers l * stride doubles behind lane 0's) — the command processor sees ~60 dispatches per batch, not
  // ~60 per theta.  Model slots and their fits are not touched.
  const int64_t NP = round_up(N, NB);
  const int DP = pad_dim(d);
  auto up = [](int64_t v) { return round_up(v, 32); };
  int64_t off = 0;
  const int64_t o_ls = off;    off += up(GPBO_MAX_DIM);
  const int64_t o_Xs = off;    off += up(NP * DP);
  const int64_t o_K = off;     off += up(NP * NP);
  const int64_t o_L = off;     off += up(NP * NP);
  const int64_t o_W = off;     off += up(NP * NP);
  const int64_t o_dinv = off;  off += up((NP / NB) * NB * NB);
  const int64_t o_tmp = off;   off += up(std::max<int64_t>(NP * NP / 2, NP * (int64_t)GPBO_MAX_DIM));
  const int64_t o_yn = off;    off += up(NP);
  const int64_t o_tvec = off;  off += up(NP);
  const int64_t o_alpha = off; off += up(NP);
  const int64_t o_scal = off;  off += up(8 + GPBO_MAX_DIM);
  const int64_t o_info = off;  off += 32;
  const int64_t stride = off;
  int rc;
  {
    std::shared_lock<std::shared_mutex> not_while_capturing(g_capture_mu);      // (hipMalloc / hipFree when the slab grows)
    if ((rc = ensure(ctx, &ctx->lml_slab, &ctx->cap_lml_slab, stride * n_theta))) return rc;
  }
  if (!reuse_inputs && (rc = lml_upload_inputs(ctx, X, y_norm, N, d))) return rc;
  double* base = ctx->lml_slab;
  // Lanes are processed in groups: a group runs the launch sequence once for its lanes on its own stream.  Small
  // problems are dispatch-bound (every kernel is tiny): ONE group of all lanes.  From NP = 2048 on the big GEMMs fill
  // the chip by themselves and what is left to win is hiding one lane's latency-bound steps (the diagonal-block
  // kernels) behind another lane's GEMMs — which a SECOND stream does and a third does not: three lanes on three streams
  // take what two take plus one alone (2.16 against 1.22 + 0.94 ms at N = 2048; it is not the hardware queues — four streams of
  // one-workgroup kernels do run side by side, scripts/probes/stream_queues.hip — but what two evaluations in the same phase
  // leave free of the chip).  So: two groups, and inside a group lane = a grid dimension, where the chain's launches are
  // shared (the diagonal blocks of all its lanes factor side by side in one step launch).  Until round 6: one lane per group
  // from NP = 2048 on.  profiles/r06_lanes_grouping.json, ms for 3 / 4 / 6 lanes:
  //   N = 2048: one lane per group 2.16 / 2.36 / 2.72, two groups 1.48 / 1.72 / 2.25;  N = 3072: 4.01 / 4.63 / 5.78 -> 3.17 / 3.82 / 5.31
  //   N = 4096: 5.59 / 8.33 / 11.14 -> (three lanes on three streams stay) / 7.19 / 10.01 with two lanes per group;  N = 6144: 21.5 -> 18.9 at 4
  int per_group = n_theta;
  if (NP >= 4096) per_group = (n_theta <= 3) ? 1 : 2;
  else if (NP >= 2048) per_group = (n_theta + 1) / 2;
  if (const char* e = dbg_env("GPBO_LML_PER_GROUP")) per_group = std::max(1, std::min(atoi(e), n_theta));   // A/B runs (debug build)
  const int n_groups = (n_theta + per_group - 1) / per_group;
  if ((rc = pick_lane_streams(ctx, n_groups))) return rc;
  static const bool graphs_allowed = !(dbg_env("GPBO_LML_GRAPH") && dbg_env("GPBO_LML_GRAPH")[0] == '0');
  hipStream_t stream0 = ctx->stream;
  void* red0 = ctx->red; int64_t cap_red0 = ctx->cap_red;
  int* info0 = ctx->info_dev;
  void* pinned0 = ctx->pinned;
  auto restore = [&]() {
    ctx->stream = stream0; ctx->red = red0; ctx->cap_red = cap_red0; ctx->info_dev = info0; ctx->pinned = pinned0;
    ctx->lanes = 1; ctx->lane_stride = 0; ctx->no_timing = false; ctx->no_lookahead = false;
  };
  rc = GPBO_OK;
  for (int g = 0; g < n_groups && rc == GPBO_OK; ++g) {
    const int l0 = g * per_group;
    const int gl = std::min(per_group, n_theta - l0);             // lanes of this group
    double* gbase = base + (int64_t)l0 * stride;
    Model m;     // a view of the group's first lane (not owning)
    m.N = N; m.NP = NP; m.d = d; m.DP = DP; m.kernel = kernel; m.precision = GPBO_F64; m.noise = noise;
    m.ls = gbase + o_ls; m.Xs = gbase + o_Xs; m.K = gbase + o_K; m.L = gbase + o_L; m.W = gbase + o_W;
    m.dinv = gbase + o_dinv; m.tmp = gbase + o_tmp; m.yn = gbase + o_yn; m.tvec = gbase + o_tvec; m.alpha = gbase + o_alpha;
    // theta enters through the pinned length-scale words ([lane][64]) that the sequence's first copy reads
    char* window = (char*)pinned0 + PIN_LANE_WINDOW * (size_t)(1 + g);
    double* ls_h = (double*)window;
    for (int l = 0; l < gl; ++l)
      for (int t = 0; t < GPBO_MAX_DIM; ++t)
        ls_h[l * GPBO_MAX_DIM + t] =
            (t < d) ? (n_ls == 1 ? length_scales[l0 + l] : length_scales[(int64_t)(l0 + l) * n_ls + t]) : 1.0;
    ctx->stream = ctx->lml_stream[g];
    ctx->red = gbase + o_scal; ctx->cap_red = (8 + GPBO_MAX_DIM) * 8;
    ctx->info_dev = (int*)(gbase + o_info);
    ctx->pinned = window;
    ctx->lanes = gl; ctx->lane_stride = stride;
    ctx->no_timing = true;
    static const bool la_lanes = dbg_env("GPBO_CHOL_LA_LANES") && dbg_env("GPBO_CHOL_LA_LANES")[0] == '1';
    ctx->no_lookahead = n_groups > 1 && !la_lanes;
    const bool fused = use_fused(m);     // one launch for the whole group: nothing to capture
    // ... and the strip path's ~17 launches are enqueued faster than the device runs them: replaying them from a graph bought nothing
    // at a fixed shape (six lanes at N = 512: 0.292 ms replayed, 0.283 launched) and cost a maximize() loop — whose N grows by one
    // per step, a new shape every call — ~1 ms of capture + instantiation per suggest() (profiles/r05_maximize_loop.json)
    const bool no_graph = fused || use_mid(m);
    auto enqueue = [&](double** oh, int** ih) {
      if (fused) return enqueue_fused(ctx, m, nullptr, nullptr, ctx->lml_X, ctx->lml_y, noise, eval_gradient ? 2 : 1, n_ls, 0, ih, oh);
      int r = enqueue_factor(ctx, m, nullptr, nullptr, ctx->lml_X, ctx->lml_y, noise, ih, false);
      if (r == GPBO_OK) r = lml_tail(ctx, m, n_ls, eval_gradient, oh);
      return r;
    };
    // An evaluation is ~60 short launches: the second time the same problem shape comes by, the group's sequence is
    // captured into a hipGraph and from then on replayed with one launch.
    LmlLane* found = nullptr;
    LmlLane* victim = &ctx->lml_lane[0];
    for (auto& e : ctx->lml_lane) {
      if (e.seen && e.N == N && e.d == d && e.kernel == kernel && e.n_ls == n_ls && e.eval_gradient == eval_gradient &&
          e.noise == noise && e.lanes == gl && e.group == g && e.X == ctx->lml_X && e.y == ctx->lml_y && e.K == gbase) {
        found = &e;
        break;
      }
      if ((!e.seen && victim->seen) || (e.seen == victim->seen && e.used < victim->used)) victim = &e;
    }
    const bool same = found != nullptr;
    LmlLane& key = same ? *found : *victim;
    if (!same) {
      if (key.exec) { (void)hipGraphExecDestroy(key.exec); key.exec = nullptr; }
      key.seen = false;
    }
    key.used = ++ctx->lml_lane_clock;
    bool launched = false;
    if (same && key.exec && !no_graph) {
      hipError_t e = hipGraphLaunch(key.exec, ctx->stream);
      if (e != hipSuccess) { restore(); GPBO_HIP(ctx, e); }
      launched = true;
    } else {
      if (key.exec) { (void)hipGraphExecDestroy(key.exec); key.exec = nullptr; }
      if (same && graphs_allowed && !ctx->lml_graph_off && !no_graph) {
        hipGraph_t graph = nullptr;
        double* oh = nullptr; int* ih = nullptr;
        hipError_t e;
        int crc = GPBO_ERR_HIP;
        bool instantiated = false;
        {
          // Captures of different contexts (the lanes of a device group run on one thread per device) are taken one at a
          // time, process-wide: concurrent captures were seen to invalidate each other on this runtime
          // (tests/test_gpu_sharded.py, three virtual ranks).  A capture is ~1 ms of host work once per problem shape.
          CAPTURE_LOCK;
          e = hipStreamBeginCapture(ctx->stream, GPBO_CAPTURE_MODE);
          if (e == hipSuccess) {
            crc = enqueue(&oh, &ih);
            e = hipStreamEndCapture(ctx->stream, &graph);
          }
          instantiated = e == hipSuccess && crc == GPBO_OK && graph &&
                         hipGraphInstantiate(&key.exec, graph, nullptr, nullptr, 0) == hipSuccess && key.exec;
        }
        if (instantiated) {
          (void)hipGraphDestroy(graph);
          e = hipGraphLaunch(key.exec, ctx->stream);
          if (e != hipSuccess) { restore(); GPBO_HIP(ctx, e); }
          launched = true;
        } else {   // capture is not available for this sequence on this runtime: direct launches from now on
          if (graph) (void)hipGraphDestroy(graph);
          if (key.exec) { (void)hipGraphExecDestroy(key.exec); key.exec = nullptr; }
          (void)hipGetLastError();
          ctx->lml_graph_off = true;
          CAPTURE_TRACE("gpbo: lml graph capture failed on device %d (hip %d, rc %d): direct launches from now on\n", ctx->device, (int)e, crc);
          // an invalidated capture can outlive hipStreamEndCapture on this runtime: the direct launches below need a live stream
          hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
          if (hipStreamIsCapturing(ctx->stream, &st) != hipSuccess || st != hipStreamCaptureStatusNone) {
            (void)hipGetLastError();
            hipStream_t fresh = nullptr;
            if (hipStreamCreateWithFlags(&fresh, hipStreamNonBlocking) == hipSuccess) {
              drop_lookahead(ctx, ctx->lml_stream[g]);
              (void)hipStreamDestroy(ctx->lml_stream[g]);
              ctx->lml_stream[g] = fresh;
              ctx->stream = fresh;
              CAPTURE_TRACE("gpbo: replaced the stream the dead capture sat on (device %d)\n", ctx->device);
            }
          }
        }
      }
    }
    if (!launched) {
      double* oh = nullptr; int* ih = nullptr;
      rc = enqueue(&oh, &ih);
      key.seen = (rc == GPBO_OK);
      key.N = N; key.d = d; key.kernel = kernel; key.n_ls = n_ls; key.eval_gradient = eval_gradient; key.noise = noise;
      key.lanes = gl; key.group = g; key.X = ctx->lml_X; key.y = ctx->lml_y; key.K = gbase;
    }
  }
  restore();
  for (int g = 0; g < n_groups; ++g) {
    hipError_t e = hipStreamSynchronize(ctx->lml_stream[g]);
    if (e != hipSuccess && rc == GPBO_OK) GPBO_HIP(ctx, e);
  }
  if (rc) return rc;
  for (int i = 0; i < n_theta; ++i) {
    const int g = i / per_group, l = i - g * per_group;
    const int gl = std::min(per_group, n_theta - g * per_group);
    const char* window = (const char*)ctx->pinned + PIN_LANE_WINDOW * (size_t)(1 + g);
    const char* out_h = window + (gl == 1 ? PIN_LML_OUT : PIN_LANE_OUT);      // where lml_tail / factor_resident put the
    const char* info_h = window + (gl == 1 ? PIN_INFO : PIN_LANE_INFO);    // results (single / lane mode)
    if (info) info[i] = 0;
    lml_finish((const double*)(out_h + (size_t)l * PIN_OUT_PITCH), (const int*)(info_h + (size_t)l * PIN_INFO_PITCH), N,
               n_ls, eval_gradient, lml + i, eval_gradient ? grad + (size_t)i * n_ls : nullptr, info ? info + i : nullptr);
  }
  return GPBO_OK;
}

static int need_fitted(gpbo_ctx* ctx, int slot) {
  int rc = check_slot(ctx, slot);
  if (rc) return rc;
  if (!ctx->models[slot].fitted) GPBO_FAIL(ctx, GPBO_ERR_STATE, "model slot has not been fitted");
  GPBO_HIP(ctx, hipSetDevice(ctx->device));
  return GPBO_OK;
}

int gpbo_get_K(gpbo_ctx* ctx, int slot, double* out) {
  int rc = need_fitted(ctx, slot);
  if (rc) return rc;
  // the fit assembles K directly into the buffer it factorises; the parity accessor re-assembles it from the
  // device-resident scaled inputs (same kernel, same bits)
  Model& m = ctx->models[slot];
  m.wt_valid = false;
  if ((rc = use_mid(m) ? launch_kmat_q(ctx, m, m.noise, m.K) : launch_kmat(ctx, m, m.noise, m.K))) return rc;   // the fit's own kernel
  return copy_square(ctx, m, m.K, out, 0);
}
int gpbo_get_L(gpbo_ctx* ctx, int slot, double* out) {
  int rc = need_fitted(ctx, slot);
  if (rc) return rc;
  return copy_square(ctx, ctx->models[slot], ctx->models[slot].L, out, 1);
}
int gpbo_get_Linv(gpbo_ctx* ctx, int slot, double* out) {
  int rc = need_fitted(ctx, slot);
  if (rc) return rc;
  return copy_square(ctx, ctx->models[slot], ctx->models[slot].W, out, 1);
}
int gpbo_get_alpha(gpbo_ctx* ctx, int slot, double* out) {
  int rc = need_fitted(ctx, slot);
  if (rc) return rc;
  Model& m = ctx->models[slot];
  GPBO_HIP(ctx, hipMemcpyAsync(out, m.alpha, (size_t)m.N * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  GPBO_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return GPBO_OK;
}

int gpbo_set_candidates(gpbo_ctx* ctx, const double* Xc, int64_t M, int d) {
  if (!ctx) return GPBO_ERR_INVALID;
  if (!Xc || M < 1 || d < 1 || d > GPBO_MAX_DIM) GPBO_FAIL(ctx, GPBO_ERR_INVALID, "set_candidates: bad arguments");
  GPBO_HIP(ctx, hipSetDevice(ctx->device));
  int rc;
  ctx->raw_valid = false;
  if ((rc = ensure(ctx, &ctx->Xc, &ctx->cap_Xc, M * d))) return rc;
  const size_t bytes = (size_t)M * d * sizeof(double);
  if (bytes <= SMALL_PIN_IN) {
    // small batch: through the pinned block, no stream synchronisation (the block is reused only after its last copy)
    if (ctx->small_ev_pending) GPBO_HIP(ctx, hipEventSynchronize(ctx->small_ev));
    memcpy(ctx->small_pinned, Xc, bytes);
    GPBO_HIP(ctx, hipMemcpyAsync(ctx->Xc, ctx->small_pinned, bytes, hipMemcpyHostToDevice, ctx->stream));
    GPBO_HIP(ctx, hipEventRecord(ctx->small_ev, ctx->stream));
    ctx->small_ev_pending = true;
  } else {
    GPBO_HIP(ctx, hipMemcpyAsync(ctx->Xc, Xc, bytes, hipMemcpyHostToDevice, ctx->stream));
    GPBO_HIP(ctx, hipStreamSynchronize(ctx->stream));
  }
  ctx->M = M;
  ctx->d_c = d;
  for (auto& m : ctx->models) m.M_post = -1;
  return GPBO_OK;
}

int gpbo_posterior(gpbo_ctx* ctx, int slot, double y_mean, double y_std, double* mu, double* sd) {
  int rc = need_fitted(ctx, slot);
  if (rc) return rc;
  Model& m = ctx->models[slot];
  if (ctx->M < 1) GPBO_FAIL(ctx, GPBO_ERR_STATE, "posterior: no candidates resident (call gpbo_set_candidates)");
  if (ctx->d_c != m.d) GPBO_FAIL(ctx, GPBO_ERR_INVALID, "posterior: candidate dimension differs from the fitted model");
  if ((rc = launch_posterior(ctx, m, ctx->M, y_mean, y_std))) return rc;
  const size_t bytes = (size_t)ctx->M * sizeof(double);
  if ((mu || sd) && bytes <= SMALL_PIN_OUT) {      // small batch: results land in the pinned block, then in the caller's arrays
    double* hmu = (double*)((char*)ctx->small_pinned + SMALL_PIN_IN);
    double* hsd = (double*)((char*)ctx->small_pinned + SMALL_PIN_IN + SMALL_PIN_OUT);
    if (mu) GPBO_HIP(ctx, hipMemcpyAsync(hmu, m.mu, bytes, hipMemcpyDeviceToHost, ctx->stream));
    if (sd) GPBO_HIP(ctx, hipMemcpyAsync(hsd, m.sd, bytes, hipMemcpyDeviceToHost, ctx->stream));
    GPBO_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (mu) memcpy(mu, hmu, bytes);
    if (sd) memcpy(sd, hsd, bytes);
    return GPBO_OK;
  }
  if (mu) GPBO_HIP(ctx, hipMemcpyAsync(mu, m.mu, bytes, hipMemcpyDeviceToHost, ctx->stream));
  if (sd) GPBO_HIP(ctx, hipMemcpyAsync(sd, m.sd, bytes, hipMemcpyDeviceToHost, ctx->stream));
  if (mu || sd) GPBO_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return GPBO_OK;
}

int gpbo_predict(gpbo_ctx* ctx, int slot, const double* Xc, int64_t M, int d, double y_mean,
                 double y_std, double* mu, double* sd) {
  int rc = gpbo_set_candidates(ctx, Xc, M, d);
  if (rc) return rc;
  return gpbo_posterior(ctx, slot, y_mean, y_std, mu, sd);
}

int gpbo_take_negative_variance_flag(gpbo_ctx* ctx, int* seen) {
  if (!ctx || !seen) GPBO_FAIL(ctx, GPBO_ERR_INVALID, "gpbo_take_negative_variance_flag: null argument");
  GPBO_HIP(ctx, hipStreamSynchronize(ctx->stream));   // the finalize kernels of every enqueued posterior have written it
  int* flag = (int*)((char*)ctx->pinned_aux + PIN_AUX_NEGVAR);
  *seen = *flag;
  *flag = 0;
  return GPBO_OK;
}

int gpbo_predict_cov(gpbo_ctx* ctx, int slot, const double* Xc, int64_t M, int d, double y_mean, double y_std,
                     double* mu, double* cov) {
  if (!ctx) return GPBO_ERR_INVALID;
  if (!cov) GPBO_FAIL(ctx, GPBO_ERR_INVALID, "predict_cov: NULL output");
  if (M < 1 || M > 16384) GPBO_FAIL(ctx, GPBO_ERR_INVALID, "predict_cov: M out of range [1, 16384]");
  int rc = gpbo_set_candidates(ctx, Xc, M, d);
  if (rc) return rc;
  if ((rc = need_fitted(ctx, slot))) return rc;
  Model& m = ctx->models[slot];
  if (d != m.d) GPBO_FAIL(ctx, GPBO_ERR_INVALID, "predict_cov: candidate dimension differs from the fitted model");
  if (mu) {   // the mean through the usual posterior path (_gpr.py:443-447)
    if ((rc = launch_posterior(ctx, m, M, y_mean, y_std))) return rc;
    GPBO_HIP(ctx, hipMemcpyAsync(mu, m.mu, (size_t)M * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  }
  double* cov_dev = nullptr;
  int64_t ldc = 0;
  if ((rc = launch_posterior_cov(ctx, m, M, y_std, &cov_dev, &ldc))) return rc;
  GPBO_HIP(ctx, hipMemcpy2DAsync(cov, (size_t)M * sizeof(double), cov_dev, (size_t)ldc * sizeof(double), (size_t)M * sizeof(double),
                                 (size_t)M, hipMemcpyDeviceToHost, ctx->stream));
  GPBO_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return GPBO_OK;
}

int gpbo_predict_grad(gpbo_ctx* ctx, int slot, const double* Xc, int64_t M, int d, double y_mean, double y_std,
                      double* mu, double* sd, double* dmu, double* dsd) {
  if (!ctx) return GPBO_ERR_INVALID;
  if (!mu || !sd || !dmu || !dsd) GPBO_FAIL(ctx, GPBO_ERR_INVALID, "predict_grad: NULL output");
  if (M < 1 || M > 4 * GPBO_MAX_SEEDS) GPBO_FAIL(ctx, GPBO_ERR_INVALID, "predict_grad: M out of range [1, 256]");
  int rc = gpbo_set_candidates(ctx, Xc, M, d);
  if (rc) return rc;
  if ((rc = need_fitted(ctx, slot))) return rc;
  Model& m = ctx->models[slot];
  if (d != m.d) GPBO_FAIL(ctx, GPBO_ERR_INVALID, "predict_grad: candidate dimension differs from the fitted model");
  double *dmu_dev = nullptr, *dsd_dev = nullptr;
  if ((rc = launch_posterior_grad(ctx, m, M, y_mean, y_std, &dmu_dev, &dsd_dev))) return rc;
  GPBO_HIP(ctx, hipMemcpyAsync(mu, m.mu, (size_t)M * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  GPBO_HIP(ctx, hipMemcpyAsync(sd, m.sd, (size_t)M * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  GPBO_HIP(ctx, hipMemcpyAsync(dmu, dmu_dev, (size_t)M * d * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  GPBO_HIP(ctx, hipMemcpyAsync(dsd, dsd_dev, (size_t)M * d * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  GPBO_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return GPBO_OK;
}

int gpbo_acq_argbest(gpbo_ctx* ctx, int acq, double acq_param, double y_max, int n_constraints,
                     const double* lb, const double* ub, int k_seeds, int64_t index_offset,
                     int64_t* best_idx, double* best_val, int64_t* seed_idx, double* seed_val,
                     double* ys_out) {
  AcqArgs a{};
  int rc = build_acq_args(ctx, "acq_argbest", acq, acq_param, y_max, n_constraints, lb, ub, k_seeds, best_idx, best_val,
                          seed_idx, seed_val, &a);
  if (rc) return rc;
  ev_begin(ctx, T_ACQ);
  rc = launch_acq_argbest(ctx, a, ctx->M, k_seeds, index_offset, best_idx, best_val, seed_idx, seed_val);
  ev_end(ctx, T_ACQ);
  if (rc) return rc;
  if (ys_out) {
    GPBO_HIP(ctx, hipMemcpyAsync(ys_out, ctx->ys, (size_t)ctx->M * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    GPBO_HIP(ctx, hipStreamSynchronize(ctx->stream));
  }
  return GPBO_OK;
}

int gpbo_set_timing(gpbo_ctx* ctx, int on) {
  if (!ctx) return GPBO_ERR_INVALID;
  ctx->timing_off = !on;
  if (!on)
    for (int i = 0; i < T_COUNT; ++i) ctx->ev[i].used = false;     // gpbo_last_timings answers -1 from here on
  return GPBO_OK;
}

int gpbo_last_timings(gpbo_ctx* ctx, float* ms, int n) {
  if (!ctx || !ms) return GPBO_ERR_INVALID;
  GPBO_HIP(ctx, hipStreamSynchronize(ctx->stream));
  for (int i = 0; i < n; ++i) {
    ms[i] = -1.f;
    if (i < T_COUNT && ctx->ev[i].used) {
      float t = 0.f;
      if (hipEventElapsedTime(&t, ctx->ev[i].a, ctx->ev[i].b) == hipSuccess) ms[i] = t;
    }
  }
  return GPBO_OK;
}

#ifdef GPBO_DEBUG   // self-tests and micro-benchmarks: libgpbo_dbg.so only (include/gpbo.h, "debug build")
int gpbo_debug_cholesky(gpbo_ctx* ctx, const double* A, int64_t n, int variant, int iters, double* L_out, double* dinv_out,
                        long long* stamps_out, double* ms_out, int* info_out) {
  if (!ctx || !A || n < 64 || n % 64 || iters < 1 || variant != 3) return GPBO_ERR_INVALID;   // (variant 2, the round-2 schedule, is retired)
  GPBO_HIP(ctx, hipSetDevice(ctx->device));
  Model m;
  int rc = alloc_model(ctx, m, n, 4);
  if (rc) return rc;
  m.N = m.NP = n; m.d = 1; m.DP = 4;
  long long* stamps_dev = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  const size_t sq = (size_t)n * n * sizeof(double);
  auto done = [&](int code) {
    if (stamps_dev) (void)hipFree(stamps_dev);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    free_model(m);
    return code;
  };
  if (hipMalloc((void**)&stamps_dev, 16 * sizeof(long long)) != hipSuccess) return done(GPBO_ERR_HIP);
  (void)hipMemset(stamps_dev, 0, 16 * sizeof(long long));
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  m.wt_valid = false;
  if (hipMemcpy(m.K, A, sq, hipMemcpyHostToDevice) != hipSuccess) return done(GPBO_ERR_HIP);
  double best = 1e30;
  for (int it = 0; it < iters && !rc; ++it) {
    (void)hipMemcpyAsync(m.L, m.K, sq, hipMemcpyDeviceToDevice, ctx->stream);
    (void)hipMemsetAsync(ctx->info_dev, 0, sizeof(int), ctx->stream);
    (void)hipEventRecord(e0, ctx->stream);
    rc = launch_cholesky128(ctx, m, chol_outer_width(m.NP), it == iters - 1 ? stamps_dev : nullptr);   // stamps: the last (warm) run
    (void)hipEventRecord(e1, ctx->stream);
    if (hipStreamSynchronize(ctx->stream) != hipSuccess) return done(GPBO_ERR_HIP);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  if (rc) return done(rc);
  if (ms_out) *ms_out = best;
  if (L_out && hipMemcpy(L_out, m.L, sq, hipMemcpyDeviceToHost) != hipSuccess) return done(GPBO_ERR_HIP);
  if (dinv_out && hipMemcpy(dinv_out, m.dinv, (size_t)(n / 64) * 4096 * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess)
    return done(GPBO_ERR_HIP);
  if (stamps_out && hipMemcpy(stamps_out, stamps_dev, 16 * sizeof(long long), hipMemcpyDeviceToHost) != hipSuccess) return done(GPBO_ERR_HIP);
  if (info_out && hipMemcpy(info_out, ctx->info_dev, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return done(GPBO_ERR_HIP);
  return done(GPBO_OK);
}

int gpbo_debug_gemm(gpbo_ctx* ctx, int m, int n, int k, double alpha, const double* A,
                    const double* B, int b_trans, double beta, double* C) {
  if (!ctx || !A || !B || !C) return GPBO_ERR_INVALID;
  GPBO_HIP(ctx, hipSetDevice(ctx->device));
  double *dA = nullptr, *dB = nullptr, *dC = nullptr;
  GPBO_HIP(ctx, hipMalloc((void**)&dA, (size_t)m * k * 8));
  GPBO_HIP(ctx, hipMalloc((void**)&dB, (size_t)n * k * 8));
  GPBO_HIP(ctx, hipMalloc((void**)&dC, (size_t)m * n * 8));
  GPBO_HIP(ctx, hipMemcpy(dA, A, (size_t)m * k * 8, hipMemcpyHostToDevice));
  GPBO_HIP(ctx, hipMemcpy(dB, B, (size_t)n * k * 8, hipMemcpyHostToDevice));
  GPBO_HIP(ctx, hipMemcpy(dC, C, (size_t)m * n * 8, hipMemcpyHostToDevice));
  GemmArgs g{};
  g.m = m; g.n = n; g.k = k; g.alpha = alpha; g.beta = beta;
  g.A = dA; g.lda = k; g.B = dB; g.ldb = b_trans ? k : n; g.b_trans = b_trans;
  g.C = dC; g.ldc = n; g.batch = 1;
  int rc = launch_gemm(ctx, g);
  if (!rc) {
    GPBO_HIP(ctx, hipStreamSynchronize(ctx->stream));
    GPBO_HIP(ctx, hipMemcpy(C, dC, (size_t)m * n * 8, hipMemcpyDeviceToHost));
  }
  (void)hipFree(dA); (void)hipFree(dB); (void)hipFree(dC);
  return rc;
}

int gpbo_debug_gemm_bench(gpbo_ctx* ctx, int m, int n, int k, int b_trans, int a_trans, int lower_only, int iters,
                          double* out) {
  if (!ctx || !out || m < 64 || n < 64 || k < 16 || iters < 1 || (a_trans && b_trans)) return GPBO_ERR_INVALID;
  GPBO_HIP(ctx, hipSetDevice(ctx->device));
  double *dA = nullptr, *dB = nullptr, *dC = nullptr;
  const size_t na = (size_t)m * k, nb = (size_t)n * k, nc = (size_t)m * n;
  GPBO_HIP(ctx, hipMalloc((void**)&dA, na * 8));
  GPBO_HIP(ctx, hipMalloc((void**)&dB, nb * 8));
  GPBO_HIP(ctx, hipMalloc((void**)&dC, nc * 8));
  {
    // random (not zero) operands: the clock a kernel sustains depends on the data (MI355X_MICROARCH.md, DVFS)
    std::vector<double> h(std::max(na, nb));
    uint64_t sx = 0x9E3779B97F4A7C15ull;
    for (auto& v : h) { sx ^= sx << 13; sx ^= sx >> 7; sx ^= sx << 17; v = (double)(sx >> 11) * (1.0 / 9007199254740992.0) - 0.5; }
    GPBO_HIP(ctx, hipMemcpy(dA, h.data(), na * 8, hipMemcpyHostToDevice));
    GPBO_HIP(ctx, hipMemcpy(dB, h.data(), nb * 8, hipMemcpyHostToDevice));
    GPBO_HIP(ctx, hipMemset(dC, 0, nc * 8));
  }
  GemmArgs g{};
  g.m = m; g.n = n; g.k = k; g.alpha = 1.0; g.beta = 0.0;
  g.A = dA; g.lda = a_trans ? m : k; g.a_trans = a_trans;
  g.B = dB; g.ldb = b_trans ? k : n; g.b_trans = b_trans;
  g.C = dC; g.ldc = n; g.batch = 1; g.lower_only = lower_only;
  int rc = launch_gemm(ctx, g);    // warm-up
  hipEvent_t e0, e1;
  GPBO_HIP(ctx, hipEventCreate(&e0));
  GPBO_HIP(ctx, hipEventCreate(&e1));
  GPBO_HIP(ctx, hipEventRecord(e0, ctx->stream));
  for (int it = 0; it < iters && rc == GPBO_OK; ++it) rc = launch_gemm(ctx, g);
  GPBO_HIP(ctx, hipEventRecord(e1, ctx->stream));
  GPBO_HIP(ctx, hipStreamSynchronize(ctx->stream));
  float ms = 0.f;
  GPBO_HIP(ctx, hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  (void)hipFree(dA); (void)hipFree(dB); (void)hipFree(dC);
  const double flops = 2.0 * m * (double)n * k * (lower_only ? 0.5 : 1.0);
  out[0] = ms / iters;
  out[1] = flops / (ms / iters * 1e-3) / 1e12;
  return rc;
}

int gpbo_debug_select(gpbo_ctx* ctx, const double* ys, int64_t M, int k, int variant, int iters, int64_t* idx_out, double* val_out,
                      int64_t* first_nan_out, float* ms_out) {
  if (!ctx) return GPBO_ERR_INVALID;
  if (!ys || !idx_out || !val_out || M < 1 || k < 1 || k > GPBO_MAX_SEEDS || (variant != 1 && variant != 2) || iters < 0)
    GPBO_FAIL(ctx, GPBO_ERR_INVALID, "debug_select: bad arguments");
  GPBO_HIP(ctx, hipSetDevice(ctx->device));
  return debug_select(ctx, ys, M, k, variant, iters, idx_out, val_out, first_nan_out, ms_out);
}

int gpbo_debug_latency_probe(gpbo_ctx* ctx, long long* out, int n) {
  if (!ctx || !out || n < 1) return GPBO_ERR_INVALID;
  GPBO_HIP(ctx, hipSetDevice(ctx->device));
  return run_latency_probe(ctx, out, n);
}

#endif  // GPBO_DEBUG

int gpbo_mfma_f64_peak(gpbo_ctx* ctx, int iters, double* tflops) {
  if (!ctx || !tflops || iters < 1) return GPBO_ERR_INVALID;
  GPBO_HIP(ctx, hipSetDevice(ctx->device));
  return run_mfma_peak(ctx, iters, tflops);
}

int gpbo_mfma_f64_probe(gpbo_ctx* ctx, int iters, int waves_per_simd, int mode, double* out) {
  if (!ctx || !out || iters < 1 || waves_per_simd < 1 || waves_per_simd > 8) return GPBO_ERR_INVALID;
  GPBO_HIP(ctx, hipSetDevice(ctx->device));
  return run_mfma_probe(ctx, iters, waves_per_simd, mode, out);
}

#ifdef GPBO_DEBUG
int gpbo_hybrid_probe(gpbo_ctx* ctx, int iters, int cfg, double* out) {
  if (!ctx || !out || iters < 1 || cfg < 0 || cfg > 4) return GPBO_ERR_INVALID;
  GPBO_HIP(ctx, hipSetDevice(ctx->device));
  return run_hybrid_probe(ctx, iters, cfg, out);
}
#endif

int gpbo_hbm_copy_peak(gpbo_ctx* ctx, int64_t bytes, double* gbps) {
  if (!ctx || !gbps || bytes < (1 << 20)) return GPBO_ERR_INVALID;
  GPBO_HIP(ctx, hipSetDevice(ctx->device));
  return run_copy_peak(ctx, bytes, gbps);
}

}  // extern "C"
