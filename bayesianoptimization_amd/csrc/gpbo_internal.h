// Internal declarations shared by the HIP translation units of libgpbo (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "gpbo.h"

namespace gpbo {

constexpr int NB = 64;          // Cholesky / inverse block size (one MFMA GEMM tile edge)
constexpr int POST_ROWS = 256;  // W rows owned by one posterior workgroup (8 waves x 32 rows)
constexpr int POST_CANDS = 128; // candidate padding granule (Mp = round_up(M, 128); kernels tile 64 candidates)
constexpr int POST_BK = 16;     // train points (k) per LDS stage of the fp64 kernels

enum TimingSlot {
  T_FIT = 0, T_POST_MAIN = 1, T_POST_FINAL = 2, T_ACQ = 3, T_KMAT = 4, T_CHOL = 5, T_TRTRI = 6, T_COUNT = 8
};

struct Model {
  bool fitted = false;
  bool wp_packed = false;  // the fused small-problem fit has already written Wp (finish_enqueue skips pack_w_kernel)
  bool wt_valid = false;   // K holds W^T for the slot's current fit (polish_fused.hip: the row walk of the local searches with W in memory)
  int64_t N = 0, NP = 0;   // observations, padded to a multiple of NB
  int d = 0, DP = 0;       // input dimension, padded to {4,8,16,32,64}
  int kernel = 0;
  int precision = 0;
  double noise = 0.0;      // value added to the diagonal of K at the last fit
  int64_t cap_NP = 0;      // allocated capacity (NP) of the square buffers
  int cap_DP = 0;
  double* ls = nullptr;    // [GPBO_MAX_DIM] length scale per dimension (device)
  double* Xs = nullptr;    // [NP][DP] train points / length_scale, zero padded
  double* K = nullptr;     // [NP][NP] kernel matrix + noise (lower triangle valid)
  double* L = nullptr;     // [NP][NP] Cholesky factor (lower triangle valid)
  double* W = nullptr;     // [NP][NP] L^-1 (upper triangle zero)
  double* Wp = nullptr;    // NP*NP doubles, MFMA-fragment packed W for the posterior kernel
  float* Wp32 = nullptr;   // NP*NP floats, W rounded to fp32 in v_mfma_f32_16x16x4_f32 fragment order (precision F32)
  int64_t cap_Wp32 = 0;
  double* dinv = nullptr;  // [NP/NB][NB][NB] inverses of the diagonal blocks of L
  double* tmp = nullptr;   // NP*NP/2 doubles workspace (trtri)
  double* yn = nullptr;    // [NP] normalised targets, zero padded
  double* tvec = nullptr;  // [NP] W*y
  double* alpha = nullptr; // [NP]
  // posterior outputs for the resident candidates
  double* mu = nullptr;
  double* sd = nullptr;
  int64_t cap_M = 0;
  int64_t M_post = -1;     // number of candidates mu/sd are valid for (-1: none)
};

// gpbo_lml_batch, per lane group it has run: the problem shape and, from the second run on, the captured launch sequence.  A pool
// looked up by the whole key: the lanes of a call are dealt to groups by their number (gpbo_api.hip), and a theta search's rounds
// come with 6, 6, 5, 3, 2, 1, 1 ... live runs — every grouping it passes through keeps its graphs.
struct LmlLane {
  hipGraphExec_t exec = nullptr;
  bool seen = false;
  int64_t N = 0;
  int d = 0, kernel = 0, n_ls = 0, eval_gradient = 0, lanes = 0, group = 0;
  double noise = 0.0;
  const void* X = nullptr; const void* y = nullptr; const void* K = nullptr;
  uint64_t used = 0;      // the pool's clock at its last use (the least recently used entry is the one replaced)
};
constexpr int LML_GRAPH_POOL = 24;

struct EventPair {
  hipEvent_t a = nullptr, b = nullptr;
  bool used = false;
};

// Cholesky look-ahead (chol_kernels.hip): the bulk stream that carries the far part of a rank-512 trailing update while
// the next outer panel's chain runs on `main`, and the events the two streams hand each other.  One per main stream
// (the context's own, a slot's, an LML lane's), created on first use.
struct LookAhead {
  hipStream_t main = nullptr, bulk = nullptr;
  std::vector<hipEvent_t> ev;
};

}  // namespace gpbo

struct gpbo_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  std::string err;
  gpbo::Model models[GPBO_MAX_MODELS];
  // gpbo_lml_batch: its stream (lml_stream[0]); the lanes' buffers live in lml_slab
  hipStream_t lml_stream[GPBO_LML_BATCH_MAX] = {};
  int lane_streams_picked = 0;   // how many of them were chosen to run side by side (pick_lane_streams, gpbo_api.hip)
  bool no_timing = false;   // batch lanes do not touch the timing events
  bool timing_off = false;  // gpbo_set_timing(ctx, 0): no event records at all (a record is a marker packet on the stream)
  bool no_lookahead = false;   // several lane streams in flight: the Cholesky look-ahead would only add streams to a full chip
  // gpbo_fit_begin / gpbo_fit_wait: a slot's fit enqueued on the slot's own stream, its staging words in pinned window
  // 1 + slot (the windows gpbo_lml_batch uses for its groups — it waits for pending fits first), its pivot word in info_slots
  hipStream_t slot_stream[GPBO_MAX_MODELS] = {};
  int* info_slots = nullptr;                       // device: GPBO_MAX_MODELS potrf info words
  int* pending_info[GPBO_MAX_MODELS] = {};         // host (pinned) word a pending fit's pivot check lands in; null = none
  // Lane mode (gpbo_lml_batch): every fit/LML launcher runs its kernel for `lanes` models at once; model l's buffers
  // sit l * lane_stride doubles behind the ones of the Model passed in (one slab, same layout per lane)
  int lanes = 1;
  int64_t lane_stride = 0;
  double* lml_slab = nullptr; int64_t cap_lml_slab = 0;
  gpbo::LmlLane lml_lane[gpbo::LML_GRAPH_POOL];
  uint64_t lml_lane_clock = 0;
  double* lml_X = nullptr; int64_t cap_lml_X = 0;   // the batch's raw inputs, uploaded once per call
  double* lml_y = nullptr; int64_t cap_lml_y = 0;
  int64_t lml_N = 0; int lml_d = 0;                 // shape of the resident inputs (0: none)
  bool lml_graph_off = false;   // stream capture failed once: direct launches only
  // candidates
  double* Xc = nullptr;    // [M][d] raw
  int64_t cap_Xc = 0;      // capacity in doubles
  int64_t M = 0;
  int d_c = 0;
  double* Xc_raw = nullptr;   // [M][d] the candidates BEFORE gpbo_transform_candidates (what gpbo_get_candidate_rows returns then)
  int64_t cap_Xc_raw = 0;
  bool raw_valid = false;     // Xc holds kernel_transform(Xc_raw); cleared by everything that rewrites Xc
  double* stage = nullptr; // [d][M] stream-order image of device-generated candidates (mt19937.hip)
  int64_t cap_stage = 0;
  unsigned* mt_work = nullptr;   // MT19937 jump-ahead work area: returned state | 34-block stretch | sub-stream states
  int64_t cap_mt_work = 0;
  uint16_t* mt_bits = nullptr;   // jump polynomial table on the device (set-bit lists), its offsets, its stride and size
  int* mt_offset = nullptr;
  int64_t mt_table_stride = -1;
  int mt_table_count = 0;
  void* mt_desc = nullptr;       // sub-stream descriptors + polynomial indices of the current call (device)
  int64_t cap_mt_desc = 0;
  double* Xcs = nullptr;   // [Mp][DP] scaled/padded workspace
  int64_t cap_Xcs = 0;
  double* part = nullptr;  // [nchunks][Mp] partial |W k*|^2
  int64_t cap_part = 0;
  double* mu_part = nullptr;  // [nchunks][Mp] (v3: one partial mean per 256-train-point chunk) or [Mp]
  int64_t cap_mu_part = 0;
  double* kst = nullptr;   // materialised k* slab [NP][slab width] (posterior v3)
  int64_t cap_kst = 0;
  double* ys = nullptr;    // [M] negated acquisition values
  int64_t cap_ys = 0;
  void* red = nullptr;     // reduction scratch
  int64_t cap_red = 0;
  int* info_dev = nullptr; // potrf info word
  void* pinned = nullptr;  // pinned host staging: window 0 = fit/LML words (PIN_* below), windows 1..8 = gpbo_lml_batch groups
  void* pinned_base = nullptr;      // the allocation `pinned` points into (never re-pointed) and the address the device sees it at:
  char* pinned_base_dev = nullptr;  // fused_small.hip reads length scales from it and writes pivot word / LML scalars into it
  void* fused_stage = nullptr;      // pinned: X / y of a small host-side fit per window (FUSED_STAGE_BYTES each), read by the fused kernel
  char* fused_stage_dev = nullptr;
  void* pinned_aux = nullptr;   // last window of the same allocation: selection / candidate staging (PIN_AUX_*); never re-pointed
  int* negvar = nullptr;        // device-visible address of the PIN_AUX_NEGVAR word
  // small batches (the host optimisers' rounds: tens to hundreds of points per call, hundreds of calls per suggest):
  // candidates and results cross PCIe through this pinned block instead of the caller's pageable arrays — the runtime
  // stages pageable copies through its own buffers and blocks on them, ~15-20 us per copy
  void* small_pinned = nullptr;          // SMALL_PIN_BYTES: [candidates in | mu out | sd out]
  void* polish_pinned = nullptr;         // gpbo_polish_seeds: [the round's points | per model [dmu | dsd | mu | sd] coming back]
  char* polish_pinned_dev = nullptr;     // ... as the device sees it: the round's kernels read / write it directly (no copy nodes)
  int64_t cap_polish_pinned = 0;
  hipEvent_t small_ev = nullptr;         // the last H2D out of small_pinned has completed
  bool small_ev_pending = false;
  // hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a per-DEVICE setting: one bit per kernel family, per context (a
  // process-wide flag would leave every device but the first of a gpbo_group at the 64 KiB default)
  unsigned func_attrs = 0;
  std::vector<gpbo::LookAhead> lookahead;
  gpbo::EventPair ev[gpbo::T_COUNT];
  // RCCL
  void* comm = nullptr;          // ncclComm_t; exchanged atomically by whoever aborts it (comm.hip: abort_comm)
  bool comm_lost = false;        // the communicator was aborted after a collective failed or timed out
  bool debug_fail_next_acq = false;   // debug build: gpbo_debug_fail_next_acq armed (the field exists in both builds: one layout)
  int world = 1, rank = 0;
  void* comm_buf = nullptr;      // device: [send records | gathered records | 8-byte reduction word]
  int64_t cap_comm_buf = 0;
  void* comm_host = nullptr;     // pinned: gathered records / reduction word
  int64_t cap_comm_host = 0;
};

namespace gpbo {

constexpr size_t SMALL_PIN_IN = 128 * 1024, SMALL_PIN_OUT = 32 * 1024;   // bytes: candidates in; mu, sd out (each)
constexpr size_t SMALL_PIN_BYTES = SMALL_PIN_IN + 2 * SMALL_PIN_OUT;
constexpr unsigned ATTR_GEMM128 = 4u, ATTR_CHOL128 = 16u, ATTR_FUSED = 32u, ATTR_MID = 64u, ATTR_KINV_GRAD = 128u, ATTR_POLISH_FUSED = 256u, ATTR_GEMM_FAT = 512u, ATTR_CHOL_FUSED = 1024u;
// fused_small.hip: the whole fit / LML evaluation of a problem of NP <= fused_max_np() as one launch of one workgroup per model
constexpr int FUSED_NP_DEFAULT = 64, FUSED_NP_CAP = 512;
// mid_fit.hip: fused_max_np() < NP <= mid_max_np(): the strip algorithms, ~15 launches
constexpr int MID_NP_DEFAULT = 768, MID_NP_CAP = 1024;
// the local searches of gpbo_polish_seeds as one launch (polish_fused.hip): up to this padded size, one model — the kernel's own
// limit: at N = 512 the launch still beats the lockstep rounds (profiles/r06_polish_fused_ab.json: 0.36-0.48 against 0.53-0.59 ms for
// 8-10 evaluations, 2.09 against 2.13 for 48)
constexpr int POLISH_FUSED_NP_DEFAULT = 512;
// pinned staging of a small host-side fit's X (N, d) | y (N), read by the first kernel directly (one window per PIN window)
constexpr int STAGE_NP_CAP = MID_NP_CAP;
static_assert(STAGE_NP_CAP >= FUSED_NP_CAP, "the staging window serves both small paths");
constexpr size_t FUSED_STAGE_BYTES = ((size_t)STAGE_NP_CAP * GPBO_MAX_DIM + STAGE_NP_CAP) * sizeof(double);

// ---- pinned host staging layout -------------------------------------------------------------------------------
// ONE allocation of PIN_WINDOWS windows of PIN_WINDOW bytes.  Window 0 (ctx->pinned) carries the words of a fit /
// LML evaluation; gpbo_lml_batch re-points ctx->pinned at windows 1..GPBO_LML_BATCH_MAX for its groups (their own
// sub-layout, PIN_LANE_* in gpbo_api.hip); the LAST window (ctx->pinned_aux) belongs to the selection and candidate
// entry points, so that no two subsystems share a byte whatever stays in flight.
constexpr size_t PIN_WINDOW = 16384;
constexpr int PIN_WINDOWS = 2 + GPBO_LML_BATCH_MAX;
// window 0
constexpr size_t PIN_LS = 0;                                         // [GPBO_MAX_DIM] doubles: length scales
constexpr size_t PIN_LS_BYTES = GPBO_MAX_DIM * sizeof(double);
constexpr size_t PIN_INFO = 1024;                                    // potrf info word
constexpr size_t PIN_LML_OUT = 2048;                                 // yT alpha, sum log L_ii, gradient[GPBO_MAX_DIM]
constexpr size_t PIN_LML_OUT_BYTES = (2 + GPBO_MAX_DIM) * sizeof(double);
static_assert(PIN_LS + PIN_LS_BYTES <= PIN_INFO, "length scales overlap the info word");
static_assert(PIN_INFO + sizeof(int) <= PIN_LML_OUT, "info word overlaps the LML scalars");
static_assert(PIN_LML_OUT + PIN_LML_OUT_BYTES <= PIN_WINDOW, "LML scalars leave the window");
// aux window
constexpr size_t PIN_AUX_SEL_OUT = 256;                              // SelState + picks[GPBO_MAX_SEEDS + 1] coming back
constexpr size_t PIN_AUX_SEL_OUT_BYTES = 32 + 16 * (GPBO_MAX_SEEDS + 1);
constexpr size_t PIN_AUX_CAND = 2048;                                // [lo | hi (or hi - lo)][GPBO_MAX_DIM] doubles | MT19937 key[624]
constexpr size_t PIN_AUX_CAND_BYTES = 2 * GPBO_MAX_DIM * sizeof(double) + 624 * sizeof(uint32_t);
static_assert(PIN_AUX_SEL_OUT + PIN_AUX_SEL_OUT_BYTES <= PIN_AUX_CAND, "selection results overlap the candidate staging");
constexpr size_t PIN_AUX_NEGVAR = 8192;                              // int: a finalize kernel clipped a NEGATIVE variance (_gpr.py:479-485)
static_assert(PIN_AUX_CAND + PIN_AUX_CAND_BYTES <= PIN_AUX_NEGVAR, "candidate staging overlaps the clipped-variance flag");
static_assert(PIN_AUX_NEGVAR + sizeof(int) <= PIN_WINDOW, "clipped-variance flag leaves the window");

void set_global_error(const std::string& s);

// Environment policy.  The product library reads exactly four variables, none of which changes a result:
//   GPBO_KSTAR_GB (k* slab workspace budget), GPBO_COMM_TIMEOUT_S, GPBO_GROUP_TIMEOUT_S (deadlines of the multi-GPU
//   failure path) and GPBO_GROUP_HOST_MERGE (a device group merges its shards' records on the host instead of over RCCL).
// Everything else — A/B switches between kernel variants, dispatch thresholds, probes, fault injection — goes through
// dbg_env(), which is getenv() only in the -DGPBO_DEBUG build (libgpbo_dbg.so: the tests' and scripts' library) and a
// constant NULL in the product (tests/test_abi.py greps the sources for getenv against that allow-list).
#ifdef GPBO_DEBUG
inline const char* dbg_env(const char* name) { return getenv(name); }
#else
inline const char* dbg_env(const char*) { return nullptr; }
#endif

#define GPBO_HIP(ctx, expr)                                                                  \
  do {                                                                                       \
    hipError_t _e = (expr);                                                                  \
    if (_e != hipSuccess) {                                                                  \
      char _b[512];                                                                          \
      snprintf(_b, sizeof(_b), "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),        \
               __FILE__, __LINE__);                                                          \
      if (ctx) (ctx)->err = _b;                                                              \
      gpbo::set_global_error(_b);                                                            \
      return GPBO_ERR_HIP;                                                                   \
    }                                                                                        \
  } while (0)

#define GPBO_FAIL(ctx, code, msg)                                                            \
  do {                                                                                       \
    if (ctx) (ctx)->err = (msg);                                                             \
    gpbo::set_global_error(msg);                                                             \
    return (code);                                                                           \
  } while (0)

// k(d2) for the posterior-side k* generation (device only).  Same formulas as sklearn (kernels.py:1722-1724,
// 1559-1560) with two cost cuts that stay within ~1 ulp: sqrt via v_rsq_f64 + Goldschmidt/Newton without the
// subnormal rescaling (d2 is a sum of squares of O(1) numbers; exact 0 handled), and K^2/3 as K^2 * (1/3).
// The fit-side kernel matrix (kmat_kernel) uses the same function, so K and k* share one arithmetic.
#ifdef __HIPCC__
__device__ __forceinline__ double gpbo_sqrt_pos(double x) {
  const double y = __builtin_amdgcn_rsq(x);
  double g = x * y, h = 0.5 * y;
  double r = fma(-h, g, 0.5);
  g = fma(g, r, g);
  h = fma(h, r, h);
  r = fma(-h, g, 0.5);
  g = fma(g, r, g);
  h = fma(h, r, h);
  g = fma(fma(-g, g, x), h, g);
  return x > 0.0 ? g : x;   // 0 -> 0 (rsq(0) = inf would give NaN), NaN -> NaN
}
// exp(x) for x <= 0 (the only arguments a stationary kernel has): n = rint(x / ln 2), r = x - n ln 2 in two pieces (the high one
// has 32 significant bits, so n * ln2_hi and the first difference are exact for |n| < 2^11), exp(r) by the degree-13 Taylor
// polynomial (|r| <= 0.347: truncation 4e-18), result = ldexp(p, n) — v_ldexp_f64 rounds into the subnormals and to 0 by itself,
// x = 0 gives exactly 1, NaN stays NaN.  22 instructions against ~40 for the library's exp (which also serves x > 0, overflow
// and the errno-style cases); < 1 ulp (tests/test_gpu_parity.py::test_kernel_value_exp_against_numpy).  Round 4.
// Round 5: arguments below -800 are taken as -800 first (the result is an exact 0 from -745.2 on): for x = -inf the
// reduction was inf - inf = NaN (an RBF entry of points 1e160 apart: NumPy gives 0), and beyond |x| ~ 1e40 the reduced argument
// was garbage and the polynomial overflowed.  The comparison is false for NaN, which therefore still propagates.
__device__ __forceinline__ double gpbo_exp_nonpos(double x) {
  x = (x < -800.0) ? -800.0 : x;
  const double n = __builtin_rint(x * 1.44269504088896338700e+00);
  double r = fma(-n, 6.93147180369123816490e-01, x);       // ln2_hi = 0x3FE62E42FEE00000
  r = fma(-n, 1.90821492927058770002e-10, r);              // ln2_lo
  double p = 1.6059043836821613e-10;                        // 1/13!
  p = fma(p, r, 2.08767569878680989792e-09);                // 1/12!
  p = fma(p, r, 2.50521083854417187751e-08);                // 1/11!
  p = fma(p, r, 2.75573192239858906526e-07);                // 1/10!
  p = fma(p, r, 2.75573192239858906526e-06);                // 1/9!
  p = fma(p, r, 2.48015873015873015873e-05);                // 1/8!
  p = fma(p, r, 1.98412698412698412698e-04);                // 1/7!
  p = fma(p, r, 1.38888888888888888889e-03);                // 1/6!
  p = fma(p, r, 8.33333333333333333333e-03);                // 1/5!
  p = fma(p, r, 4.16666666666666666667e-02);                // 1/4!
  p = fma(p, r, 1.66666666666666666667e-01);                // 1/3!
  p = fma(p, r, 0.5);
  p = fma(p, r, 1.0);
  p = fma(p, r, 1.0);
  // n is an integer-valued double or NaN; beyond int range the conversion saturates, which is still "result 0"
  return __builtin_amdgcn_ldexp(p, (int)fmax(n, -2147483000.0));
}
template <int KERNEL>
__device__ __forceinline__ double gpbo_kernel_value(double d2) {
  if (KERNEL == GPBO_KERNEL_MATERN25) {
    const double k = gpbo_sqrt_pos(d2) * 2.23606797749978969641;
    return (1.0 + k + (k * k) * 0.33333333333333333333) * gpbo_exp_nonpos(-k);
  } else {
    return gpbo_exp_nonpos(-0.5 * d2);
  }
}
#endif

inline int64_t round_up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }
inline int pad_dim(int d) { return d <= 4 ? 4 : d <= 8 ? 8 : d <= 16 ? 16 : d <= 32 ? 32 : 64; }

template <typename T>
int ensure(gpbo_ctx* ctx, T** p, int64_t* cap, int64_t need) {
  if (need <= *cap && *p) return GPBO_OK;
  if (*p) {
    GPBO_HIP(ctx, hipFree(*p));
    *p = nullptr;
    *cap = 0;
  }
  void* q = nullptr;
  GPBO_HIP(ctx, hipMalloc(&q, (size_t)need * sizeof(T)));
  *p = (T*)q;
  *cap = need;
  return GPBO_OK;
}

// k* slab budget in bytes: GPBO_KSTAR_GB (default 4), clipped to 80 % of what the device could give the slab —
// hipMemGetInfo is asked only when the slab buffer would have to grow (it costs tens of microseconds per call).
int64_t kstar_slab_budget_bytes(gpbo_ctx* ctx, int64_t want_bytes_if_unlimited);   // posterior_kernel.hip

// ---- launchers implemented in the kernel translation units ---------------------------------
// fit_kernels.hip
int launch_prescale(gpbo_ctx* ctx, const double* X, int64_t n, int d, int DP, const double* ls,
                    double* out, int64_t n_pad);
int launch_kmat(gpbo_ctx* ctx, Model& m, double noise, double* out);   // out: m.K, or m.L (factorised in place)
int launch_fill_w_diag(gpbo_ctx* ctx, Model& m, bool zero_fill = true);
int launch_trmv(gpbo_ctx* ctx, Model& m);
int launch_append_row(gpbo_ctx* ctx, Model& m, int64_t j);   // row j (== current m.N) from the prescaled m.Xs[j]
int launch_pack_w(gpbo_ctx* ctx, Model& m);
struct GemmArgs {
  int m, n, k;            // multiples of 64 / 64 / 16
  double alpha, beta;
  const double* A; int64_t lda; int64_t strideA;
  const double* B; int64_t ldb; int64_t strideB;
  double* C; int64_t ldc; int64_t strideC;
  int lanes = 1; int64_t lane_stride = 0;   // filled in by launch_gemm from the context (lane mode)
  int batch;
  int b_trans;            // B given as (n,k) row-major
  int lower_only;         // skip output tiles strictly above the diagonal (m == n)
  int a_lower;            // A is lower triangular (k == m): k-loop stops at the row tile's diagonal
  int b_lower;            // B is lower triangular (k == n, not transposed): k-loop starts at the column tile
  int a_trans;            // A given as (k,m) row-major (C = A^T B)
  int k_from_tile;        // k-loop starts at max(row tile, column tile): W^T W with W lower triangular
  int tri_grid;           // set by launch_gemm: blockIdx.x = linear lower-triangle index of the output tile (lower_only products)
  int skip00;             // leave the leading skip00 x skip00 output tiles alone (64x64-tile kernel only): the diagonal-block
                          // workgroup of the same launch (or an earlier launch) owns them
  int fat;                // 0: launch_gemm decides from this product's shape whether the sixteen-wave tile kernel takes it (another
                          // summation order over k, i.e. other bits); +1 / -1: the caller has decided — a product that one code path
                          // launches whole and another in pieces (the Cholesky's trailing updates) must come out the same either way
};
int launch_gemm(gpbo_ctx* ctx, const GemmArgs& g);
// launch_gemm's own rule for g.fat == 0 (exposed for callers that split a product)
bool gemm_fat_rule(const GemmArgs& g);
// chol_kernels.hip: blocked Cholesky of m.L in place + inverted 64x64 diagonal blocks (128-column steps, `outer`-column panels);
// stamps (device, >= 8 words, may be null): in-kernel clocks of the first diagonal workgroup
int launch_cholesky128(gpbo_ctx* ctx, Model& m, int outer, long long* stamps);
// fused_small.hip: the fit (mode 0: ... + packed W) or an LML evaluation (1: value, 2: value + gradient) of m (and, in lane mode, of
// ctx->lanes models) as ONE launch; src 0: raw X / y / ls_in given (device-visible), 1: m.Xs / m.yn / m.ls resident.  The pivot word
// and the LML scalars land in the device-visible host words info_out / out (pitches per lane, in ints / doubles).
int fused_max_np();
int launch_fused_small(gpbo_ctx* ctx, Model& m, int mode, int src, int n_ls, const double* X, const double* y, const double* ls_in,
                       double* scal, int* info_out, int64_t info_pitch, double* out, int64_t out_pitch);
// mid_fit.hip: the strip path's launches (lane-aware through ctx->lanes / lane_stride)
int mid_max_np();
int launch_mid_inputs(gpbo_ctx* ctx, Model& m, const double* X, const double* y, const double* ls_in);
int launch_kmat_q(gpbo_ctx* ctx, Model& m, double noise, double* out);
int launch_w_strip(gpbo_ctx* ctx, Model& m, bool pack);                                   // W, the strips' share of W y (m.tmp), Wp if pack
int launch_alpha_strip(gpbo_ctx* ctx, Model& m, int* info_out, int64_t info_pitch);     // alpha; pivot word -> info_out (device-visible host word)
// posterior_kernel.hip
int launch_posterior(gpbo_ctx* ctx, Model& m, int64_t M, double y_mean, double y_std);
int launch_posterior_grad(gpbo_ctx* ctx, Model& m, int64_t M, double y_mean, double y_std, double** dmu_dev, double** dsd_dev,
                          double** packed_dev = nullptr,    // packed: [dmu | dsd | mu | sd] contiguous (mu / sd not in the model's buffers)
                          const double* xc_in = nullptr,    // the points (M, d), device-visible, instead of the resident candidate set
                          double* packed_out = nullptr);    // where the packed block goes (device-visible) instead of ctx->mu_part
// posterior_kernel_v2.hip
// Both ends of a posterior pass inside the fused kernel's launch (round 6), when ONE workgroup owns every row of its candidates
// (one row chunk): the candidate tile is scaled on its way into LDS (prescale_elem's division: no prescale launch, no scaled
// image in memory) and the epilogue writes mu and sd itself (posterior_finalize_elem: no partials, no finalize launch).
struct PostEnds {
  const double* Xc;     // raw candidates [M][d]
  const double* ls;     // length scales [d]
  int d;
  int64_t M;
  double y_mean, y_std;
  double* mu;
  double* sd;
  int* negvar;
};
int launch_posterior_v2(gpbo_ctx* ctx, Model& m, int64_t Mp, int nchunks, const PostEnds* ends = nullptr);
int launch_posterior_v3(gpbo_ctx* ctx, Model& m, int64_t Mp, int nchunks);
int launch_posterior_v4(gpbo_ctx* ctx, Model& m, int64_t Mp, int* part_chunks, const PostEnds* ends = nullptr);   // fused, 512-row chunks (NP <= 1024)
int launch_kstar_slab(gpbo_ctx* ctx, Model& m, double* Kst, int64_t ldk, int64_t Mp, int64_t m0, int nchunks);
// posterior_cov.hip
int launch_posterior_cov(gpbo_ctx* ctx, Model& m, int64_t M, double y_std, double** cov_dev, int64_t* ld_cov);
// posterior_kernel_f32.hip
int launch_pack_w32(gpbo_ctx* ctx, Model& m);
int launch_posterior_f32(gpbo_ctx* ctx, Model& m, int64_t Mp, int nchunks, int* part_chunks);
// acq_kernels.hip
struct AcqArgs {
  int acq; double param; double y_max; int n_constraints;
  double lb[GPBO_MAX_MODELS]; double ub[GPBO_MAX_MODELS];
  const double* mu[GPBO_MAX_MODELS]; const double* sd[GPBO_MAX_MODELS];
};
int launch_acq_argbest(gpbo_ctx* ctx, const AcqArgs& a, int64_t M, int k_seeds, int64_t offset,
                       int64_t* best_idx, double* best_val, int64_t* seed_idx, double* seed_val);
// (value, global index) records of one shard, as they travel between GPUs: record 0 = the arg-best (value NaN: "my
// first NaN sits at this index"), records 1..k = argsort[:k] (index -1 = padding)
struct BestRecord { double v; int64_t i; };
// acq + selection enqueued on ctx->stream, the 1 + k_seeds records left ON THE DEVICE at `records_dev`
int launch_acq_records(gpbo_ctx* ctx, const AcqArgs& a, int64_t M, int k_seeds, int64_t offset, BestRecord* records_dev);
// gpbo_debug_select: the selection launches alone over caller-supplied values (variant 1: k passes, 2: threshold + ranks), timed
int debug_select(gpbo_ctx* ctx, const double* ys_host, int64_t M, int k, int variant, int iters, int64_t* idx_out, double* val_out,
                 int64_t* first_nan_out, float* ms_out);
// the reference's argmin / min / argsort[:k] over the union of `world` shards from their records (host; identical on every rank)
void merge_records(const BestRecord* all, int world, int k_seeds, int64_t* best_idx, double* best_val, int64_t* seed_idx,
                   double* seed_val);
// gpbo_api.hip: the theta search's raw inputs made resident on ctx's device (gpbo_lml_batch with X == NULL reads them)
int lml_upload_inputs(gpbo_ctx* ctx, const double* X, const double* y_norm, int64_t N, int d);
// gpbo_api.hip: argument checks + posterior pointers of gpbo_acq_argbest, shared with the multi-GPU entry points
int build_acq_args(gpbo_ctx* ctx, const char* who, int acq, double acq_param, double y_max, int n_constraints,
                   const double* lb, const double* ub, int k_seeds, const void* best_idx, const void* best_val,
                   const void* seed_idx, const void* seed_val, AcqArgs* out);
// posterior_small.hip
int launch_posterior_small(gpbo_ctx* ctx, Model& m, int M, double y_mean, double y_std);
int small_batch_limit(int64_t NP);   // largest M the GEMV path takes (posterior_small.hip)
// polish_fused.hip: gpbo_polish_seeds' runs as one launch (one workgroup per run).  host_block / dev_block: the two addresses of one
// device-visible pinned block of polish_fused_pinned_bytes(n_seeds, d); results land there (layout: polish_fused.hip).  eval_repeat = R > 0:
// no search, R evaluations at every seed (the debug entry's timing and parity seam).
int polish_fused_max_np();
bool polish_fused_serves(const Model& m);
size_t polish_fused_pinned_bytes(int n_seeds, int d);
int launch_polish_fused(gpbo_ctx* ctx, Model& m, int acq, double acq_param, double y_max, double y_mean, double y_std, const double* seeds,
                        int n_seeds, const double* box_lo, const double* box_hi, int max_iter, int eval_repeat, double* host_block,
                        double* dev_block);
int launch_posterior_grad_small(gpbo_ctx* ctx, Model& m, int M, double y_mean, double y_std, double* dmu_dev, double* dsd_dev,
                                double* mu_out, double* sd_out);
// lml_kernels.hip
// (out / grad: lane l's words l * out_pitch doubles behind lane 0's — device memory or device-visible pinned host words)
int launch_lml_terms(gpbo_ctx* ctx, Model& m, double* out2, int64_t out_pitch);
int launch_lml_grad(gpbo_ctx* ctx, Model& m, int n_ls, const double* Kinv, double* partial, double* out, int64_t out_pitch,
                    bool with_terms);   // out[2..] = gradient; with_terms: out[0], out[1] too (launch_lml_terms' job, same arithmetic)
// comm.hip: ncclCommCount of the context's communicator (0: none, -1: not answered)
int comm_nranks(gpbo_ctx* ctx);
// mt_jump.hip: states_dev[w] = block 1 + poly_idx[w] * stride_blocks of the MT19937 sequence whose block 0 is key_dev
int mt_jump_states(gpbo_ctx* ctx, const unsigned* key_dev, int64_t stride_blocks, int max_k, const int* poly_idx_dev,
                   int n_states, unsigned* seq_dev, unsigned* states_dev, unsigned* windows_dev);
// probe.hip
int run_mfma_peak(gpbo_ctx* ctx, int iters, double* tflops);
int run_copy_peak(gpbo_ctx* ctx, int64_t bytes, double* gbps);
int run_mfma_probe(gpbo_ctx* ctx, int iters, int waves_per_simd, int mode, double* out4);
#ifdef GPBO_DEBUG
int run_hybrid_probe(gpbo_ctx* ctx, int iters, int cfg, double* out3);
// latency_probe.hip (debug build only: the one kernel of the library that uses scratch)
int run_latency_probe(gpbo_ctx* ctx, long long* out_host, int n);
#endif

inline void ev_begin(gpbo_ctx* ctx, int slot) {
  if (ctx->no_timing || ctx->timing_off) return;
  EventPair& e = ctx->ev[slot];
  if (!e.a) { (void)hipEventCreate(&e.a); (void)hipEventCreate(&e.b); }
  (void)hipEventRecord(e.a, ctx->stream);
  e.used = false;
}
inline void ev_end(gpbo_ctx* ctx, int slot) {
  if (ctx->no_timing || ctx->timing_off) return;
  EventPair& e = ctx->ev[slot];
  (void)hipEventRecord(e.b, ctx->stream);
  e.used = true;
}

}  // namespace gpbo
