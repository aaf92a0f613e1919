// Acquisition values + arg-best / top-k selection over the resident candidates (gfx950).
// Compiled with -ffp-contract=off: the elementwise formulas follow the reference operation by
// operation (no FMA contraction), so UCB/EI/POI values differ from NumPy only through mu/sd.
//
// Replaces (reference paths relative to /root/reference):
//   -1 * base_acq(mean, std) [* p_constraints]      bayes_opt/acquisition.py:198-217
//   UCB  mean + kappa*std                            bayes_opt/acquisition.py:485
//   POI  norm.cdf((mean - y_max - xi)/std)           bayes_opt/acquisition.py:660-661
//   EI   a*norm.cdf(z) + std*norm.pdf(z)             bayes_opt/acquisition.py:847-849
//   p_c  prod_j norm(mu_j,sd_j).cdf(ub_j) - cdf(lb_j) bayes_opt/constraint.py:199-221
//   ys.argmin(), ys.min(), argsort(ys)[:k]           bayes_opt/acquisition.py:313-317
// norm.cdf = scipy.special.ndtr (Cephes: 0.5+0.5*erf(x/sqrt2) for |x/sqrt2| < sqrt(.5), else
// 0.5*erfc(|x|/sqrt2) reflected); norm.pdf = exp(-x^2/2)/sqrt(2 pi) (scipy _continuous_distns.py:360-369).
#include "gpbo_internal.h"

#include <cmath>
#include <limits>
#include <vector>

namespace gpbo {

__device__ __forceinline__ double ndtr_dev(double a) {
  if (a != a) return a;
  const double x = a * 0.70710678118654752440;  // a * sqrt(1/2)
  const double z = fabs(x);
  double y;
  if (z < 0.70710678118654752440) {
    y = 0.5 + 0.5 * erf(x);
  } else {
    y = 0.5 * erfc(z);
    if (x > 0) y = 1.0 - y;
  }
  return y;
}

__device__ __forceinline__ double norm_pdf_dev(double x) {
  return exp(-(x * x) / 2.0) / 2.50662827463100050242;  // sqrt(2*pi)
}

// scipy.stats.norm(loc, scale).cdf(b): NaN unless scale > 0
__device__ __forceinline__ double cdf_loc_scale(double b, double loc, double scale) {
  if (!(scale > 0.0)) return std::numeric_limits<double>::quiet_NaN();
  return ndtr_dev((b - loc) / scale);
}

struct AcqDev {
  int acq;
  double param, y_max;
  int n_constraints;
  double lb[GPBO_MAX_MODELS], ub[GPBO_MAX_MODELS];
  const double* mu[GPBO_MAX_MODELS];
  const double* sd[GPBO_MAX_MODELS];
};

// -1 * base_acq(mean, std) [* p_constraints] of candidate m (acquisition.py:198-217), from the models' resident mu / sd
__device__ __forceinline__ double acq_value(const AcqDev& a, const int64_t m) {
  const double mean = a.mu[0][m], sd = a.sd[0][m];
  double base;
  if (a.acq == GPBO_ACQ_UCB) {
    base = mean + a.param * sd;
  } else {
    const double aa = mean - a.y_max - a.param;
    const double z = aa / sd;
    if (a.acq == GPBO_ACQ_EI) base = aa * ndtr_dev(z) + sd * norm_pdf_dev(z);
    else base = ndtr_dev(z);
  }
  double v = -1.0 * base;
  if (a.n_constraints > 0) {
    double p = 1.0;
    for (int j = 0; j < a.n_constraints; ++j) {
      const double cm = a.mu[j + 1][m], cs = a.sd[j + 1][m];
      const double pl = (a.lb[j] != -INFINITY) ? cdf_loc_scale(a.lb[j], cm, cs) : 0.0;
      const double pu = (a.ub[j] != INFINITY) ? cdf_loc_scale(a.ub[j], cm, cs) : 1.0;
      const double pj = pu - pl;
      p = (j == 0) ? pj : p * pj;
    }
    v = v * p;
  }
  return v;
}

__global__ __launch_bounds__(256) void acq_kernel(AcqDev a, int64_t M, double* __restrict__ ys) {
  const int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (m >= M) return;
  ys[m] = acq_value(a, m);
}

// ------------------------------------------------------------------------------------------------
// Selection.  Sort key = (value, index) with NaN greater than everything (NumPy sorts NaN last) and
// -0.0 == 0.0; ties break to the lowest index.  One pass finds the smallest key strictly greater
// than the previous pick, so k passes yield argsort(ys)[:k] deterministically.
struct Key {
  double v;
  int64_t i;
};

__device__ __forceinline__ bool key_less(const Key& a, const Key& b) {
  const bool an = a.v != a.v, bn = b.v != b.v;
  if (an != bn) return bn;            // non-NaN < NaN
  if (!an && a.v != b.v) return a.v < b.v;
  return a.i < b.i;
}

struct SelState {     // lives in device memory, carried from pass to pass
  Key prev;           // last pick (i = -1 before the first pass)
  int64_t first_nan;  // lowest index holding NaN (INT64_MAX if none)
};

constexpr int SEL_BLOCK = 256;
constexpr int SEL_ITEMS = 16;   // elements per thread per block sweep

__device__ __forceinline__ Key key_min(const Key& a, const Key& b) { return key_less(b, a) ? b : a; }

__device__ Key block_reduce_key(Key k, Key* sh) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    Key o;
    o.v = __shfl_xor(k.v, off);
    o.i = __shfl_xor(k.i, off);
    k = key_min(k, o);
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) sh[wave] = k;
  __syncthreads();
  if (threadIdx.x == 0) {
    Key r = sh[0];
    for (int w = 1; w < SEL_BLOCK / 64; ++w) r = key_min(r, sh[w]);
    sh[0] = r;
  }
  __syncthreads();
  Key r = sh[0];
  __syncthreads();
  return r;
}

// Selection = argmin, min and argsort(ys)[:k] (acquisition.py:313-317) in TWO launches (round 1: two launches per pick).
// select_block_topk_kernel: every workgroup keeps its SEL_BLOCK * SEL_ITEMS values in registers and extracts its own k
// smallest keys, in order, by k block reductions; select_merge_topk_kernel: one workgroup extracts the k smallest of the
// nblocks * k survivors the same way.  The order is key_less throughout — lexicographic (value, index), NaNs last — so the
// result is the one k global passes give.  Empty slots carry the sentinel {NaN, INT64_MAX}, which sorts after every key.
__global__ __launch_bounds__(SEL_BLOCK) void select_block_topk_kernel(const double* __restrict__ ys, int64_t M, int k,
                                                                      Key* __restrict__ partial,
                                                                      int64_t* __restrict__ nan_partial) {
  __shared__ Key sh[SEL_BLOCK / 64];
  __shared__ int64_t shn[SEL_BLOCK / 64];
  const int64_t base = (int64_t)blockIdx.x * (SEL_BLOCK * SEL_ITEMS) + threadIdx.x;
  double v[SEL_ITEMS];
  int64_t fn = INT64_MAX;
#pragma unroll
  for (int it = 0; it < SEL_ITEMS; ++it) {
    const int64_t m = base + (int64_t)it * SEL_BLOCK;
    v[it] = (m < M) ? ys[m] : 0.0;
    if (m < M && v[it] != v[it] && m < fn) fn = m;
  }
  Key prev;
  prev.v = 0.0; prev.i = -1;
  for (int t = 0; t < k; ++t) {
    Key best;
    best.v = std::numeric_limits<double>::quiet_NaN();
    best.i = INT64_MAX;
#pragma unroll
    for (int it = 0; it < SEL_ITEMS; ++it) {
      Key c;
      c.v = v[it];
      c.i = base + (int64_t)it * SEL_BLOCK;
      if (c.i < M && (prev.i < 0 || key_less(prev, c))) best = key_min(best, c);
    }
    const Key r = block_reduce_key(best, sh);
    if (threadIdx.x == 0) partial[(int64_t)blockIdx.x * k + t] = r;
    prev = r;     // the sentinel sorts after everything: once it is picked, every later pick is the sentinel too
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const int64_t o = __shfl_xor(fn, off);
    fn = o < fn ? o : fn;
  }
  if ((threadIdx.x & 63) == 0) shn[threadIdx.x >> 6] = fn;
  __syncthreads();
  if (threadIdx.x == 0) {
    int64_t f = shn[0];
    for (int w = 1; w < SEL_BLOCK / 64; ++w) f = shn[w] < f ? shn[w] : f;
    nan_partial[blockIdx.x] = f;
  }
}

__global__ __launch_bounds__(SEL_BLOCK) void select_merge_topk_kernel(const Key* __restrict__ partial,
                                                                      const int64_t* __restrict__ nan_partial,
                                                                      int nblocks, int k, SelState* st,
                                                                      Key* __restrict__ picks) {
  __shared__ Key sh[SEL_BLOCK / 64];
  __shared__ int64_t shn[SEL_BLOCK / 64];
  const int64_t n = (int64_t)nblocks * k;
  Key prev;
  prev.v = 0.0; prev.i = -1;
  for (int t = 0; t < k; ++t) {
    Key best;
    best.v = std::numeric_limits<double>::quiet_NaN();
    best.i = INT64_MAX;
    for (int64_t j = threadIdx.x; j < n; j += SEL_BLOCK) {
      const Key c = partial[j];
      if (c.i != INT64_MAX && (prev.i < 0 || key_less(prev, c))) best = key_min(best, c);
    }
    const Key r = block_reduce_key(best, sh);
    if (threadIdx.x == 0) picks[t] = r;
    prev = r;
  }
  int64_t fn = INT64_MAX;
  for (int b = threadIdx.x; b < nblocks; b += SEL_BLOCK) fn = nan_partial[b] < fn ? nan_partial[b] : fn;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const int64_t o = __shfl_xor(fn, off);
    fn = o < fn ? o : fn;
  }
  if ((threadIdx.x & 63) == 0) shn[threadIdx.x >> 6] = fn;
  __syncthreads();
  if (threadIdx.x == 0) {
    int64_t f = shn[0];
    for (int w = 1; w < SEL_BLOCK / 64; ++w) f = shn[w] < f ? shn[w] : f;
    st->first_nan = f;
    st->prev = prev;
  }
}

// ------------------------------------------------------------------------------------------------
// Selection, second form (the default; GPBO_SELECT_V2=0 switches back to the k passes above, gpbo_debug_select runs either).
// The k block reductions above are a dependency chain: ~4 us per pick whatever M is (40 + 15 us of a 0.84 ms step at C2,
// profiles/r02_trace_C2_kernel_stats.csv).  Here the picks of a workgroup come out together:
//   1. every thread finds the smallest key among its own items;
//   2. every wave ranks its 64 thread minima against each other (64 LDS broadcasts, no chain); the k-th smallest of one wave's
//      minima bounds the workgroup's k-th smallest key from above, and so does tau = the least of the waves' bounds;
//   3. the keys <= tau — at least k of them (the wave that set tau owns k), normally only a few more — go to an LDS list;
//   4. every listed key counts the listed keys that sort before it: that count is its place among the picks.
// Same order as key_less — (value, index), -0.0 == 0.0, NaN after every number, the empty slot {NaN, INT64_MAX} after
// every NaN — through an order-preserving 64-bit image of the value, so the picks are the ones the k passes give.  A list
// that would outgrow its LDS (values arranged so that one thread owns many of the smallest) falls back to the k passes.
struct IKey {
  unsigned long long kv;
  int64_t i;
};

__device__ __forceinline__ unsigned long long order_bits(double v) {
  if (v != v) return ~0ull;
  const double c = (v == 0.0) ? 0.0 : v;             // -0.0 and 0.0 are one value to the order
  const unsigned long long b = (unsigned long long)__double_as_longlong(c);
  return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ bool ikey_less(const IKey& a, const IKey& b) { return a.kv < b.kv || (a.kv == b.kv && a.i < b.i); }
__device__ __forceinline__ IKey ikey_sentinel() { IKey s; s.kv = ~0ull; s.i = INT64_MAX; return s; }
__device__ __forceinline__ IKey ikey_of(const Key& k) {
  IKey r;
  r.i = k.i;
  r.kv = (k.i == INT64_MAX) ? ~0ull : order_bits(k.v);
  return r;
}

constexpr int SEL2_CAP = 1024;      // listed keys per workgroup (16 KiB of LDS)

struct Sel2Shared {
  IKey tmin[SEL_BLOCK];
  IKey bound[SEL_BLOCK / 64];
  IKey list[SEL2_CAP];
  int count;
};

// steps 1-2: `mine` = this thread's smallest key (the sentinel if it has none) -> tau, the same on every thread
__device__ IKey sel2_threshold(const IKey& mine, int k, Sel2Shared& sh) {
  const int wave = threadIdx.x >> 6;
  sh.tmin[threadIdx.x] = mine;
  if ((threadIdx.x & 63) == 0) sh.bound[wave] = ikey_sentinel();
  if (threadIdx.x == 0) sh.count = 0;
  __syncthreads();
  int rank = 0;
  for (int j = 0; j < 64; ++j) rank += ikey_less(sh.tmin[wave * 64 + j], mine) ? 1 : 0;
  if (rank == k - 1 && mine.i != INT64_MAX) sh.bound[wave] = mine;      // real keys are distinct: at most one thread per wave
  __syncthreads();
  IKey tau = sh.bound[0];
  for (int w = 1; w < SEL_BLOCK / 64; ++w) {
    const IKey o = sh.bound[w];
    if (ikey_less(o, tau)) tau = o;
  }
  return tau;
}

__device__ __forceinline__ void sel2_list(const IKey& c, const IKey& tau, int cap, Sel2Shared& sh) {
  if (c.i != INT64_MAX && !ikey_less(tau, c)) {
    const int pos = atomicAdd(&sh.count, 1);
    if (pos < cap) sh.list[pos] = c;
  }
}

// step 4: out[r] = the listed key of rank r < k (value re-read from ys), the sentinel where the list is shorter than k
__device__ void sel2_emit(int n_listed, int k, const double* __restrict__ ys, Key* __restrict__ out, Sel2Shared& sh) {
  for (int j = threadIdx.x; j < n_listed; j += SEL_BLOCK) {
    const IKey c = sh.list[j];
    int rank = 0;
    for (int q = 0; q < n_listed; ++q) rank += ikey_less(sh.list[q], c) ? 1 : 0;
    if (rank < k) {
      Key r;
      r.v = ys[c.i];
      r.i = c.i;
      out[rank] = r;
    }
  }
  for (int t = n_listed + threadIdx.x; t < k; t += SEL_BLOCK) {
    Key r;
    r.v = std::numeric_limits<double>::quiet_NaN();
    r.i = INT64_MAX;
    out[t] = r;
  }
}

// ITEMS values per thread: 16 for the big batches (a chip's worth of workgroups at M = 2^20), 4 up to M = 2^18, where 16 left the
// block stage to 16 workgroups at BASELINE config 2 (M = 65 536: 24.6 us; round 6).  ACQ: the values are not read but MADE here —
// acq_value over the resident mu / sd, written to ys on the way (the separate acq_kernel launch and its pass over ys go away).
// A grid of ONE workgroup (M <= 256 ITEMS: BASELINE config 1) is its own merge: the picks and the state go straight to `picks` / `st`.
template <int ITEMS, bool ACQ>
__global__ __launch_bounds__(SEL_BLOCK) void select_block_topk_v2_kernel(double* __restrict__ ys, int64_t M, int k, int cap,
                                                                         Key* __restrict__ partial,
                                                                         int64_t* __restrict__ nan_partial, AcqDev acq, SelState* st,
                                                                         Key* __restrict__ picks) {
  __shared__ Sel2Shared sh;
  __shared__ Key shk[SEL_BLOCK / 64];
  __shared__ int64_t shn[SEL_BLOCK / 64];
  const int64_t base = (int64_t)blockIdx.x * (SEL_BLOCK * ITEMS) + threadIdx.x;
  const bool single = gridDim.x == 1;
  double v[ITEMS];
  int64_t fn = INT64_MAX;
  IKey mine = ikey_sentinel();
#pragma unroll
  for (int it = 0; it < ITEMS; ++it) {
    const int64_t m = base + (int64_t)it * SEL_BLOCK;
    if (ACQ) {
      v[it] = (m < M) ? acq_value(acq, m) : 0.0;
      if (m < M) ys[m] = v[it];
    } else {
      v[it] = (m < M) ? ys[m] : 0.0;
    }
    if (m < M) {
      if (v[it] != v[it] && m < fn) fn = m;
      IKey c;
      c.kv = order_bits(v[it]);
      c.i = m;
      if (ikey_less(c, mine)) mine = c;
    }
  }
  const IKey tau = sel2_threshold(mine, k, sh);
#pragma unroll
  for (int it = 0; it < ITEMS; ++it) {
    const int64_t m = base + (int64_t)it * SEL_BLOCK;
    IKey c;
    c.kv = order_bits(v[it]);
    c.i = (m < M) ? m : INT64_MAX;
    sel2_list(c, tau, cap, sh);
  }
  __syncthreads();      // (ACQ: this workgroup's stores to ys are also behind it — sel2_emit reads the listed values back)
  const int n_listed = sh.count;
  Key* out = single ? picks : partial + (int64_t)blockIdx.x * k;
  if (n_listed <= cap) {
    sel2_emit(n_listed, k, ys, out, sh);
  } else {
    // the k passes of select_block_topk_kernel
    Key prev;
    prev.v = 0.0; prev.i = -1;
    for (int t = 0; t < k; ++t) {
      Key best;
      best.v = std::numeric_limits<double>::quiet_NaN();
      best.i = INT64_MAX;
#pragma unroll
      for (int it = 0; it < ITEMS; ++it) {
        Key c;
        c.v = v[it];
        c.i = base + (int64_t)it * SEL_BLOCK;
        if (c.i < M && (prev.i < 0 || key_less(prev, c))) best = key_min(best, c);
      }
      const Key r = block_reduce_key(best, shk);
      if (threadIdx.x == 0) out[t] = r;
      prev = r;
    }
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const int64_t o = __shfl_xor(fn, off);
    fn = o < fn ? o : fn;
  }
  if ((threadIdx.x & 63) == 0) shn[threadIdx.x >> 6] = fn;
  __syncthreads();
  if (threadIdx.x == 0) {
    int64_t f = shn[0];
    for (int w = 1; w < SEL_BLOCK / 64; ++w) f = shn[w] < f ? shn[w] : f;
    if (single) {      // (the __syncthreads above also orders this workgroup's picks[] stores before the read below)
      st->first_nan = f;
      __threadfence();
      st->prev = picks[k - 1];
    } else {
      nan_partial[blockIdx.x] = f;
    }
  }
}

__global__ __launch_bounds__(SEL_BLOCK) void select_merge_topk_v2_kernel(const double* __restrict__ ys,
                                                                         const Key* __restrict__ partial,
                                                                         const int64_t* __restrict__ nan_partial, int nblocks, int k,
                                                                         int cap, SelState* st, Key* __restrict__ picks) {
  __shared__ Sel2Shared sh;
  __shared__ Key shk[SEL_BLOCK / 64];
  __shared__ int64_t shn[SEL_BLOCK / 64];
  const int64_t n = (int64_t)nblocks * k;
  IKey mine = ikey_sentinel();
  for (int64_t j = threadIdx.x; j < n; j += SEL_BLOCK) {
    const IKey c = ikey_of(partial[j]);
    if (ikey_less(c, mine)) mine = c;
  }
  const IKey tau = sel2_threshold(mine, k, sh);
  for (int64_t j = threadIdx.x; j < n; j += SEL_BLOCK) sel2_list(ikey_of(partial[j]), tau, cap, sh);
  __syncthreads();
  const int n_listed = sh.count;
  if (n_listed <= cap) {
    sel2_emit(n_listed, k, ys, picks, sh);
  } else {
    Key prev;
    prev.v = 0.0; prev.i = -1;
    for (int t = 0; t < k; ++t) {
      Key best;
      best.v = std::numeric_limits<double>::quiet_NaN();
      best.i = INT64_MAX;
      for (int64_t j = threadIdx.x; j < n; j += SEL_BLOCK) {
        const Key c = partial[j];
        if (c.i != INT64_MAX && (prev.i < 0 || key_less(prev, c))) best = key_min(best, c);
      }
      const Key r = block_reduce_key(best, shk);
      if (threadIdx.x == 0) picks[t] = r;
      prev = r;
    }
  }
  int64_t fn = INT64_MAX;
  for (int b = threadIdx.x; b < nblocks; b += SEL_BLOCK) fn = nan_partial[b] < fn ? nan_partial[b] : fn;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const int64_t o = __shfl_xor(fn, off);
    fn = o < fn ? o : fn;
  }
  if ((threadIdx.x & 63) == 0) shn[threadIdx.x >> 6] = fn;
  __syncthreads();      // also orders this workgroup's picks[] stores before the read below (same workgroup, global memory)
  if (threadIdx.x == 0) {
    int64_t f = shn[0];
    for (int w = 1; w < SEL_BLOCK / 64; ++w) f = shn[w] < f ? shn[w] : f;
    st->first_nan = f;
    __threadfence();
    st->prev = picks[k - 1];
  }
}

static int select_v2_cap() {      // read per call: gpbo_debug_select and the tests switch it inside one process
  const char* e = dbg_env("GPBO_SELECT_V2_CAP");
  const int c = e ? atoi(e) : SEL2_CAP;
  return c < 1 ? 1 : (c > SEL2_CAP ? SEL2_CAP : c);
}
static bool select_v2_enabled() {
  const char* e = dbg_env("GPBO_SELECT_V2");
  return e ? atoi(e) != 0 : true;
}

// the two selection launches over ctx->ys[0..M): SelState and picks[npass] at the head of ctx->red
// `acq` (v2 only): the values are made by the block stage itself from the models' mu / sd (and stored to ctx->ys)
static int enqueue_select(gpbo_ctx* ctx, int64_t M, int npass, bool v2, SelState** st_out, Key** picks_out, const AcqDev* acq = nullptr) {
  int rc;
  const char* ie = dbg_env("GPBO_SELECT_ITEMS");      // debug build: 16 = rounds 2-5 (A/B)
  const int items = v2 ? ((ie && atoi(ie) == 16) ? 16 : (M <= ((int64_t)1 << 18) ? 4 : 16)) : SEL_ITEMS;
  const int nblocks = (int)((M + SEL_BLOCK * items - 1) / (SEL_BLOCK * items));
  // scratch layout: SelState | picks[npass] | partial[nblocks][npass] | nan_partial[nblocks]
  const int64_t bytes = sizeof(SelState) + sizeof(Key) * (npass + (int64_t)nblocks * npass) + sizeof(int64_t) * nblocks + 64;
  {
    char* p = (char*)ctx->red;
    int64_t cap = ctx->cap_red;
    if ((rc = ensure(ctx, &p, &cap, bytes))) return rc;
    ctx->red = p;
    ctx->cap_red = cap;
  }
  char* base = (char*)ctx->red;
  SelState* st = (SelState*)base;
  Key* picks = (Key*)(base + sizeof(SelState));
  Key* partial = picks + npass;
  int64_t* nan_partial = (int64_t*)(partial + (int64_t)nblocks * npass);
  if (v2) {
    const int cap = select_v2_cap();
    const AcqDev none{};
    const dim3 grid((unsigned)nblocks), block(SEL_BLOCK);
    if (items == 4 && acq) select_block_topk_v2_kernel<4, true><<<grid, block, 0, ctx->stream>>>(ctx->ys, M, npass, cap, partial, nan_partial, *acq, st, picks);
    else if (items == 4) select_block_topk_v2_kernel<4, false><<<grid, block, 0, ctx->stream>>>(ctx->ys, M, npass, cap, partial, nan_partial, none, st, picks);
    else if (acq) select_block_topk_v2_kernel<16, true><<<grid, block, 0, ctx->stream>>>(ctx->ys, M, npass, cap, partial, nan_partial, *acq, st, picks);
    else select_block_topk_v2_kernel<16, false><<<grid, block, 0, ctx->stream>>>(ctx->ys, M, npass, cap, partial, nan_partial, none, st, picks);
    if (nblocks > 1)
      select_merge_topk_v2_kernel<<<dim3(1), dim3(SEL_BLOCK), 0, ctx->stream>>>(ctx->ys, partial, nan_partial, nblocks, npass, cap, st, picks);
  } else {
    select_block_topk_kernel<<<dim3((unsigned)nblocks), dim3(SEL_BLOCK), 0, ctx->stream>>>(ctx->ys, M, npass, partial, nan_partial);
    select_merge_topk_kernel<<<dim3(1), dim3(SEL_BLOCK), 0, ctx->stream>>>(partial, nan_partial, nblocks, npass, st, picks);
  }
  GPBO_HIP(ctx, hipGetLastError());
  *st_out = st; *picks_out = picks;
  return GPBO_OK;
}

// acq values + the selection enqueued on ctx->stream; SelState and picks[npass] stay in ctx->red
static int enqueue_acq_select(gpbo_ctx* ctx, const AcqArgs& a, int64_t M, int k_seeds, SelState** st_out, Key** picks_out,
                              int* npass_out) {
  int rc;
  if ((rc = ensure(ctx, &ctx->ys, &ctx->cap_ys, M))) return rc;
  const int npass = k_seeds > 0 ? k_seeds : 1;
  AcqDev d;
  d.acq = a.acq; d.param = a.param; d.y_max = a.y_max; d.n_constraints = a.n_constraints;
  for (int j = 0; j < GPBO_MAX_MODELS; ++j) { d.lb[j] = a.lb[j]; d.ub[j] = a.ub[j]; d.mu[j] = a.mu[j]; d.sd[j] = a.sd[j]; }
  // one or two picks: the passes are the shorter chain (k = 1: 11 us against 16, profiles/r03_select_probe.json)
  const bool v2 = select_v2_enabled() && npass >= 3;
  const char* fe = dbg_env("GPBO_SELECT_FUSED_ACQ");      // debug build: 0 = the separate acq_kernel launch of rounds 1-5 (A/B)
  const bool fused_acq = v2 && !(fe && fe[0] == '0');
  if (!fused_acq) {
    acq_kernel<<<dim3((unsigned)((M + 255) / 256)), dim3(256), 0, ctx->stream>>>(d, M, ctx->ys);
    GPBO_HIP(ctx, hipGetLastError());
  }
  if ((rc = enqueue_select(ctx, M, npass, v2, st_out, picks_out, fused_acq ? &d : nullptr))) return rc;
  *npass_out = npass;
  return GPBO_OK;
}

// gpbo_debug_select: the selection alone over caller-supplied values (both forms, timed) — see include/gpbo.h
int debug_select(gpbo_ctx* ctx, const double* ys_host, int64_t M, int k, int variant, int iters, int64_t* idx_out, double* val_out,
                 int64_t* first_nan_out, float* ms_out) {
  int rc;
  if ((rc = ensure(ctx, &ctx->ys, &ctx->cap_ys, M))) return rc;
  GPBO_HIP(ctx, hipMemcpyAsync(ctx->ys, ys_host, (size_t)M * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  SelState* st = nullptr; Key* picks = nullptr;
  if ((rc = enqueue_select(ctx, M, k, variant == 2, &st, &picks))) return rc;      // warm-up and the answer
  hipEvent_t e0, e1;
  GPBO_HIP(ctx, hipEventCreate(&e0));
  GPBO_HIP(ctx, hipEventCreate(&e1));
  GPBO_HIP(ctx, hipEventRecord(e0, ctx->stream));
  for (int t = 0; t < iters; ++t)
    if ((rc = enqueue_select(ctx, M, k, variant == 2, &st, &picks))) break;
  (void)hipEventRecord(e1, ctx->stream);
  std::vector<char> host(sizeof(SelState) + sizeof(Key) * (size_t)k);
  hipError_t he = hipMemcpyAsync(host.data(), st, host.size(), hipMemcpyDeviceToHost, ctx->stream);
  if (he == hipSuccess) he = hipStreamSynchronize(ctx->stream);
  float ms = 0.0f;
  if (he == hipSuccess) he = hipEventElapsedTime(&ms, e0, e1);
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  if (rc) return rc;
  GPBO_HIP(ctx, he);
  const SelState* hst = (const SelState*)host.data();
  const Key* hp = (const Key*)(host.data() + sizeof(SelState));
  for (int t = 0; t < k; ++t) { idx_out[t] = hp[t].i == INT64_MAX ? -1 : hp[t].i; val_out[t] = hp[t].v; }
  if (first_nan_out) *first_nan_out = hst->first_nan == INT64_MAX ? -1 : hst->first_nan;
  if (ms_out) *ms_out = iters > 0 ? ms / (float)iters : 0.0f;
  return GPBO_OK;
}

int launch_acq_argbest(gpbo_ctx* ctx, const AcqArgs& a, int64_t M, int k_seeds, int64_t offset,
                       int64_t* best_idx, double* best_val, int64_t* seed_idx, double* seed_val) {
  SelState* st; Key* picks; int npass;
  int rc = enqueue_acq_select(ctx, a, M, k_seeds, &st, &picks, &npass);
  if (rc) return rc;
  // results -> pinned host staging
  static_assert(sizeof(SelState) + sizeof(Key) * (GPBO_MAX_SEEDS + 1) <= PIN_AUX_SEL_OUT_BYTES, "selection results outgrew their window");
  char* hp = (char*)ctx->pinned_aux + PIN_AUX_SEL_OUT;
  GPBO_HIP(ctx, hipMemcpyAsync(hp, st, sizeof(SelState) + sizeof(Key) * npass, hipMemcpyDeviceToHost, ctx->stream));
  GPBO_HIP(ctx, hipStreamSynchronize(ctx->stream));
  const SelState* hst = (const SelState*)hp;
  const Key* hpicks = (const Key*)(hp + sizeof(SelState));
  if (hst->first_nan != INT64_MAX) {      // numpy argmin/min: the first NaN wins
    *best_idx = hst->first_nan + offset;
    *best_val = std::numeric_limits<double>::quiet_NaN();
  } else {
    *best_idx = hpicks[0].i + offset;
    *best_val = hpicks[0].v;
  }
  for (int t = 0; t < k_seeds; ++t) {
    const bool valid = hpicks[t].i != INT64_MAX;   // fewer than k candidates
    seed_idx[t] = valid ? hpicks[t].i + offset : -1;
    seed_val[t] = hpicks[t].v;
  }
  return GPBO_OK;
}

// SelState + picks -> the 1 + k records a shard contributes to the exchange (global indices), still on the device
__global__ void pack_records_kernel(const SelState* __restrict__ st, const Key* __restrict__ picks, int k_seeds,
                                    int64_t offset, BestRecord* __restrict__ out) {
  const int t = threadIdx.x;
  if (t > k_seeds) return;
  BestRecord r;
  if (t == 0) {
    if (st->first_nan != INT64_MAX) { r.v = std::numeric_limits<double>::quiet_NaN(); r.i = st->first_nan + offset; }
    else { r.v = picks[0].v; r.i = picks[0].i + offset; }
  } else {
    const Key k = picks[t - 1];
    r.v = k.v;
    r.i = (k.i != INT64_MAX) ? k.i + offset : -1;
  }
  out[t] = r;
}

int launch_acq_records(gpbo_ctx* ctx, const AcqArgs& a, int64_t M, int k_seeds, int64_t offset, BestRecord* records_dev) {
  SelState* st; Key* picks; int npass;
  int rc = enqueue_acq_select(ctx, a, M, k_seeds, &st, &picks, &npass);
  if (rc) return rc;
  pack_records_kernel<<<dim3(1), dim3(128), 0, ctx->stream>>>(st, picks, k_seeds, offset, records_dev);
  GPBO_HIP(ctx, hipGetLastError());
  return GPBO_OK;
}

// Host merge of the gathered records: first NaN overall wins the arg-best (numpy argmin), otherwise lexicographic
// (value, index) with -0.0 == 0.0; the seeds are the k smallest by the same order with NaNs last and padding dropped.
void merge_records(const BestRecord* all, int world, int k_seeds, int64_t* best_idx, double* best_val, int64_t* seed_idx,
                   double* seed_val) {
  const int stride = 1 + k_seeds;
  auto less = [](const BestRecord& a, const BestRecord& b) {
    const bool an = a.v != a.v, bn = b.v != b.v;
    if (an != bn) return bn;
    if (!an && a.v != b.v) return a.v < b.v;
    return a.i < b.i;
  };
  bool any_nan = false;
  int64_t nan_idx = INT64_MAX;
  BestRecord best{0.0, -1};
  for (int r = 0; r < world; ++r) {
    const BestRecord& b = all[(size_t)r * stride];
    if (b.v != b.v) { any_nan = true; if (b.i < nan_idx) nan_idx = b.i; }
    else if (best.i < 0 || less(b, best)) best = b;
  }
  if (any_nan) { *best_idx = nan_idx; *best_val = std::numeric_limits<double>::quiet_NaN(); }
  else { *best_idx = best.i; *best_val = best.v; }
  // k smallest of the union of the per-rank sorted lists: selection by repeated minimum (k, world tiny)
  std::vector<int> head((size_t)world, 0);   // cursor per rank
  for (int t = 0; t < k_seeds; ++t) {
    int pick = -1;
    for (int r = 0; r < world; ++r) {
      while (head[r] < k_seeds && all[(size_t)r * stride + 1 + head[r]].i < 0) ++head[r];   // padding
      if (head[r] >= k_seeds) continue;
      if (pick < 0 || less(all[(size_t)r * stride + 1 + head[r]], all[(size_t)pick * stride + 1 + head[pick]])) pick = r;
    }
    if (pick < 0) { seed_idx[t] = -1; seed_val[t] = std::numeric_limits<double>::quiet_NaN(); continue; }
    const BestRecord& s = all[(size_t)pick * stride + 1 + head[pick]];
    seed_idx[t] = s.i; seed_val[t] = s.v;
    ++head[pick];
  }
}

}  // namespace gpbo
