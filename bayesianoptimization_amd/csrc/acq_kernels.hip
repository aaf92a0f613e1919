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

__global__ __launch_bounds__(256) void acq_kernel(AcqDev a, int64_t M, double* __restrict__ ys) {
  const int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (m >= M) return;
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
  ys[m] = v;
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

// acq values + k selection passes enqueued on ctx->stream; SelState and picks[npass] stay in ctx->red
static int enqueue_acq_select(gpbo_ctx* ctx, const AcqArgs& a, int64_t M, int k_seeds, SelState** st_out, Key** picks_out,
                              int* npass_out) {
  int rc;
  if ((rc = ensure(ctx, &ctx->ys, &ctx->cap_ys, M))) return rc;
  const int nblocks = (int)((M + SEL_BLOCK * SEL_ITEMS - 1) / (SEL_BLOCK * SEL_ITEMS));
  const int npass = k_seeds > 0 ? k_seeds : 1;
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

  AcqDev d;
  d.acq = a.acq; d.param = a.param; d.y_max = a.y_max; d.n_constraints = a.n_constraints;
  for (int j = 0; j < GPBO_MAX_MODELS; ++j) { d.lb[j] = a.lb[j]; d.ub[j] = a.ub[j]; d.mu[j] = a.mu[j]; d.sd[j] = a.sd[j]; }
  acq_kernel<<<dim3((unsigned)((M + 255) / 256)), dim3(256), 0, ctx->stream>>>(d, M, ctx->ys);
  GPBO_HIP(ctx, hipGetLastError());

  select_block_topk_kernel<<<dim3((unsigned)nblocks), dim3(SEL_BLOCK), 0, ctx->stream>>>(ctx->ys, M, npass, partial, nan_partial);
  select_merge_topk_kernel<<<dim3(1), dim3(SEL_BLOCK), 0, ctx->stream>>>(partial, nan_partial, nblocks, npass, st, picks);
  GPBO_HIP(ctx, hipGetLastError());
  *st_out = st; *picks_out = picks; *npass_out = npass;
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
