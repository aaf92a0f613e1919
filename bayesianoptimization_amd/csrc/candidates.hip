// On-device candidate generation — a THROUGHPUT MODE, not stream-compatible with the
// reference: TargetSpace.random_sample (bayes_opt/target_space.py:565-603) draws from NumPy's MT19937
// RandomState on the host (92 ms for 2^20 x 16 on the GPU box's EPYC; index parity with the reference requires
// exactly that stream, which stays the default).  Here a counter-based Philox4x32-10 generator produces the
// (M, d) uniform matrix directly in HBM: no host sampling, no H2D copy.  Doubles are formed like NumPy's
// random_sample: (a >> 5) * 2^26 + (b >> 6) over 2^53, then lo + (hi - lo) * u.
#include "gpbo_internal.h"

namespace gpbo {

__device__ __forceinline__ void philox4x32_10(unsigned (&c)[4], unsigned k0, unsigned k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const unsigned long long p0 = (unsigned long long)0xD2511F53u * c[0];
    const unsigned long long p1 = (unsigned long long)0xCD9E8D57u * c[2];
    const unsigned n0 = (unsigned)(p1 >> 32) ^ c[1] ^ k0;
    const unsigned n1 = (unsigned)p1;
    const unsigned n2 = (unsigned)(p0 >> 32) ^ c[3] ^ k1;
    const unsigned n3 = (unsigned)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
}

// element index e = m * d + t; each Philox call yields two doubles (elements 2q, 2q+1)
__global__ __launch_bounds__(256) void generate_candidates_kernel(double* __restrict__ Xc, int64_t total, int d,
                                                                  const double* __restrict__ lohi,
                                                                  unsigned long long seed) {
  const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t e0 = 2 * q;
  if (e0 >= total) return;
  unsigned c[4] = {(unsigned)q, (unsigned)((unsigned long long)q >> 32), 0u, 0u};
  philox4x32_10(c, (unsigned)seed, (unsigned)(seed >> 32));
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int64_t e = e0 + h;
    if (e >= total) break;
    const unsigned a = c[2 * h] >> 5, b = c[2 * h + 1] >> 6;
    const double u = ((double)a * 67108864.0 + (double)b) / 9007199254740992.0;
    const int t = (int)(e % d);
    Xc[e] = lohi[t] + (lohi[GPBO_MAX_DIM + t] - lohi[t]) * u;
  }
}

__global__ void gather_rows_kernel(const double* __restrict__ Xc, int d, const int64_t* __restrict__ idx, int n,
                                   int64_t M, double* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * d) return;
  const int r = i / d, t = i - r * d;
  const int64_t m = idx[r];
  out[i] = (m >= 0 && m < M) ? Xc[m * d + t] : __longlong_as_double(0x7ff8000000000000LL);
}

// ---- input transforms of a mixed space (TargetSpace.kernel_transform, bayes_opt/target_space.py:340-347) -----------------
// IntParameter.kernel_transform = np.round (parameter.py:308-320: round half to even = rint)
__global__ __launch_bounds__(256) void round_columns_kernel(double* __restrict__ Xc, int64_t M, int d, int col0, int ncols) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= M * ncols) return;
  const int64_t r = i / ncols;
  const int c = (int)(i - r * ncols);
  double* p = Xc + r * d + col0 + c;
  *p = rint(*p);
}
// CategoricalParameter.kernel_transform (parameter.py:434-449):  res = zeros; res[:, argmax(value, axis=1)] = 1  — the
// fancy index takes the argmax of EVERY row as a list of COLUMNS, so a column is 1 in all rows as soon as it is any row's
// argmax (correct for a single row; mirrored, not fixed).  Pass 1: which columns are some row's
// argmax (np.argmax: the first maximum; a NaN counts as the maximum, the first NaN wins); pass 2: fill.
__global__ __launch_bounds__(256) void categorical_flags_kernel(const double* __restrict__ Xc, int64_t M, int d, int col0, int ncols,
                                                                unsigned* __restrict__ flags) {
  const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (r >= M) return;
  const double* p = Xc + r * d + col0;
  int best = 0;
  double bv = p[0];
  if (bv == bv)
    for (int c = 1; c < ncols; ++c) {
      const double v = p[c];
      if (v != v) { best = c; break; }
      if (v > bv) { bv = v; best = c; }
    }
  atomicOr(&flags[best >> 5], 1u << (best & 31));
}
__global__ __launch_bounds__(256) void categorical_fill_kernel(double* __restrict__ Xc, int64_t M, int d, int col0, int ncols,
                                                               const unsigned* __restrict__ flags) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= M * ncols) return;
  const int64_t r = i / ncols;
  const int c = (int)(i - r * ncols);
  Xc[r * d + col0 + c] = ((flags[c >> 5] >> (c & 31)) & 1u) ? 1.0 : 0.0;
}

}  // namespace gpbo

using namespace gpbo;

// Columns [col0, col0 + ncols) of the resident (M, d_total) candidate matrix from a host array values (M, ncols): the
// parameters a device cannot draw from the reference's stream — IntParameter / CategoricalParameter.random_sample are
// RandomState.randint (parameter.py:280-284, 360-377), masked rejection sampling whose word consumption depends on the
// values — are drawn on the host at the right position of the SAME MT19937 stream and joined to the device-generated
// float columns here (8 B per candidate and column instead of the whole matrix).
extern "C" int gpbo_set_candidate_columns(gpbo_ctx* ctx, const double* values, int64_t M, int d_total, int col0, int ncols) {
  if (!ctx) return GPBO_ERR_INVALID;
  if (!values || M < 1 || d_total < 1 || d_total > GPBO_MAX_DIM || ncols < 1 || col0 < 0 || col0 + ncols > d_total)
    GPBO_FAIL(ctx, GPBO_ERR_INVALID, "set_candidate_columns: bad arguments");
  GPBO_HIP(ctx, hipSetDevice(ctx->device));
  int rc;
  ctx->raw_valid = false;
  if ((rc = ensure(ctx, &ctx->Xc, &ctx->cap_Xc, M * d_total))) return rc;
  GPBO_HIP(ctx, hipMemcpy2DAsync(ctx->Xc + col0, (size_t)d_total * sizeof(double), values, (size_t)ncols * sizeof(double),
                                 (size_t)ncols * sizeof(double), (size_t)M, hipMemcpyHostToDevice, ctx->stream));
  GPBO_HIP(ctx, hipStreamSynchronize(ctx->stream));       // the caller's array is borrowed for the call only
  ctx->M = M;
  ctx->d_c = d_total;
  for (auto& m : ctx->models) m.M_post = -1;
  return GPBO_OK;
}

// TargetSpace.kernel_transform over the resident candidates, in place: group g covers columns [col0[g], col0[g] + ncols[g])
// with kind[g] = 0 (FloatParameter: identity), 1 (IntParameter: np.round) or 2 (CategoricalParameter: see above).  The
// untransformed matrix is kept aside: gpbo_get_candidate_rows keeps returning the rows random_sample drew (x_tries of
// bayes_opt/acquisition.py:311-317), the posterior sees the transformed ones (parameter.py:484-487).
extern "C" int gpbo_transform_candidates(gpbo_ctx* ctx, int n_groups, const int* kind, const int* col0, const int* ncols) {
  if (!ctx) return GPBO_ERR_INVALID;
  if (n_groups < 1 || n_groups > GPBO_MAX_DIM || !kind || !col0 || !ncols) GPBO_FAIL(ctx, GPBO_ERR_INVALID, "transform_candidates: bad arguments");
  if (ctx->M < 1) GPBO_FAIL(ctx, GPBO_ERR_STATE, "transform_candidates: no candidates resident");
  const int d = ctx->d_c;
  const int64_t M = ctx->M;
  bool any = false;
  for (int g = 0; g < n_groups; ++g) {
    if (kind[g] < 0 || kind[g] > 2 || ncols[g] < 1 || col0[g] < 0 || col0[g] + ncols[g] > d)
      GPBO_FAIL(ctx, GPBO_ERR_INVALID, "transform_candidates: bad group");
    any = any || kind[g] != 0;
    // The reference's categorical one-hot has BATCH behaviour (a column is set in ALL rows as soon as it is any row's argmax): the
    // flags are a reduction over the whole (M, d) batch.  A context that holds one shard of a sharded job (gpbo_comm_init with
    // world > 1, a member of a gpbo_group) would reduce over its rows only and silently transform differently from the reference.
    if (kind[g] == 2 && ctx->world > 1)
      GPBO_FAIL(ctx, GPBO_ERR_UNSUPPORTED, "transform_candidates: a categorical group needs the WHOLE reference batch resident on "
                                           "one context (this context is a shard of a multi-device job)");
  }
  GPBO_HIP(ctx, hipSetDevice(ctx->device));
  if (!any) return GPBO_OK;
  int rc;
  if (ctx->raw_valid) GPBO_FAIL(ctx, GPBO_ERR_STATE, "transform_candidates: the resident candidates are already transformed");
  if ((rc = ensure(ctx, &ctx->Xc_raw, &ctx->cap_Xc_raw, M * d))) return rc;
  GPBO_HIP(ctx, hipMemcpyAsync(ctx->Xc_raw, ctx->Xc, (size_t)M * d * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
  {
    char* p = (char*)ctx->red;
    int64_t cap = ctx->cap_red;
    if ((rc = ensure(ctx, &p, &cap, 4096))) return rc;
    ctx->red = p;
    ctx->cap_red = cap;
  }
  unsigned* flags = (unsigned*)ctx->red;     // GPBO_MAX_DIM bits per group, one 64-byte line each
  GPBO_HIP(ctx, hipMemsetAsync(flags, 0, (size_t)n_groups * 16 <= 4096 ? (size_t)n_groups * 16 : 4096, ctx->stream));
  for (int g = 0; g < n_groups; ++g) {
    if (kind[g] == 1) {
      const int64_t n = M * ncols[g];
      round_columns_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream>>>(ctx->Xc, M, d, col0[g], ncols[g]);
    } else if (kind[g] == 2) {
      unsigned* f = flags + 4 * g;          // 128 bits >= GPBO_MAX_DIM columns
      categorical_flags_kernel<<<dim3((unsigned)((M + 255) / 256)), dim3(256), 0, ctx->stream>>>(ctx->Xc_raw, M, d, col0[g], ncols[g], f);
      const int64_t n = M * ncols[g];
      categorical_fill_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream>>>(ctx->Xc, M, d, col0[g], ncols[g], f);
    }
  }
  GPBO_HIP(ctx, hipGetLastError());
  ctx->raw_valid = true;
  for (auto& m : ctx->models) m.M_post = -1;
  return GPBO_OK;
}

extern "C" int gpbo_generate_candidates(gpbo_ctx* ctx, int64_t M, int d, const double* lo, const double* hi,
                                        uint64_t seed) {
  if (!ctx) return GPBO_ERR_INVALID;
  if (!lo || !hi || M < 1 || d < 1 || d > GPBO_MAX_DIM) GPBO_FAIL(ctx, GPBO_ERR_INVALID, "generate_candidates: bad arguments");
  GPBO_HIP(ctx, hipSetDevice(ctx->device));
  int rc;
  ctx->raw_valid = false;
  if ((rc = ensure(ctx, &ctx->Xc, &ctx->cap_Xc, M * d))) return rc;
  {
    char* p = (char*)ctx->red;
    int64_t cap = ctx->cap_red;
    if ((rc = ensure(ctx, &p, &cap, (int64_t)2 * GPBO_MAX_DIM * 8 + 4096))) return rc;
    ctx->red = p;
    ctx->cap_red = cap;
  }
  double* h = (double*)((char*)ctx->pinned_aux + PIN_AUX_CAND);
  for (int t = 0; t < d; ++t) { h[t] = lo[t]; h[GPBO_MAX_DIM + t] = hi[t]; }
  GPBO_HIP(ctx, hipMemcpyAsync(ctx->red, h, 2 * GPBO_MAX_DIM * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  const int64_t total = M * d, nq = (total + 1) / 2;
  generate_candidates_kernel<<<dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, ctx->stream>>>(
      ctx->Xc, total, d, (const double*)ctx->red, (unsigned long long)seed);
  GPBO_HIP(ctx, hipGetLastError());
  GPBO_HIP(ctx, hipStreamSynchronize(ctx->stream));
  ctx->M = M;
  ctx->d_c = d;
  for (auto& m : ctx->models) m.M_post = -1;
  return GPBO_OK;
}

extern "C" int gpbo_get_candidate_rows(gpbo_ctx* ctx, const int64_t* idx, int n, double* out) {
  if (!ctx) return GPBO_ERR_INVALID;
  if (!idx || !out || n < 1 || n > 4096) GPBO_FAIL(ctx, GPBO_ERR_INVALID, "get_candidate_rows: bad arguments");
  if (ctx->M < 1) GPBO_FAIL(ctx, GPBO_ERR_STATE, "get_candidate_rows: no candidates resident");
  GPBO_HIP(ctx, hipSetDevice(ctx->device));
  const int d = ctx->d_c;
  int rc;
  {
    char* p = (char*)ctx->red;
    int64_t cap = ctx->cap_red;
    if ((rc = ensure(ctx, &p, &cap, (int64_t)n * 8 + (int64_t)n * d * 8 + 64))) return rc;
    ctx->red = p;
    ctx->cap_red = cap;
  }
  int64_t* didx = (int64_t*)ctx->red;
  double* dout = (double*)((char*)ctx->red + (((size_t)n * 8 + 63) / 64) * 64);
  GPBO_HIP(ctx, hipMemcpyAsync(didx, idx, (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream));
  // (after gpbo_transform_candidates: the rows as they were drawn, not as the kernel sees them)
  gather_rows_kernel<<<dim3((unsigned)((n * d + 255) / 256)), dim3(256), 0, ctx->stream>>>(ctx->raw_valid ? ctx->Xc_raw : ctx->Xc, d, didx, n, ctx->M, dout);
  GPBO_HIP(ctx, hipGetLastError());
  GPBO_HIP(ctx, hipMemcpyAsync(out, dout, (size_t)n * d * 8, hipMemcpyDeviceToHost, ctx->stream));
  GPBO_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return GPBO_OK;
}
