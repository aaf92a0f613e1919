// Fused posterior kernel (gfx950): k* tile generation + V = W k* on v_mfma_f64_16x16x4_f64 +
// column sum of squares, never materialising k* (M x N) or V (N x M) in HBM.
//
// Replaces, per candidate (SK = sklearn/gaussian_process):
//   K_trans = kernel_(X, X_train_)                     SK/_gpr.py:443   (Matern/RBF: kernels.py:1715-1724, 1556-1565)
//   y_mean  = K_trans @ alpha_                         SK/_gpr.py:444-447
//   V       = solve_triangular(L_, K_trans.T)          SK/_gpr.py:454-456   -> here V = W k*, W = L^-1
//   y_var   = 1 - einsum("ij,ji->i", V.T, V); clip; * y_std^2; sqrt      SK/_gpr.py:474-494
//
// Work decomposition.  The contraction is a triangular GEMM  V[N x M] = W[N x N] * K*^T[N x M]
// (N^2/2 MACs per candidate).  A workgroup (8 waves) owns POST_CANDS = 128 candidates and a chunk
// of POST_ROWS = 256 rows of W; wave w owns 32 rows (2 x 8 MFMA tiles = 128 accumulator VGPRs).
// It walks k (train points) in stages of 16: all 8 waves cooperatively generate the 16 x 128 k*
// stage tile into LDS (VALU: distance, sqrt, exp), then every wave multiplies its 32 x 16 slice of W
// — streamed straight from HBM/L2 into registers in MFMA-fragment order (pack_w_kernel), no LDS —
// with the tile.  k* is regenerated once per row chunk that needs it (N/512 times on average): the
// VALU cost is ~1/3 of the MFMA time and overlaps with it; nothing but 8 bytes per (chunk,
// candidate) leaves the chip.  The row-chunk partial sums are combined in a fixed order by
// posterior_finalize_kernel, so results are run-to-run deterministic.
#include "gpbo_internal.h"

#include <cstdlib>

namespace gpbo {

typedef double d4 __attribute__((ext_vector_type(4)));

struct PostArgs {
  const double* Wp;     // packed W (see pack_w_kernel)
  const double* Xs;     // [NP][DP] scaled train points
  const double* alpha;  // [NP]
  const double* Xcs;    // [Mp][DP] scaled candidates
  double* part;         // [nchunks][Mp]
  double* mu_part;      // [Mp]
  int NP;
  int64_t Mp;
  int nchunks;
  int n_ctiles;
};

template <int KERNEL>
__device__ __forceinline__ double kernel_value_post(double d2) {
  if (KERNEL == GPBO_KERNEL_MATERN25) {
    double k = sqrt(d2) * 2.23606797749978969641;
    return (1.0 + k + k * k / 3.0) * exp(-k);
  } else {
    return exp(-0.5 * d2);
  }
}

template <int DP, int KERNEL, bool XC_LDS, int SCHED>
__global__ __launch_bounds__(512, 2) void posterior_kernel(PostArgs p) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* Ks = smem;                                       // [2][POST_BK][KS_STRIDE]
  double* Xl = smem + 2 * POST_BK * KS_STRIDE;             // [DP][128] (XC_LDS only)

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int bid = blockIdx.x;
  const int r = p.nchunks - 1 - bid / p.n_ctiles;   // heaviest row chunks are dispatched first
  const int ct = bid - (bid / p.n_ctiles) * p.n_ctiles;
  const bool last = (r == p.nchunks - 1);
  const int NP = p.NP;
  const int k_end = min(NP, (r + 1) * POST_ROWS);
  const int n_stages = k_end / POST_BK;

  // --- generation role: candidate c of the tile, k-group kg (4 consecutive train points per stage)
  const int c = (wave & 1) * 64 + lane;
  const int kg = wave >> 1;
  const double* xcp = p.Xcs + ((int64_t)ct * POST_CANDS + c) * DP;
  double xc[XC_LDS ? 1 : DP];
  if constexpr (XC_LDS) {
#pragma unroll
    for (int t = 0; t < DP; t += 2) {
      const double2 v = *reinterpret_cast<const double2*>(xcp + t);
      if (kg == 0) {
        Xl[t * POST_CANDS + c] = v.x;
        Xl[(t + 1) * POST_CANDS + c] = v.y;
      }
    }
  } else {
#pragma unroll
    for (int t = 0; t < DP; t += 2) {
      const double2 v = *reinterpret_cast<const double2*>(xcp + t);
      xc[t] = v.x;
      xc[t + 1] = v.y;
    }
  }

  // --- MFMA role: 32 rows of W starting at slab_row0
  const int slab = r * (POST_ROWS / 32) + wave;
  const int slab_row0 = slab * 32;
  const bool active = slab_row0 < NP;   // false only in a ragged last chunk (NP not a multiple of 256)
  const int64_t pairs = NP / 8;
  // inactive waves stream a valid slab (NP/32 - 1) so the main loop stays branch-free; their sums are dropped
  const int slab_ld = active ? slab : (NP / 32 - 1);
  const double2* wp = reinterpret_cast<const double2*>(p.Wp) + (int64_t)slab_ld * pairs * 128 + lane;

  d4 acc[2][8];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[t][j] = d4{0.0, 0.0, 0.0, 0.0};
  double mu_acc = 0.0;

  // k* values of this thread's 4 train points of `stage` (pure VALU + scalar loads: no LDS access, so the
  // scheduler is free to weave it between the MFMAs); gen_store publishes them to the stage tile.
  auto gen_compute = [&](int stage, double (&kv)[4]) {
    const int j0 = stage * POST_BK + kg * 4;
    const double* xr = p.Xs + (int64_t)j0 * DP;  // wave-uniform -> scalar loads
    double d2[4] = {0.0, 0.0, 0.0, 0.0};
    if constexpr (XC_LDS) {
      // wide inputs: walk the dimensions 8 at a time (bounds the live scalar registers)
#pragma unroll 1
      for (int tb = 0; tb < DP; tb += 8) {
#pragma unroll
        for (int tt = 0; tt < 8; ++tt) {
          const double x = Xl[(tb + tt) * POST_CANDS + c];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const double df = x - xr[e * DP + tb + tt];
            d2[e] = fma(df, df, d2[e]);
          }
        }
      }
    } else {
#pragma unroll
      for (int t = 0; t < DP; ++t) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const double df = xc[t] - xr[e * DP + t];
          d2[e] = fma(df, df, d2[e]);
        }
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      kv[e] = kernel_value_post<KERNEL>(d2[e]);
      mu_acc = fma(kv[e], p.alpha[j0 + e], mu_acc);  // only the last chunk's value is stored
    }
  };
  auto gen_store = [&](const double (&kv)[4], int buf) {
#pragma unroll
    for (int e = 0; e < 4; ++e) Ks[(buf * POST_BK + kg * 4 + e) * KS_STRIDE + c] = kv[e];
  };

  double2 a_cur[2][2], a_nxt[2][2];
  auto loadA = [&](int stage, double2(&a)[2][2]) {
#pragma unroll
    for (int pp = 0; pp < 2; ++pp)
#pragma unroll
      for (int t = 0; t < 2; ++t) a[pp][t] = wp[((int64_t)(stage * 2 + pp) * 2 + t) * 64];
  };

  if (XC_LDS) __syncthreads();
  {
    double kv0[4];
    gen_compute(0, kv0);
    gen_store(kv0, 0);
  }
  loadA(0, a_cur);
  __syncthreads();

  auto mma_step = [&](int buf, int q) {
    const double a0 = (q & 1) ? a_cur[q >> 1][0].y : a_cur[q >> 1][0].x;
    const double a1 = (q & 1) ? a_cur[q >> 1][1].y : a_cur[q >> 1][1].x;
    const double* kb = Ks + (buf * POST_BK + q * 4 + (lane >> 4)) * KS_STRIDE + (lane & 15);
#pragma unroll
    for (int jt = 0; jt < 8; ++jt) {
      const double b = kb[jt * 16];
      acc[0][jt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b, acc[0][jt], 0, 0, 0);
      acc[1][jt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b, acc[1][jt], 0, 0, 0);
    }
  };

  // Loop 1: stages strictly left of this chunk's diagonal block — every wave multiplies, stage s+1 always
  // exists; the body is one basic block so the scheduler interleaves the VALU generation with the MFMAs.
  const int n_full = r * (POST_ROWS / POST_BK);
  int s = 0;
  for (; s < n_full; ++s) {
    const int buf = s & 1;
    double kv[4];
    loadA(s + 1, a_nxt);
    gen_compute(s + 1, kv);
    mma_step(buf, 0);
    mma_step(buf, 1);
    mma_step(buf, 2);
    mma_step(buf, 3);
    gen_store(kv, buf ^ 1);   // after the last read of this iteration: no LDS ordering against the MFMA feeds
    if constexpr (SCHED > 0) {
      // ask the scheduler for "1 MFMA, then SCHED VALU" so the generation hides in the MFMA shadow
#pragma unroll
      for (int i = 0; i < 64; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, SCHED, 0);
      }
    }
#pragma unroll
    for (int pp = 0; pp < 2; ++pp)
#pragma unroll
      for (int t = 0; t < 2; ++t) a_cur[pp][t] = a_nxt[pp][t];
    __syncthreads();
  }
  // Loop 2: the diagonal block of the chunk (<= 16 stages): a wave stops multiplying once the stage lies
  // right of its 32-row slab (W is lower triangular there).
  for (; s < n_stages; ++s) {
    const int buf = s & 1;
    const bool has_next = (s + 1 < n_stages);
    const bool domma = (s * POST_BK <= slab_row0 + 31);
    const bool domma_next = has_next && ((s + 1) * POST_BK <= slab_row0 + 31);
    if (domma_next) loadA(s + 1, a_nxt);
    if (has_next) {
      double kv[4];
      gen_compute(s + 1, kv);
      gen_store(kv, buf ^ 1);
    }
    if (domma) {
#pragma unroll
      for (int q = 0; q < 4; ++q) mma_step(buf, q);
    }
#pragma unroll
    for (int pp = 0; pp < 2; ++pp)
#pragma unroll
      for (int t = 0; t < 2; ++t) a_cur[pp][t] = a_nxt[pp][t];
    __syncthreads();
  }

  // --- epilogue: column sums of squares over this chunk's rows, fixed reduction order
  double* red = Ks;                              // [8][128]
  double* mured = Ks + 8 * POST_CANDS;           // [4][128]
#pragma unroll
  for (int jt = 0; jt < 8; ++jt) {
    double s = 0.0;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) s = fma(acc[t][jt][rr], acc[t][jt][rr], s);
    s += __shfl_xor(s, 16);
    s += __shfl_xor(s, 32);
    if (lane < 16) red[wave * POST_CANDS + jt * 16 + lane] = active ? s : 0.0;
  }
  if (last) mured[kg * POST_CANDS + c] = mu_acc;
  __syncthreads();
  if (tid < POST_CANDS) {
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < 8; ++w) s += red[w * POST_CANDS + tid];
    const int64_t m = (int64_t)ct * POST_CANDS + tid;
    p.part[(int64_t)r * p.Mp + m] = s;
    if (last) p.mu_part[m] = ((mured[tid] + mured[POST_CANDS + tid]) + mured[2 * POST_CANDS + tid]) +
                             mured[3 * POST_CANDS + tid];
  }
}

// mu = y_std * (k* . alpha) + y_mean ; sd = sqrt(max(1 - sum_chunks part, 0) * y_std^2)
__global__ __launch_bounds__(256) void posterior_finalize_kernel(const double* __restrict__ part,
                                                                 const double* __restrict__ mu_part,
                                                                 int nchunks, int n_mu, int64_t Mp, int64_t M,
                                                                 double y_mean, double y_std,
                                                                 double* __restrict__ mu,
                                                                 double* __restrict__ sd) {
  const int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (m >= M) return;
  double ss = 0.0;
  for (int r = 0; r < nchunks; ++r) ss += part[(int64_t)r * Mp + m];
  double var = 1.0 - ss;
  if (var < 0.0) var = 0.0;          // _gpr.py:479-485 (NaN stays NaN, as in numpy)
  var = var * (y_std * y_std);
  sd[m] = sqrt(var);
  double mun = 0.0;
  for (int q = 0; q < n_mu; ++q) mun += mu_part[(int64_t)q * Mp + m];
  mu[m] = y_std * mun + y_mean;
}

// Scheduling variant of the main loop: 0 = compiler default, 6 = "1 MFMA : 6 VALU" group hints (default).
// GPBO_POST_SCHED=0 selects the unhinted build for A/B runs.
static int post_sched_variant() {
  const char* e = getenv("GPBO_POST_SCHED");   // read per launch so one process can A/B both builds
  return (e && e[0] == '0') ? 0 : 6;
}

template <int DP, int KERNEL, bool XC_LDS, int SCHED>
static int launch_post_s(gpbo_ctx* ctx, const PostArgs& a, int64_t nblocks) {
  size_t lds = (size_t)(2 * POST_BK * KS_STRIDE + (XC_LDS ? DP * POST_CANDS : 0)) * sizeof(double);
  auto kern = posterior_kernel<DP, KERNEL, XC_LDS, SCHED>;
  if (lds > 64 * 1024) {
    GPBO_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  }
  kern<<<dim3((unsigned)nblocks), dim3(512), lds, ctx->stream>>>(a);
  GPBO_HIP(ctx, hipGetLastError());
  return GPBO_OK;
}

template <int DP, int KERNEL, bool XC_LDS>
static int launch_post_t(gpbo_ctx* ctx, const PostArgs& a, int64_t nblocks) {
  if (post_sched_variant() == 0) return launch_post_s<DP, KERNEL, XC_LDS, 0>(ctx, a, nblocks);
  return launch_post_s<DP, KERNEL, XC_LDS, 6>(ctx, a, nblocks);
}

template <int KERNEL>
static int launch_post_k(gpbo_ctx* ctx, int DP, const PostArgs& a, int64_t nblocks) {
  switch (DP) {
    case 4: return launch_post_t<4, KERNEL, false>(ctx, a, nblocks);
    case 8: return launch_post_t<8, KERNEL, false>(ctx, a, nblocks);
    case 16: return launch_post_t<16, KERNEL, false>(ctx, a, nblocks);
    case 32: return launch_post_t<32, KERNEL, true>(ctx, a, nblocks);
    case 64: return launch_post_t<64, KERNEL, true>(ctx, a, nblocks);
  }
  GPBO_FAIL(ctx, GPBO_ERR_UNSUPPORTED, "posterior: unsupported padded dimension");
}

int launch_posterior(gpbo_ctx* ctx, Model& m, int64_t M, double y_mean, double y_std) {
  const int64_t Mp = round_up(M, POST_CANDS);
  const int nchunks = (int)((m.NP + POST_ROWS - 1) / POST_ROWS);
  const int64_t n_ctiles = Mp / POST_CANDS;
  int rc;
  if ((rc = ensure(ctx, &ctx->Xcs, &ctx->cap_Xcs, Mp * m.DP))) return rc;
  if ((rc = ensure(ctx, &ctx->part, &ctx->cap_part, (int64_t)nchunks * Mp))) return rc;
  if ((rc = ensure(ctx, &ctx->mu_part, &ctx->cap_mu_part, (int64_t)nchunks * Mp))) return rc;
  if (Mp > m.cap_M) {
    if (m.mu) { GPBO_HIP(ctx, hipFree(m.mu)); m.mu = nullptr; }
    if (m.sd) { GPBO_HIP(ctx, hipFree(m.sd)); m.sd = nullptr; }
    m.cap_M = 0;
    GPBO_HIP(ctx, hipMalloc((void**)&m.mu, (size_t)Mp * sizeof(double)));
    GPBO_HIP(ctx, hipMalloc((void**)&m.sd, (size_t)Mp * sizeof(double)));
    m.cap_M = Mp;
  }
  if ((rc = launch_prescale(ctx, ctx->Xc, M, m.d, m.DP, m.ls, ctx->Xcs, Mp))) return rc;
  {
    // latency path: a handful of candidates (HipGPR.predict from the host optimiser) -> batched GEMV
    const char* sm = getenv("GPBO_POST_SMALL");
    if (M <= 72 && !(sm && sm[0] == '0')) {
      ev_begin(ctx, T_POST_MAIN);
      rc = launch_posterior_small(ctx, m, (int)M, y_mean, y_std);
      ev_end(ctx, T_POST_MAIN);
      if (rc) return rc;
      ev_begin(ctx, T_POST_FINAL);
      ev_end(ctx, T_POST_FINAL);
      m.M_post = M;
      return GPBO_OK;
    }
  }
  // GPBO_POST_KERNEL=1 selects the 2-waves/SIMD kernel of this file (kept for A/B); default is v2
  // (posterior_kernel_v2.hip, 4 waves/SIMD).
  const char* kv = getenv("GPBO_POST_KERNEL");
  const bool use_v1 = kv && kv[0] == '1';
  // v2 = fused generation (one kernel); v3 = k* slab + GEMM.  Default: v3 once k* would be regenerated by
  // >= 3 row chunks (NP > 512), v2 below that (the second launch costs more than the regeneration saves).
  const bool use_v2 = kv ? (kv[0] == '2') : (nchunks <= 2);
  const bool use_f32 = (m.precision == GPBO_F32);   // fp32 slab + f32 MFMA GEMM (posterior_kernel_f32.hip)
  const int n_mu = use_f32 ? nchunks : ((use_v1 || use_v2) ? 1 : nchunks);
  ev_begin(ctx, T_POST_MAIN);
  if (use_f32) {
    rc = launch_posterior_f32(ctx, m, Mp, nchunks);
  } else if (use_v1) {
    PostArgs a;
    a.Wp = m.Wp; a.Xs = m.Xs; a.alpha = m.alpha; a.Xcs = ctx->Xcs; a.part = ctx->part;
    a.mu_part = ctx->mu_part; a.NP = (int)m.NP; a.Mp = Mp; a.nchunks = nchunks; a.n_ctiles = (int)n_ctiles;
    const int64_t nblocks = n_ctiles * nchunks;
    if (nblocks > 0x7fffffffLL) GPBO_FAIL(ctx, GPBO_ERR_UNSUPPORTED, "posterior: grid too large; shard the candidates");
    if (m.kernel == GPBO_KERNEL_MATERN25) rc = launch_post_k<GPBO_KERNEL_MATERN25>(ctx, m.DP, a, nblocks);
    else rc = launch_post_k<GPBO_KERNEL_RBF>(ctx, m.DP, a, nblocks);
  } else if (use_v2) {
    rc = launch_posterior_v2(ctx, m, Mp, nchunks);
  } else {
    rc = launch_posterior_v3(ctx, m, Mp, nchunks);
  }
  ev_end(ctx, T_POST_MAIN);
  if (rc) return rc;
  ev_begin(ctx, T_POST_FINAL);
  posterior_finalize_kernel<<<dim3((unsigned)((M + 255) / 256)), dim3(256), 0, ctx->stream>>>(
      ctx->part, ctx->mu_part, nchunks, n_mu, Mp, M, y_mean, y_std, m.mu, m.sd);
  ev_end(ctx, T_POST_FINAL);
  GPBO_HIP(ctx, hipGetLastError());
  m.M_post = M;
  return GPBO_OK;
}

}  // namespace gpbo
