// Posterior launcher + finalize kernel (gfx950).
//
// gpbo_posterior replaces, per candidate (SK = sklearn/gaussian_process):
//   K_trans = kernel_(X, X_train_)                     SK/_gpr.py:443   (Matern/RBF: kernels.py:1715-1724, 1556-1565)
//   y_mean  = K_trans @ alpha_                         SK/_gpr.py:444-447
//   V       = solve_triangular(L_, K_trans.T)          SK/_gpr.py:454-456   -> here V = W k*, W = L^-1
//   y_var   = 1 - einsum("ij,ji->i", V.T, V); clip; * y_std^2; sqrt      SK/_gpr.py:474-494
// and dispatches to one of four device paths:
//   M <= small_batch_limit(NP)   posterior_small.hip      batched GEMV (latency path: predicts of the host optimisers)
//   precision F32                posterior_kernel_f32.hip fp32 k* slab + v_mfma_f32_16x16x4_f32 GEMM
//   NP <= 512 (<= 2 row chunks)  posterior_kernel_v2.hip  GEN = 1: k* generated inside the MFMA kernel
//   otherwise                    posterior_kernel_v2.hip  GEN = 2: k* slab generated once + MFMA GEMM ("v3")
// The row-chunk partial sums every path writes are combined in a fixed order by posterior_finalize_kernel,
// so results are run-to-run deterministic.  (The first version of the fused kernel — 128 candidates x 256
// rows per workgroup, 2 waves/SIMD, 403 ms per C3 launch — is in the git history; docs/LAB_NOTEBOOK.md §4.1.)
#include <algorithm>
#include <cstdlib>

#include "gpbo_internal.h"
#include "fit_bodies.h"

namespace gpbo {

typedef double d4 __attribute__((ext_vector_type(4)));

// mu = y_std * (k* . alpha) + y_mean ; sd = sqrt(max(1 - sum_chunks part, 0) * y_std^2)
__global__ __launch_bounds__(256) void posterior_finalize_kernel(const double* __restrict__ part,
                                                                 const double* __restrict__ mu_part,
                                                                 int nchunks, int n_mu, int64_t Mp, int64_t M,
                                                                 double y_mean, double y_std,
                                                                 double* __restrict__ mu,
                                                                 double* __restrict__ sd, int* __restrict__ negvar) {
  const int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (m >= M) return;
  double ss = 0.0;
  for (int r = 0; r < nchunks; ++r) ss += part[(int64_t)r * Mp + m];
  double mun = 0.0;
  for (int q = 0; q < n_mu; ++q) mun += mu_part[(int64_t)q * Mp + m];
  posterior_finalize_elem(ss, mun, y_mean, y_std, mu + m, sd + m, negvar);
}

int64_t kstar_slab_budget_bytes(gpbo_ctx* ctx, int64_t want_bytes_if_unlimited) {
  const char* e = getenv("GPBO_KSTAR_GB");      // read per call: the slab-loop test changes it between passes
  const double budget_gb = (e && atof(e) > 0.0) ? atof(e) : 4.0;
  int64_t budget = (int64_t)(budget_gb * 1e9);
  const int64_t want = want_bytes_if_unlimited < budget ? want_bytes_if_unlimited : budget;
  if (want <= ctx->cap_kst * 8) return budget;          // the buffer we hold is already big enough
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
    const int64_t avail = (int64_t)(((double)free_b + (double)ctx->cap_kst * 8.0) * 0.8);
    if (avail < budget) budget = avail;
  }
  return budget;
}

static int ensure_posterior_outputs(gpbo_ctx* ctx, Model& m, int64_t Mp) {
  if (Mp > m.cap_M) {
    if (m.mu) { GPBO_HIP(ctx, hipFree(m.mu)); m.mu = nullptr; }
    if (m.sd) { GPBO_HIP(ctx, hipFree(m.sd)); m.sd = nullptr; }
    m.cap_M = 0;
    GPBO_HIP(ctx, hipMalloc((void**)&m.mu, (size_t)Mp * sizeof(double)));
    GPBO_HIP(ctx, hipMalloc((void**)&m.sd, (size_t)Mp * sizeof(double)));
    m.cap_M = Mp;
  }
  return GPBO_OK;
}

// mu, sd and their gradients in the (raw) inputs for the M resident candidates (M <= 256): posterior_small.hip
int launch_posterior_grad(gpbo_ctx* ctx, Model& m, int64_t M, double y_mean, double y_std, double** dmu_dev, double** dsd_dev,
                          double** packed_dev, const double* xc_in, double* packed_out) {
  const int64_t Mp = round_up(M, POST_CANDS);
  int rc;
  if ((rc = ensure(ctx, &ctx->Xcs, &ctx->cap_Xcs, Mp * m.DP))) return rc;
  if ((rc = ensure(ctx, &ctx->mu_part, &ctx->cap_mu_part, std::max<int64_t>((int64_t)2 * M * m.d + 2 * M, Mp)))) return rc;
  if ((rc = ensure_posterior_outputs(ctx, m, Mp))) return rc;
  // xc_in (gpbo_polish_seeds): the round's points in device-visible pinned host memory, read by the scaling kernel itself
  if ((rc = launch_prescale(ctx, xc_in ? xc_in : ctx->Xc, M, m.d, m.DP, m.ls, ctx->Xcs, Mp))) return rc;
  // packed (gpbo_polish_seeds): mu and sd land right behind the gradients, [dmu | dsd | mu | sd], so that one block brings a
  // round's results back — in packed_out (device-visible pinned host memory, written by the last kernel itself) when given,
  // else in ctx->mu_part for the caller to copy; the model's own mu / sd buffers are then NOT written
  const bool packed = packed_dev || packed_out;
  double* base = packed_out ? packed_out : ctx->mu_part;
  *dmu_dev = base;
  *dsd_dev = base + M * m.d;
  double* mu_out = packed ? base + 2 * M * m.d : m.mu;
  double* sd_out = packed ? mu_out + M : m.sd;
  if (packed_dev) *packed_dev = base;
  ev_begin(ctx, T_POST_MAIN);
  rc = launch_posterior_grad_small(ctx, m, (int)M, y_mean, y_std, *dmu_dev, *dsd_dev, mu_out, sd_out);
  ev_end(ctx, T_POST_MAIN);
  if (rc) return rc;
  m.M_post = packed ? -1 : M;
  return GPBO_OK;
}

int launch_posterior(gpbo_ctx* ctx, Model& m, int64_t M, double y_mean, double y_std) {
  const int64_t Mp = round_up(M, POST_CANDS);
  const int nchunks = (int)((m.NP + POST_ROWS - 1) / POST_ROWS);
  int rc;
  if ((rc = ensure(ctx, &ctx->Xcs, &ctx->cap_Xcs, Mp * m.DP))) return rc;
  if ((rc = ensure(ctx, &ctx->part, &ctx->cap_part, (int64_t)nchunks * Mp))) return rc;
  if ((rc = ensure(ctx, &ctx->mu_part, &ctx->cap_mu_part, (int64_t)nchunks * Mp))) return rc;
  if ((rc = ensure_posterior_outputs(ctx, m, Mp))) return rc;
  // v2 = fused generation (one kernel, 256-row chunks); v3 = k* slab + GEMM; v4 = fused generation with 512-row chunks (the
  // long comment below).  Decided first: a fused kernel whose workgroups hold ALL rows of their candidates (one row chunk) takes the
  // raw candidates in and writes mu / sd itself (round 6: three launches -> one; GPBO_POST_FUSE_ENDS=0, debug build: the three).
  const char* kv = dbg_env("GPBO_POST_KERNEL");
  const bool use_f32 = (m.precision == GPBO_F32);   // fp32 slab + f32 MFMA GEMM (posterior_kernel_f32.hip)
  int path = (nchunks <= 1 || (nchunks == 2 && Mp < 8192)) ? 2 : ((m.NP >= 384 && m.NP <= 512 && Mp >= 16384) ? 4 : 3);
  // Two row chunks (256 < NP <= 512) and a batch around bayes_opt's DEFAULT n_random = 10 000 (round 6, scripts/r06_post_10k_ab.py,
  // profiles/r06_post_10k_ab.json; ms at M = 10 000 / 20 000 for N = 300, 384, 450, 512; the slab pair was the rule's choice):
  //   slab pair 0.125-0.147 / 0.16-0.19;  8-wave fused 0.092-0.113 / 0.12-0.17;  16-wave fused 0.080-0.100 / 0.14-0.18
  // 144 ... 256 candidate tiles are one 16-wave workgroup per CU in ONE round (and that kernel takes the raw candidates and writes
  // mu / sd itself); from there to 32 768 candidates the 8-wave kernel's two workgroups per CU fill the chip better than either.
  if (nchunks == 2 && Mp >= 9216 && Mp <= 16384) path = 4;
  else if (nchunks == 2 && Mp > 16384 && Mp < 32768) path = 2;
  if (kv && (kv[0] == '2' || kv[0] == '3' || (kv[0] == '4' && m.NP <= 1024))) path = kv[0] - '0';
  const bool use_v2 = path == 2, use_v4 = path == 4;
  const char* sm = dbg_env("GPBO_POST_SMALL");
  const bool small = M <= small_batch_limit(m.NP) && !(sm && sm[0] == '0');
  const char* fe = dbg_env("GPBO_POST_FUSE_ENDS");
  const bool fuse_ends = !small && !use_f32 && !(fe && fe[0] == '0') && ((use_v2 && nchunks == 1) || (use_v4 && m.NP <= 512));
  if (!fuse_ends && (rc = launch_prescale(ctx, ctx->Xc, M, m.d, m.DP, m.ls, ctx->Xcs, Mp))) return rc;
  {
    // latency path: a handful of candidates (HipGPR.predict from the host optimiser) -> batched GEMV
    if (small) {
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
  // The path rule (decided at the top).  v3 as soon as k* would be generated more than once: the fp64 VALU work of the generation runs instead of MFMAs, not
  // beside them, and the slab GEMM's loop carries no other VALU work (posterior_kernel_v2.hip).  For 384 <= NP <= 512 and a
  // batch that fills the chip, v4 (round 4): ONE 16-wave workgroup covers all rows, so k* is generated once and never
  // crosses HBM (the slab route: a 268 MB round trip and a second launch at C2).  Measured at M = 65 536 (scripts/
  // r04_post_small_np_ab.py, profiles/r04_post_small_np_ab.json; round 2: scripts/archive/r02_small_n_posterior_ab.py):
  //   NP = 512, d = 8 : v2 0.406  v3 0.374  v4 0.352 ms (0.58 / 0.62 / 0.66 of the fp64 matrix peak)      -> v4
  //   NP = 448, d = 8 : v2 0.357  v3 0.332  v4 0.327                                                      -> v4
  //   NP = 1024, d = 16: v2 1.48   v3 1.19   v4 1.31 (two 512-row chunks: k* generated 1.5 times)          -> v3
  //   NP = 768, d = 16, M = 2^18: v3 2.87, v4 4.72 (a ragged second chunk of 16 waves, half of them idle)  -> v3
  //   NP = 512, M = 8192: v2 0.073, v3 0.115, v4 0.087 (a grid of 128 workgroups does not fill the chip)   -> v2
  //   NP = 256: v2 0.12 vs v3 0.14 (one chunk: nothing is generated twice).
  // Why v4 gains only 6 % where the slab traffic and a launch go away: its floor is the GEMM at the matrix pipe's 0.95
  // (0.25 ms) + one generation of k* on the same datapath (~0.09 ms); one 1024-thread workgroup per CU also means every
  // s_barrier stalls the whole CU (the 16-wave slab kernel measured 3 % slower at C3 for the same reason).
  // GPBO_POST_KERNEL=2|3|4 forces a path (debug build: A/B runs; 4 only up to NP = 1024).
  const int n_mu = (use_f32 || path == 3) ? nchunks : 1;
  PostEnds ends{ctx->Xc, m.ls, m.d, M, y_mean, y_std, m.mu, m.sd, ctx->negvar};
  ev_begin(ctx, T_POST_MAIN);
  int part_chunks = nchunks;   // row chunks the sum-of-squares partials are split into (fp32 path, v4: 512-row chunks)
  if (use_f32) rc = launch_posterior_f32(ctx, m, Mp, nchunks, &part_chunks);
  else if (use_v2) rc = launch_posterior_v2(ctx, m, Mp, nchunks, fuse_ends ? &ends : nullptr);
  else if (use_v4) rc = launch_posterior_v4(ctx, m, Mp, &part_chunks, fuse_ends ? &ends : nullptr);
  else rc = launch_posterior_v3(ctx, m, Mp, nchunks);
  ev_end(ctx, T_POST_MAIN);
  if (rc) return rc;
  ev_begin(ctx, T_POST_FINAL);
  if (!fuse_ends)
    posterior_finalize_kernel<<<dim3((unsigned)((M + 255) / 256)), dim3(256), 0, ctx->stream>>>(
        ctx->part, ctx->mu_part, part_chunks, n_mu, Mp, M, y_mean, y_std, m.mu, m.sd, ctx->negvar);
  ev_end(ctx, T_POST_FINAL);
  GPBO_HIP(ctx, hipGetLastError());
  m.M_post = M;
  return GPBO_OK;
}

}  // namespace gpbo
