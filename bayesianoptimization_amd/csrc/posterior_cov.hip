// Full posterior covariance of a host batch on the device:
//   GaussianProcessRegressor.predict(X, return_cov=True)   sklearn/gaussian_process/_gpr.py:458-469
//     V     = solve_triangular(L_, K_trans.T)       -> V = W K*^T   (W = L^-1, one MFMA GEMM, W lower triangular)
//     y_cov = kernel_(X) - V.T @ V                  -> second MFMA GEMM (A given transposed) + the kernel of the batch
//     y_cov * y_train_std^2                         (_gpr.py:461-466; no clipping on this branch)
// called by bayes_opt through BayesianOptimization.predict(..., return_cov=True) (bayes_opt/bayesian_optimization.py:238).
// The reference path drags L_ (N x N) through LAPACK on the host; here K*^T, V and V^T V never leave HBM and only the
// M x M result crosses the boundary.  kernel_(X) follows kernels.py:1735-1738 / 1556-1565: exact unit diagonal.
#include "gpbo_internal.h"

namespace gpbo {

template <int KERNEL>
__global__ __launch_bounds__(256) void cov_finalize_kernel(const double* __restrict__ Xcs, int DP, int64_t M, int64_t ldc,
                                                           double scale, double* __restrict__ C) {
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t i = blockIdx.y;
  if (j >= M) return;
  double kv = 1.0;
  if (i != j) {
    const double* xi = Xcs + i * DP;
    const double* xj = Xcs + j * DP;
    double d2 = 0.0;
    for (int t = 0; t < DP; ++t) {
      const double df = xi[t] - xj[t];
      d2 = fma(df, df, d2);
    }
    kv = gpbo_kernel_value<KERNEL>(d2);
  }
  C[i * ldc + j] = (kv - C[i * ldc + j]) * scale;
}

// Requires ctx->Xc to hold the M raw candidates.  Scratch: ctx->kst = [K*^T (NP x Mp) | V (NP x Mp) | V^T V (Mp x Mp)].
int launch_posterior_cov(gpbo_ctx* ctx, Model& m, int64_t M, double y_std, double** cov_dev, int64_t* ld_cov) {
  const int64_t Mp = round_up(M, POST_CANDS);     // multiple of 128: GEMM tiles and the k* generator both fit
  const int nchunks = (int)((m.NP + POST_ROWS - 1) / POST_ROWS);
  int rc;
  if ((rc = ensure(ctx, &ctx->Xcs, &ctx->cap_Xcs, Mp * m.DP))) return rc;
  if ((rc = ensure(ctx, &ctx->mu_part, &ctx->cap_mu_part, (int64_t)nchunks * Mp))) return rc;
  if ((rc = ensure(ctx, &ctx->kst, &ctx->cap_kst, 2 * m.NP * Mp + Mp * Mp))) return rc;
  if ((rc = launch_prescale(ctx, ctx->Xc, M, m.d, m.DP, m.ls, ctx->Xcs, Mp))) return rc;
  double* Kt = ctx->kst;                 // [NP][Mp]   K*^T (train-point major)
  double* V = Kt + m.NP * Mp;            // [NP][Mp]
  double* C = V + m.NP * Mp;             // [Mp][Mp]
  if ((rc = launch_kstar_slab(ctx, m, Kt, Mp, Mp, 0, nchunks))) return rc;
  // rows N..NP-1 belong to the zero padding of the training set (W is the identity there): they must not reach V
  if (m.NP > m.N)
    GPBO_HIP(ctx, hipMemsetAsync(Kt + m.N * Mp, 0, (size_t)(m.NP - m.N) * Mp * sizeof(double), ctx->stream));
  GemmArgs g{};    // V = W K*^T ; W lower triangular: the k-loop stops at the row tile's diagonal
  g.m = (int)m.NP; g.n = (int)Mp; g.k = (int)m.NP; g.alpha = 1.0; g.beta = 0.0;
  g.A = m.W; g.lda = m.NP; g.a_lower = 1;
  g.B = Kt; g.ldb = Mp;
  g.C = V; g.ldc = Mp; g.batch = 1;
  if ((rc = launch_gemm(ctx, g))) return rc;
  GemmArgs h{};    // C = V^T V
  h.m = (int)Mp; h.n = (int)Mp; h.k = (int)m.NP; h.alpha = 1.0; h.beta = 0.0;
  h.A = V; h.lda = Mp; h.a_trans = 1;
  h.B = V; h.ldb = Mp;
  h.C = C; h.ldc = Mp; h.batch = 1;
  if ((rc = launch_gemm(ctx, h))) return rc;
  const dim3 grid((unsigned)((M + 255) / 256), (unsigned)M);
  const double scale = y_std * y_std;
  if (m.kernel == GPBO_KERNEL_MATERN25)
    cov_finalize_kernel<GPBO_KERNEL_MATERN25><<<grid, dim3(256), 0, ctx->stream>>>(ctx->Xcs, m.DP, M, Mp, scale, C);
  else
    cov_finalize_kernel<GPBO_KERNEL_RBF><<<grid, dim3(256), 0, ctx->stream>>>(ctx->Xcs, m.DP, M, Mp, scale, C);
  GPBO_HIP(ctx, hipGetLastError());
  *cov_dev = C;
  *ld_cov = Mp;
  return GPBO_OK;
}

}  // namespace gpbo
