// Log-marginal-likelihood value and gradient pieces on the device.
//
// Replaces, per L-BFGS-B evaluation of GaussianProcessRegressor.log_marginal_likelihood(theta,
// eval_gradient=True) (sklearn/gaussian_process/_gpr.py:575-652):
//   -0.5 y^T alpha - sum(log diag L) - N/2 log(2 pi)                       _gpr.py:601-607
//   0.5 * einsum("ijl,jik->kl", alpha alpha^T - K^-1, dK/dtheta)           _gpr.py:619-645
// with dK/dlog(l) of Matern(nu=2.5) = 5/3 * D * (sqrt(5 D.sum) + 1) * exp(-sqrt(5 D.sum)) and of RBF = D * K
// (kernels.py:1764-1766, 1567-1582; D = squared scaled coordinate differences, summed for an isotropic
// length scale).  K^-1 = W^T W comes from the MFMA GEMM; this file holds the reductions.
#include "gemm_tile.h"
#include "gpbo_internal.h"
#include "lml_bodies.h"

namespace gpbo {

// out[0] = y . alpha ; out[1] = sum_i log L_ii
__global__ __launch_bounds__(256) void lml_terms_kernel(const double* __restrict__ y, const double* __restrict__ alpha,
                                                        const double* __restrict__ L, int64_t N, int64_t NP,
                                                        double* __restrict__ out, int64_t lane_stride, int64_t out_pitch) {
  __shared__ double sh[4];
  const int64_t lo = (int64_t)blockIdx.x * lane_stride;
  lml_terms_body(y + lo, alpha + lo, L + lo, N, NP, out + (int64_t)blockIdx.x * out_pitch, sh, (int)threadIdx.x, true);
}

// One workgroup per lower 64x64 tile: sum over the tile of (alpha_i alpha_j - Kinv_ij) * dK_ij/dtheta_t.
template <int KERNEL>
__global__ __launch_bounds__(256) void lml_grad_kernel(const double* __restrict__ Xs, int DP, int n_ls, int64_t N,
                                                       int64_t NP, const double* __restrict__ alpha,
                                                       const double* __restrict__ Kinv, double* __restrict__ partial,
                                                       int64_t lane_stride) {
  const int bj = blockIdx.x, bi = blockIdx.y;
  if (bj > bi) return;
  const int64_t lo = (int64_t)blockIdx.z * lane_stride;
  extern __shared__ __attribute__((aligned(16))) double lg_smem[];
  lml_grad_tile_body<KERNEL>(Xs + lo, DP, n_ls, N, NP, alpha + lo, Kinv + lo, partial + lo, bi, bj, lg_smem, (int)threadIdx.x, true);
}

// K^-1 = W^T W and the gradient tile sums in ONE launch (the strip path's evaluations, NP <= mid_max_np()): a workgroup computes its
// lower 64x64 tile of K^-1 (gemm_tile_body: the arithmetic of the stand-alone GEMM, k from the tile's diagonal block down) into LDS
// and reduces it against dK/dtheta on the spot — K^-1 never reaches memory, and one launch fewer stands on the evaluation's chain.
// Dynamic LDS: 4096 (the tile) + max(GT_LDS_DOUBLES, 2 * DP * 64 + 8) doubles.
template <int KERNEL>
__global__ __launch_bounds__(256) void kinv_grad_kernel(GemmArgs g, const double* __restrict__ Xs, int DP, int n_ls, int64_t N, int64_t NP,
                                                        const double* __restrict__ alpha, double* __restrict__ partial, int64_t lane_stride) {
  const int bj = blockIdx.x, bi = blockIdx.y;
  if (bj > bi) return;
  extern __shared__ __attribute__((aligned(16))) double kg_smem[];
  double* ctile = kg_smem;
  double* work = kg_smem + 4096;
  const int zl = (int)blockIdx.z;
  const int64_t lo = (int64_t)zl * lane_stride;
  gemm_tile_body<false, true, true>(g, bi, bj, zl, 0, work, (int)threadIdx.x, true, ctile);
  __syncthreads();
  lml_grad_tile_body<KERNEL, true>(Xs + lo, DP, n_ls, N, NP, alpha + lo, nullptr, partial + lo, bi, bj, work, (int)threadIdx.x, true, ctile);
}

// out[2 + t] = 0.5 * sum over tiles (fixed order) of partial[tile][t] for workgroup t < n_ls; workgroup n_ls (launched only when
// the caller asks for it) computes the two LML terms out[0], out[1] — the evaluation's scalars leave in ONE launch.
__global__ __launch_bounds__(256) void lml_grad_final_kernel(const double* __restrict__ partial, int64_t ntiles,
                                                             int n_ls, double* __restrict__ out, int64_t lane_stride,
                                                             int64_t out_pitch, const double* __restrict__ y,
                                                             const double* __restrict__ alpha, const double* __restrict__ L, int64_t N,
                                                             int64_t NP) {
  __shared__ double sh[4];
  const int64_t lo = (int64_t)blockIdx.y * lane_stride;
  double* o = out + (int64_t)blockIdx.y * out_pitch;
  if ((int)blockIdx.x == n_ls) {
    lml_terms_body(y + lo, alpha + lo, L + lo, N, NP, o, sh, (int)threadIdx.x, true);
    return;
  }
  lml_grad_final_body(partial + lo, ntiles, n_ls, o + 2, (int)blockIdx.x, sh, (int)threadIdx.x, true);
}

int launch_lml_terms(gpbo_ctx* ctx, Model& m, double* out2, int64_t out_pitch) {
  lml_terms_kernel<<<dim3((unsigned)ctx->lanes), dim3(256), 0, ctx->stream>>>(m.yn, m.alpha, m.L, m.N, m.NP, out2,
                                                                                ctx->lane_stride, out_pitch);
  GPBO_HIP(ctx, hipGetLastError());
  return GPBO_OK;
}

// Kinv must hold K^-1 (lower 64x64 tiles incl. full diagonal tiles); partial needs ntiles*n_ls doubles.
// out: the evaluation's scalars [y.alpha, sum log L_ii, gradient...]; with_terms: the first two are computed here as well.
// Kinv null: K^-1 = W^T W is formed tile by tile inside the gradient launch (kinv_grad_kernel).
int launch_lml_grad(gpbo_ctx* ctx, Model& m, int n_ls, const double* Kinv, double* partial, double* out, int64_t out_pitch,
                    bool with_terms) {
  const unsigned nb = (unsigned)(m.NP / 64);
  const int64_t ntiles = (int64_t)nb * (nb + 1) / 2;
  const size_t lds = (size_t)(2 * m.DP * 64 + 8) * sizeof(double);
  dim3 grid(nb, nb, (unsigned)ctx->lanes);
  const int64_t ls = ctx->lane_stride;
  if (!Kinv) {
    GemmArgs g{};
    g.m = (int)m.NP; g.n = (int)m.NP; g.k = (int)m.NP; g.alpha = 1.0; g.beta = 0.0;
    g.A = m.W; g.lda = m.NP; g.a_trans = 1;
    g.B = m.W; g.ldb = m.NP;
    g.C = nullptr; g.ldc = 64; g.batch = 1; g.lower_only = 1; g.k_from_tile = 1;
    g.lanes = ctx->lanes; g.lane_stride = ls;
    const size_t lds2 = (size_t)(4096 + std::max(GT_LDS_DOUBLES, 2 * m.DP * 64 + 8)) * sizeof(double);
    if (!(ctx->func_attrs & ATTR_KINV_GRAD)) {
      const int cap = (int)((4096 + 2 * GPBO_MAX_DIM * 64 + 8) * sizeof(double));
      GPBO_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kinv_grad_kernel<GPBO_KERNEL_MATERN25>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, cap));
      GPBO_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kinv_grad_kernel<GPBO_KERNEL_RBF>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, cap));
      ctx->func_attrs |= ATTR_KINV_GRAD;
    }
    if (m.kernel == GPBO_KERNEL_MATERN25)
      kinv_grad_kernel<GPBO_KERNEL_MATERN25><<<grid, dim3(256), lds2, ctx->stream>>>(g, m.Xs, m.DP, n_ls, m.N, m.NP, m.alpha, partial, ls);
    else
      kinv_grad_kernel<GPBO_KERNEL_RBF><<<grid, dim3(256), lds2, ctx->stream>>>(g, m.Xs, m.DP, n_ls, m.N, m.NP, m.alpha, partial, ls);
    GPBO_HIP(ctx, hipGetLastError());
  } else if (m.kernel == GPBO_KERNEL_MATERN25)
    lml_grad_kernel<GPBO_KERNEL_MATERN25><<<grid, dim3(256), lds, ctx->stream>>>(m.Xs, m.DP, n_ls, m.N, m.NP, m.alpha, Kinv, partial, ls);
  else
    lml_grad_kernel<GPBO_KERNEL_RBF><<<grid, dim3(256), lds, ctx->stream>>>(m.Xs, m.DP, n_ls, m.N, m.NP, m.alpha, Kinv, partial, ls);
  GPBO_HIP(ctx, hipGetLastError());
  lml_grad_final_kernel<<<dim3((unsigned)(n_ls + (with_terms ? 1 : 0)), (unsigned)ctx->lanes), dim3(256), 0, ctx->stream>>>(
      partial, ntiles, n_ls, out, ls, out_pitch, m.yn, m.alpha, m.L, m.N, m.NP);
  GPBO_HIP(ctx, hipGetLastError());
  return GPBO_OK;
}

}  // namespace gpbo
