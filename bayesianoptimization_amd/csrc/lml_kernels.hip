// Log-marginal-likelihood value and gradient pieces on the device.
//
// Replaces, per L-BFGS-B evaluation of GaussianProcessRegressor.log_marginal_likelihood(theta,
// eval_gradient=True) (sklearn/gaussian_process/_gpr.py:575-652):
//   -0.5 y^T alpha - sum(log diag L) - N/2 log(2 pi)                       _gpr.py:601-607
//   0.5 * einsum("ijl,jik->kl", alpha alpha^T - K^-1, dK/dtheta)           _gpr.py:619-645
// with dK/dlog(l) of Matern(nu=2.5) = 5/3 * D * (sqrt(5 D.sum) + 1) * exp(-sqrt(5 D.sum)) and of RBF = D * K
// (kernels.py:1764-1766, 1567-1582; D = squared scaled coordinate differences, summed for an isotropic
// length scale).  K^-1 = W^T W comes from the MFMA GEMM; this file holds the reductions.
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

// out[t] = 0.5 * sum over tiles (fixed order) of partial[tile][t]
__global__ __launch_bounds__(256) void lml_grad_final_kernel(const double* __restrict__ partial, int64_t ntiles,
                                                             int n_ls, double* __restrict__ out, int64_t lane_stride,
                                                             int64_t out_pitch) {
  __shared__ double sh[4];
  const int64_t lo = (int64_t)blockIdx.y * lane_stride;
  lml_grad_final_body(partial + lo, ntiles, n_ls, out + (int64_t)blockIdx.y * out_pitch, (int)blockIdx.x, sh, (int)threadIdx.x, true);
}

int launch_lml_terms(gpbo_ctx* ctx, Model& m, double* out2, int64_t out_pitch) {
  lml_terms_kernel<<<dim3((unsigned)ctx->lanes), dim3(256), 0, ctx->stream>>>(m.yn, m.alpha, m.L, m.N, m.NP, out2,
                                                                                ctx->lane_stride, out_pitch);
  GPBO_HIP(ctx, hipGetLastError());
  return GPBO_OK;
}

// Kinv must hold K^-1 (lower 64x64 tiles incl. full diagonal tiles); partial needs ntiles*n_ls doubles.
int launch_lml_grad(gpbo_ctx* ctx, Model& m, int n_ls, const double* Kinv, double* partial, double* grad, int64_t out_pitch) {
  const unsigned nb = (unsigned)(m.NP / 64);
  const int64_t ntiles = (int64_t)nb * (nb + 1) / 2;
  const size_t lds = (size_t)(2 * m.DP * 64 + 8) * sizeof(double);
  dim3 grid(nb, nb, (unsigned)ctx->lanes);
  const int64_t ls = ctx->lane_stride;
  if (m.kernel == GPBO_KERNEL_MATERN25)
    lml_grad_kernel<GPBO_KERNEL_MATERN25><<<grid, dim3(256), lds, ctx->stream>>>(m.Xs, m.DP, n_ls, m.N, m.NP, m.alpha, Kinv, partial, ls);
  else
    lml_grad_kernel<GPBO_KERNEL_RBF><<<grid, dim3(256), lds, ctx->stream>>>(m.Xs, m.DP, n_ls, m.N, m.NP, m.alpha, Kinv, partial, ls);
  GPBO_HIP(ctx, hipGetLastError());
  lml_grad_final_kernel<<<dim3((unsigned)n_ls, (unsigned)ctx->lanes), dim3(256), 0, ctx->stream>>>(partial, ntiles, n_ls, grad, ls, out_pitch);
  GPBO_HIP(ctx, hipGetLastError());
  return GPBO_OK;
}

}  // namespace gpbo
