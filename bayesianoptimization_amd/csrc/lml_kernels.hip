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

namespace gpbo {

__device__ __forceinline__ double block_sum_256(double v, double* sh) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();
  if (lane == 0) sh[wave] = v;
  __syncthreads();
  return ((sh[0] + sh[1]) + sh[2]) + sh[3];
}

// out[0] = y . alpha ; out[1] = sum_i log L_ii
__global__ __launch_bounds__(256) void lml_terms_kernel(const double* __restrict__ y, const double* __restrict__ alpha,
                                                        const double* __restrict__ L, int64_t N, int64_t NP,
                                                        double* __restrict__ out, int64_t lane_stride) {
  __shared__ double sh[4];
  y += (int64_t)blockIdx.x * lane_stride;
  alpha += (int64_t)blockIdx.x * lane_stride;
  L += (int64_t)blockIdx.x * lane_stride;
  out += (int64_t)blockIdx.x * lane_stride;
  double a = 0.0, b = 0.0;
  for (int64_t i = threadIdx.x; i < N; i += 256) {
    a = fma(y[i], alpha[i], a);
    b += log(L[i * NP + i]);
  }
  const double sa = block_sum_256(a, sh);
  const double sb = block_sum_256(b, sh);
  if (threadIdx.x == 0) {
    out[0] = sa;
    out[1] = sb;
  }
}

// One workgroup per lower 64x64 tile: sum over the tile of (alpha_i alpha_j - Kinv_ij) * dK_ij/dtheta_t.
template <int KERNEL>
__global__ __launch_bounds__(256) void lml_grad_kernel(const double* __restrict__ Xs, int DP, int n_ls, int64_t N,
                                                       int64_t NP, const double* __restrict__ alpha,
                                                       const double* __restrict__ Kinv, double* __restrict__ partial,
                                                       int64_t lane_stride) {
  const int bj = blockIdx.x, bi = blockIdx.y;
  if (bj > bi) return;
  Xs += (int64_t)blockIdx.z * lane_stride;
  alpha += (int64_t)blockIdx.z * lane_stride;
  Kinv += (int64_t)blockIdx.z * lane_stride;
  partial += (int64_t)blockIdx.z * lane_stride;
  extern __shared__ __attribute__((aligned(16))) double lg_smem[];
  double* XiT = lg_smem;             // [DP][64]
  double* XjT = lg_smem + DP * 64;   // [DP][64]
  double* sh = XjT + DP * 64;        // [4]
  const int tid = threadIdx.x;
  for (int e = tid; e < 64 * DP; e += 256) {
    int r = e / DP, t = e - r * DP;
    XiT[t * 64 + r] = Xs[((int64_t)bi * 64 + r) * DP + t];
    XjT[t * 64 + r] = Xs[((int64_t)bj * 64 + r) * DP + t];
  }
  __syncthreads();
  const int ty = tid >> 4, tx = tid & 15;
  double d2[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) d2[a][b] = 0.0;
  for (int t = 0; t < DP; ++t) {
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const double df = XiT[t * 64 + ty * 4 + a] - XjT[t * 64 + tx * 4 + b];
        d2[a][b] = fma(df, df, d2[a][b]);
      }
  }
  const double wgt = (bi == bj) ? 1.0 : 2.0;   // off-diagonal tiles stand for their mirror image too
  double coef[4][4];
  double s_iso = 0.0;
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const int64_t i = (int64_t)bi * 64 + ty * 4 + a;
    const double ai = (i < N) ? alpha[i] : 0.0;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int64_t j = (int64_t)bj * 64 + tx * 4 + b;
      double c = 0.0;
      if (i < N && j < N && i != j) {
        const double aj = alpha[j];
        const double kin = Kinv[i * NP + j];
        double g;
        if (KERNEL == GPBO_KERNEL_MATERN25) {
          const double tmp = sqrt(5.0 * d2[a][b]);
          g = 5.0 / 3.0 * (tmp + 1.0) * gpbo_exp_nonpos(-tmp);
        } else {
          g = gpbo_exp_nonpos(-0.5 * d2[a][b]);
        }
        c = wgt * (ai * aj - kin) * g;
      }
      coef[a][b] = c;
      s_iso = fma(c, d2[a][b], s_iso);
    }
  }
  const int64_t tile = (int64_t)bi * (bi + 1) / 2 + bj;
  if (n_ls == 1) {
    const double tot = block_sum_256(s_iso, sh);
    if (tid == 0) partial[tile] = tot;
  } else {
    for (int t = 0; t < n_ls; ++t) {
      double s = 0.0;
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const double df = XiT[t * 64 + ty * 4 + a] - XjT[t * 64 + tx * 4 + b];
          s = fma(coef[a][b], df * df, s);
        }
      const double tot = block_sum_256(s, sh);
      if (tid == 0) partial[tile * n_ls + t] = tot;
    }
  }
}

// out[t] = 0.5 * sum over tiles (fixed order) of partial[tile][t]
__global__ __launch_bounds__(256) void lml_grad_final_kernel(const double* __restrict__ partial, int64_t ntiles,
                                                             int n_ls, double* __restrict__ out, int64_t lane_stride) {
  __shared__ double sh[4];
  const int t = blockIdx.x;
  partial += (int64_t)blockIdx.y * lane_stride;
  out += (int64_t)blockIdx.y * lane_stride;
  double s = 0.0;
  for (int64_t k = threadIdx.x; k < ntiles; k += 256) s += partial[k * n_ls + t];
  const double tot = block_sum_256(s, sh);
  if (threadIdx.x == 0) out[t] = 0.5 * tot;
}

int launch_lml_terms(gpbo_ctx* ctx, Model& m, double* out2_dev) {
  lml_terms_kernel<<<dim3((unsigned)ctx->lanes), dim3(256), 0, ctx->stream>>>(m.yn, m.alpha, m.L, m.N, m.NP, out2_dev,
                                                                                ctx->lane_stride);
  GPBO_HIP(ctx, hipGetLastError());
  return GPBO_OK;
}

// Kinv must hold K^-1 (lower 64x64 tiles incl. full diagonal tiles); partial needs ntiles*n_ls doubles.
int launch_lml_grad(gpbo_ctx* ctx, Model& m, int n_ls, const double* Kinv, double* partial, double* grad_dev) {
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
  lml_grad_final_kernel<<<dim3((unsigned)n_ls, (unsigned)ctx->lanes), dim3(256), 0, ctx->stream>>>(partial, ntiles, n_ls, grad_dev, ls);
  GPBO_HIP(ctx, hipGetLastError());
  return GPBO_OK;
}

}  // namespace gpbo
