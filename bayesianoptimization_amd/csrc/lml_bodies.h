// Device bodies of the log-marginal-likelihood reductions (gfx950), shared by lml_kernels.hip (one launch per phase) and
// fused_small.hip (a whole evaluation of a small problem in ONE workgroup).  `tid` = index within the 256-thread group working on one
// virtual block; `write` false = go through the motions (same barriers) without storing.  Device code only.
#pragma once

#include "gpbo_internal.h"

namespace gpbo {

// sum over the 256 threads of a group (fixed tree: wave shuffles, then the four waves in order); `sh`: 4 doubles of the group's own;
// two barriers
__device__ __forceinline__ double block_sum_256(double v, double* sh, const int tid) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
  const int wave = tid >> 6, lane = tid & 63;
  __syncthreads();
  if (lane == 0) sh[wave] = v;
  __syncthreads();
  return ((sh[0] + sh[1]) + sh[2]) + sh[3];
}

// out[0] = y . alpha ; out[1] = sum_i log L_ii   (one 256-thread group; four barriers)
__device__ __forceinline__ void lml_terms_body(const double* y, const double* alpha, const double* L, const int64_t N, const int64_t NP,
                                               double* out, double* sh, const int tid, const bool write) {
  double a = 0.0, b = 0.0;
  for (int64_t i = tid; i < N; i += 256) {
    a = fma(y[i], alpha[i], a);
    b += log(L[i * NP + i]);
  }
  const double sa = block_sum_256(a, sh, tid);
  const double sb = block_sum_256(b, sh, tid);
  if (tid == 0 && write) {
    out[0] = sa;
    out[1] = sb;
  }
}

// Lower 64x64 tile (bi, bj <= bi) by a 256-thread group: sum over the tile of (alpha_i alpha_j - Kinv_ij) * dK_ij/dtheta_t.
// `smem`: 2 * DP * 64 + 4 doubles; 1 + 2 * (n_ls == 1 ? 1 : n_ls) barriers.  KT: the tile of K^-1 comes as the 64x64 row-major image
// `ktile` (the fused K^-1 + gradient kernel keeps it in LDS) instead of from Kinv.
template <int KERNEL, bool KT = false>
__device__ __forceinline__ void lml_grad_tile_body(const double* Xs, const int DP, const int n_ls, const int64_t N, const int64_t NP,
                                                   const double* alpha, const double* Kinv, double* partial, const int bi, const int bj,
                                                   double* smem, const int tid, const bool write, const double* ktile = nullptr) {
  double* XiT = smem;             // [DP][64]
  double* XjT = smem + DP * 64;   // [DP][64]
  double* sh = XjT + DP * 64;     // [4]
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
        double kin;
        if constexpr (KT) kin = ktile[(ty * 4 + a) * 64 + tx * 4 + b];
        else kin = Kinv[i * NP + j];
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
    const double tot = block_sum_256(s_iso, sh, tid);
    if (tid == 0 && write) partial[tile] = tot;
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
      const double tot = block_sum_256(s, sh, tid);
      if (tid == 0 && write) partial[tile * n_ls + t] = tot;
    }
  }
}

// out[t] = 0.5 * sum over tiles (fixed order) of partial[tile][t]   (one 256-thread group per t; two barriers)
__device__ __forceinline__ void lml_grad_final_body(const double* partial, const int64_t ntiles, const int n_ls, double* out, const int t,
                                                    double* sh, const int tid, const bool write) {
  double s = 0.0;
  for (int64_t k = tid; k < ntiles; k += 256) s += partial[k * n_ls + t];
  const double tot = block_sum_256(s, sh, tid);
  if (tid == 0 && write) out[t] = 0.5 * tot;
}

}  // namespace gpbo
