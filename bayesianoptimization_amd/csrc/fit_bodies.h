// Device bodies of the fit-side kernels (gfx950), shared by fit_kernels.hip (one launch per phase) and fused_small.hip (the whole
// fit of a small problem in ONE workgroup).  Each body is the arithmetic of the kernel of the same name, with the thread / block
// indices as parameters: `tid` = index within the 256-thread group that works on one virtual block, the block ids spelled out.
// Device code only.  What each replaces in the reference: see fit_kernels.hip.
#pragma once

#include "gpbo_internal.h"

namespace gpbo {

// X / length_scale into a zero-padded [n_pad][DP] image (true division, as numpy does): element idx
__device__ __forceinline__ void prescale_elem(const double* X, const int64_t n, const int d, const int DP, const double* ls, double* out,
                                              const int64_t idx) {
  const int64_t row = idx / DP;
  const int t = (int)(idx - row * DP);
  double v = 0.0;
  if (row < n && t < d) v = X[row * d + t] / ls[t];
  out[idx] = v;
}

// mu = y_std * (k* . alpha) + y_mean ; sd = sqrt(max(1 - |W k*|^2, 0) * y_std^2) for one candidate (sklearn _gpr.py:444-447, 474-494):
// posterior_finalize_kernel's element, also the epilogue of a fused posterior launch that owns all rows of its candidates.
__device__ __forceinline__ void posterior_finalize_elem(const double ss, const double mun, const double y_mean, const double y_std,
                                                        double* mu, double* sd, int* negvar) {
  double var = 1.0 - ss;
  if (var < 0.0) {                   // _gpr.py:479-485 (NaN stays NaN, as in numpy); the host warns as sklearn does
    *negvar = 1;
    var = 0.0;
  }
  var = var * (y_std * y_std);
  *sd = sqrt(var);
  *mu = y_std * mun + y_mean;
}

// Kernel value from a squared scaled distance: ONE arithmetic for both sides of the GP (gpbo_kernel_value in
// gpbo_internal.h: v_rsq-seeded sqrt, K^2 * (1/3)) — the fit-side K and the posterior-side k* agree bit for bit
// for equal distances, and both stay within ~1 ulp of sklearn's expression (kernels.py:1722-1724, 1559-1560).
template <int KERNEL>
__device__ __forceinline__ double kernel_value(double d2) {
  return gpbo_kernel_value<KERNEL>(d2);
}

// One lower 64x64 tile (bi, bj <= bi) of K by a 256-thread group: 4x4 outputs per thread, the two point tiles staged k-major in
// `smem` (2 * DP * 64 doubles).  One barrier.
template <int KERNEL>
__device__ __forceinline__ void kmat_tile_body(const double* Xs, const int DP, const int64_t N, const int64_t NP, const double noise,
                                               double* K, const int bi, const int bj, double* smem, const int tid) {
  double* XiT = smem;            // [DP][64]
  double* XjT = smem + DP * 64;  // [DP][64]
  // (consecutive threads = consecutive points of one dimension: the dimension-major LDS image is written 512 contiguous
  // bytes per wave.  Until round 4 consecutive threads walked the dimensions of one point — LDS addresses 512 B apart, a
  // DP-way bank conflict on every staging store: SQ_LDS_BANK_CONFLICT 5.1e6 cycles per launch at N = 4096,
  // profiles/r04_pmc_kmat.txt)
  for (int e = tid; e < 64 * DP; e += 256) {
    const int t = e >> 6, r = e & 63;
    XiT[e] = Xs[((int64_t)bi * 64 + r) * DP + t];
    XjT[e] = Xs[((int64_t)bj * 64 + r) * DP + t];
  }
  __syncthreads();
  const int ty = tid >> 4, tx = tid & 15;
  double acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b2 = 0; b2 < 4; ++b2) acc[a][b2] = 0.0;
  for (int t = 0; t < DP; ++t) {
    double xi[4], xj[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) xi[a] = XiT[t * 64 + ty * 4 + a];
#pragma unroll
    for (int b2 = 0; b2 < 4; ++b2) xj[b2] = XjT[t * 64 + tx * 4 + b2];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b2 = 0; b2 < 4; ++b2) {
        double df = xi[a] - xj[b2];
        acc[a][b2] = fma(df, df, acc[a][b2]);
      }
  }
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const int64_t i = (int64_t)bi * 64 + ty * 4 + a;
    double out[4];
#pragma unroll
    for (int b2 = 0; b2 < 4; ++b2) {
      const int64_t j = (int64_t)bj * 64 + tx * 4 + b2;
      double v;
      if (i >= N || j >= N) v = (i == j) ? 1.0 : 0.0;       // identity padding
      else if (i == j) v = 1.0 + noise;                       // unit diagonal (+ alpha, _gpr.py:347)
      else v = kernel_value<KERNEL>(acc[a][b2]);
      out[b2] = v;
    }
    double2* dst = reinterpret_cast<double2*>(K + i * NP + (int64_t)bj * 64 + tx * 4);
    dst[0] = make_double2(out[0], out[1]);
    dst[1] = make_double2(out[2], out[3]);
  }
}

// linear lower-triangle index b = bi (bi + 1) / 2 + bj  ->  (bi, bj)
__device__ __forceinline__ void lower_tile_of(const int b, int& bi, int& bj) {
  bi = (int)((sqrt(8.0 * (double)b + 1.0) - 1.0) * 0.5);
  while ((bi + 1) * (bi + 2) / 2 <= b) ++bi;
  while (bi * (bi + 1) / 2 > b) --bi;
  bj = b - bi * (bi + 1) / 2;
}

// W := blockdiag(dinv): diagonal block kb by a 256-thread group.  `zero_right`: W has NOT been zero-filled (an LML evaluation,
// trtri in gpbo_api.hip) — the 64x64 tile right of an even diagonal block is cleared here, so that a reader that cuts its k-range
// at 128-row granularity (gemm128_f64_kernel over a triangular operand) still finds zeros above the diagonal of its 128-blocks.
__device__ __forceinline__ void fill_w_diag_body(const double* dinv, double* W, const int64_t NP, const int kb, const int tid,
                                                 const bool zero_right = false) {
  const double* D = dinv + (int64_t)kb * 4096;
  const bool right = zero_right && !(kb & 1) && ((int64_t)kb + 2) * 64 <= NP;
  for (int e = tid; e < 4096; e += 256) {
    int r = e >> 6, c = e & 63;
    W[((int64_t)kb * 64 + r) * NP + (int64_t)kb * 64 + c] = D[e];
    if (right) W[((int64_t)kb * 64 + r) * NP + (int64_t)(kb + 1) * 64 + c] = 0.0;
  }
}

// t[i] = (W y)[i]: one wave per row, fixed shuffle tree
__device__ __forceinline__ void trmv_lower_row(const double* W, const double* y, double* t, const int64_t NP, const int64_t i,
                                               const int lane) {
  const double* row = W + i * NP;
  double s = 0.0;
  for (int64_t j = lane; j <= i; j += 64) s = fma(row[j], y[j], s);
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off);
  if (lane == 0) t[i] = s;
}

// alpha = W^T t in two deterministic passes: (column block of 64) x (row split) partial sums, then a
// fixed-order reduction over the row splits.
constexpr int TRMV_SPLITS = 16;

// partial sums of column block bx, row split by; `red`: 4 * 64 doubles of the 256-thread group's own; one barrier
__device__ __forceinline__ void trmv_lower_t_body(const double* W, const double* t, double* partial, const int64_t NP, const int bx,
                                                  const int by, double* red, const int tid) {
  const int ig = tid >> 6, jl = tid & 63;
  const int64_t j0 = (int64_t)bx * 64;
  const int64_t rows = NP - j0;                                   // rows j0 .. NP-1 hold non-zeros
  const int64_t chunk = (rows + TRMV_SPLITS - 1) / TRMV_SPLITS;
  const int64_t r0 = j0 + (int64_t)by * chunk;
  const int64_t r1 = min(NP, r0 + chunk);
  double s = 0.0;
  for (int64_t i = r0 + ig; i < r1; i += 4) s = fma(W[i * NP + j0 + jl], t[i], s);
  red[ig * 64 + jl] = s;
  __syncthreads();
  if (ig == 0) partial[(int64_t)by * NP + j0 + jl] = ((red[jl] + red[64 + jl]) + red[128 + jl]) + red[192 + jl];
}

__device__ __forceinline__ void trmv_reduce_elem(const double* partial, double* alpha, const int64_t NP, const int64_t j) {
  double s = 0.0;
#pragma unroll
  for (int r = 0; r < TRMV_SPLITS; ++r) s += partial[(int64_t)r * NP + j];
  alpha[j] = s;
}

// Pack W into the order the posterior kernel's waves consume it: for row slab s (32 rows), k-pair p
// (8 columns) and 16-row tile t, 64 lanes x 2 doubles contiguous (1 KiB): lane l, element e holds
// W[32 s + 16 t + (l & 15)][8 p + 4 e + (l >> 4)] — the A fragment of v_mfma_f64_16x16x4_f64 for
// k-steps 2p and 2p+1.  Entries outside the N x N lower triangle are zero.
__device__ __forceinline__ void pack_w_elem(const double* W, double* Wp, const int64_t N, const int64_t NP, const int64_t idx) {
  const int e = (int)(idx & 1);
  const int lane = (int)((idx >> 1) & 63);
  const int t = (int)((idx >> 7) & 1);
  const int64_t sp = idx >> 8;
  const int64_t pairs = NP / 8;
  const int64_t s = sp / pairs, p = sp - s * pairs;
  const int64_t row = 32 * s + 16 * t + (lane & 15);
  const int64_t colx = 8 * p + 4 * e + (lane >> 4);
  double v = 0.0;
  if (row < N && colx < N && colx <= row) v = W[row * NP + colx];
  Wp[idx] = v;
}

}  // namespace gpbo
