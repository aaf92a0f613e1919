// Small-batch posterior (M <= a few hundred candidates, see small_batch_limit): the latency path behind HipGPR.predict when the reference's
// "smart" stage (bayes_opt/acquisition.py:322-420: L-BFGS-B with finite differences) asks for one point — or,
// with the batched finite-difference gradient of fused_acquisition.py, d + 1 points — at a time.  Same arithmetic as posterior_kernel_v2 (sklearn _gpr.py:443-494), organised as a memory-bound
// batched GEMV over the row-major W = L^-1 (read once, ~N^2/2 * 8 B) instead of an MFMA GEMM:
//   kstar_small_kernel : k*[c][k] for all train points (N x M values)
//   gemv_small_kernel  : v[c][i] = sum_k W[i][k] k*[c][k], one wave per 4 rows, fixed shuffle tree
//   finalize_small     : mu = y_std * (k* . alpha) + y_mean ; sd = sqrt(max(1 - sum_i v^2, 0)) * y_std
#include <algorithm>
#include <cstdlib>

#include "gpbo_internal.h"

namespace gpbo {

constexpr int SMALL_MAX = 1024;   // scratch bound: n_seeds * (d + 1) finite-difference points of a lockstep round

// Largest batch the GEMV path takes before the MFMA path is the faster one.  The GEMV path re-reads W once per
// pass of 16 candidates; the MFMA path's time is flat in M until its (row chunk x 64-candidate tile) grid fills
// the chip, and is set by the longest row chunk on one CU.  Measured crossovers on MI355X
// (profiles/r01_small_batch_latency.json, predict() latency incl. ~95 us of host/PCIe overhead):
//   N = 512: ~60   N = 1024: ~550   N = 2048: ~300   N = 4096: ~130   N = 8192: ~70
// (N <= 512 runs the single fused MFMA kernel, which is why its crossover is low.)  GPBO_SMALL_MAX overrides
// the rule (A/B runs, and tests that pin one path).
int small_batch_limit(int64_t NP) {
  if (const char* e = dbg_env("GPBO_SMALL_MAX")) {
    const long v = atol(e);
    return (int)(v < 0 ? 0 : (v > SMALL_MAX ? SMALL_MAX : v));
  }
  if (NP <= 512) return 48;
  if (NP <= 1024) return 512;
  if (NP <= 2048) return 256;
  if (NP <= 4096) return 128;
  return 72;
}

template <int KERNEL>
__global__ __launch_bounds__(256) void kstar_small_kernel(const double* __restrict__ Xs, const double* __restrict__ Xcs,
                                                          int DP, int64_t NP, int M, double* __restrict__ ks) {
  const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (k >= NP) return;
  const double* xr = Xs + k * DP;
  const int c_end = min(M, ((int)blockIdx.y + 1) * 16);
  for (int c = (int)blockIdx.y * 16; c < c_end; ++c) {
    const double* xc = Xcs + (int64_t)c * DP;
    double d2 = 0.0;
    for (int t = 0; t < DP; ++t) {
      const double df = xc[t] - xr[t];
      d2 = fma(df, df, d2);
    }
    ks[(int64_t)c * NP + k] = gpbo_kernel_value<KERNEL>(d2);
  }
}

// vsq[c][i] = (sum_k W[i][k] ks[c][k])^2 for i < N (rows >= N contribute 0); one wave per R rows, MS candidates.
template <int MS, int R>
__global__ __launch_bounds__(256) void gemv_small_kernel(const double* __restrict__ W, const double* __restrict__ ks,
                                                         int64_t N, int64_t NP, double* __restrict__ vsq) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t i0 = ((int64_t)blockIdx.x * 4 + wave) * R;
  if (i0 >= NP) return;
  ks += (int64_t)blockIdx.y * MS * NP;    // blockIdx.y = pass of MS candidates
  vsq += (int64_t)blockIdx.y * MS * NP;
  double acc[R][MS];
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int c = 0; c < MS; ++c) acc[r][c] = 0.0;
  const int64_t kmax = min(NP, i0 + R);   // W is lower triangular with an explicit zero upper part
  for (int64_t k = lane; k < kmax; k += 64) {
    double w[R];
#pragma unroll
    for (int r = 0; r < R; ++r) w[r] = W[(i0 + r) * NP + k];
#pragma unroll
    for (int c = 0; c < MS; ++c) {
      const double kv = ks[(int64_t)c * NP + k];
#pragma unroll
      for (int r = 0; r < R; ++r) acc[r][c] = fma(w[r], kv, acc[r][c]);
    }
  }
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int c = 0; c < MS; ++c) {
      double v = acc[r][c];
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
      if (lane == 0) vsq[(int64_t)c * NP + i0 + r] = (i0 + r < N) ? v * v : 0.0;
    }
}

template <int MS, int R>
static void launch_gemv(gpbo_ctx* ctx, Model& m, const double* ks, double* vsq, int passes = 1) {
  const unsigned gb = (unsigned)((m.NP / R + 3) / 4);
  gemv_small_kernel<MS, R><<<dim3(gb, (unsigned)passes), dim3(256), 0, ctx->stream>>>(m.W, ks, m.N, m.NP, vsq);
}

__global__ __launch_bounds__(256) void finalize_small_kernel(const double* __restrict__ vsq, const double* __restrict__ ks,
                                                             const double* __restrict__ alpha, int64_t NP,
                                                             double y_mean, double y_std, double* __restrict__ mu,
                                                             double* __restrict__ sd, int* __restrict__ negvar) {
  __shared__ double sh[4];
  const int c = blockIdx.x;
  double s = 0.0, m = 0.0;
  for (int64_t i = threadIdx.x; i < NP; i += 256) {
    s += vsq[(int64_t)c * NP + i];
    m = fma(ks[(int64_t)c * NP + i], alpha[i], m);
  }
  double tot[2] = {s, m};
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    double v = tot[q];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    tot[q] = ((sh[0] + sh[1]) + sh[2]) + sh[3];
  }
  if (threadIdx.x == 0) {
    double var = 1.0 - tot[0];
    if (var < 0.0) {
      *negvar = 1;
      var = 0.0;
    }
    var = var * (y_std * y_std);
    sd[c] = sqrt(var);
    mu[c] = y_std * tot[1] + y_mean;
  }
}

// Requires ctx->Xcs to hold the scaled candidates ([M][DP]); uses ctx->part as scratch (2 * M * NP doubles).
int launch_posterior_small(gpbo_ctx* ctx, Model& m, int M, double y_mean, double y_std) {
  if (M < 1 || M > SMALL_MAX) GPBO_FAIL(ctx, GPBO_ERR_INVALID, "posterior_small: M out of range");
  int rc;
  // scratch: ks and vsq, each (SMALL_MAX + 16) rows so that a padded last pass stays inside the allocation
  const int64_t rows = SMALL_MAX + 16;
  if ((rc = ensure(ctx, &ctx->part, &ctx->cap_part, (int64_t)2 * rows * m.NP))) return rc;
  double* ks = ctx->part;
  double* vsq = ctx->part + rows * m.NP;
  const dim3 kgrid((unsigned)((m.NP + 255) / 256), (unsigned)((M + 15) / 16));
  if (m.kernel == GPBO_KERNEL_MATERN25)
    kstar_small_kernel<GPBO_KERNEL_MATERN25><<<kgrid, dim3(256), 0, ctx->stream>>>(m.Xs, ctx->Xcs, m.DP, m.NP, M, ks);
  else
    kstar_small_kernel<GPBO_KERNEL_RBF><<<kgrid, dim3(256), 0, ctx->stream>>>(m.Xs, ctx->Xcs, m.DP, m.NP, M, ks);
  GPBO_HIP(ctx, hipGetLastError());
  // passes of up to 16 candidates (the row order of the dot products does not depend on the pass width, so a
  // candidate's result is bitwise the same whether it is evaluated alone or inside a batch)
  const int full = M / 16;
  if (full > 0) launch_gemv<16, 2>(ctx, m, ks, vsq, full);   // all full passes in one launch (grid.y)
  if (M % 16) {
    const int c0 = full * 16, mc = M - c0;
    const double* ksp = ks + (int64_t)c0 * m.NP;
    double* vp = vsq + (int64_t)c0 * m.NP;
    if (mc == 1) launch_gemv<1, 4>(ctx, m, ksp, vp);
    else if (mc == 2) launch_gemv<2, 4>(ctx, m, ksp, vp);
    else if (mc <= 4) launch_gemv<4, 4>(ctx, m, ksp, vp);
    else if (mc <= 8) launch_gemv<8, 4>(ctx, m, ksp, vp);
    else launch_gemv<16, 2>(ctx, m, ksp, vp);
  }
  GPBO_HIP(ctx, hipGetLastError());
  finalize_small_kernel<<<dim3((unsigned)M), dim3(256), 0, ctx->stream>>>(vsq, ks, m.alpha, m.NP, y_mean, y_std, m.mu, m.sd, ctx->negvar);
  GPBO_HIP(ctx, hipGetLastError());
  return GPBO_OK;
}


// ================================================================================================================
// Posterior WITH its gradient in the inputs: what one L-BFGS-B evaluation of the reference's local
// search (bayes_opt/acquisition.py:365-374, `minimize(acq, x0, method="L-BFGS-B")` with finite differences: d + 1
// predicts) needs, from ONE point instead of d + 1.  With xs = x / l, k_k = k(xs, Xs_k), v = W k, u = W^T v:
//   mu      = y_std * sum_k alpha_k k_k + y_mean          d mu    / d x_t = y_std * sum_k alpha_k dk_k/dxs_t / l_t
//   var_n   = 1 - v.v                                     d var_n / d x_t = -2 * sum_k u_k dk_k/dxs_t / l_t
//   sd      = y_std * sqrt(max(var_n, 0))                 d sd    / d x_t = y_std * (d var_n / d x_t) / (2 sqrt(var_n))
//   dk_k/dxs_t = f_k * (xs_t - Xs_kt),  f_k = -(5/3) (1 + sqrt5 r) exp(-sqrt5 r)  (Matern-2.5, no 1/r singularity)
//                                       f_k = -k_k                                 (RBF)
// (sklearn has no analytic input gradient; the formulas are the derivatives of kernels.py:1722-1724 / 1559-1560 and of
// _gpr.py:443-494.)  Kernels: kstar_grad_small (k and f), gemv_small<STORE_V> (v), gemvt_small + reduce (u = W^T v in
// two deterministic passes), grad_small (the two k-sums per dimension, fixed reduction order), finalize_small.
template <int KERNEL>
__global__ __launch_bounds__(256) void kstar_grad_small_kernel(const double* __restrict__ Xs, const double* __restrict__ Xcs,
                                                               int DP, int64_t NP, int64_t N, int M, double* __restrict__ ks,
                                                               double* __restrict__ fs) {
  const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (k >= NP) return;
  const double* xr = Xs + k * DP;
  const int c_end = min(M, ((int)blockIdx.y + 1) * 16);
  for (int c = (int)blockIdx.y * 16; c < c_end; ++c) {
    const double* xc = Xcs + (int64_t)c * DP;
    double d2 = 0.0;
    for (int t = 0; t < DP; ++t) {
      const double df = xc[t] - xr[t];
      d2 = fma(df, df, d2);
    }
    const double kv = gpbo_kernel_value<KERNEL>(d2);
    double f;
    if (KERNEL == GPBO_KERNEL_MATERN25) {
      const double s = gpbo_sqrt_pos(d2) * 2.23606797749978969641;      // sqrt(5) r
      f = -1.66666666666666666667 * (1.0 + s) * gpbo_exp_nonpos(-s);
    } else {
      f = -kv;
    }
    ks[(int64_t)c * NP + k] = kv;
    fs[(int64_t)c * NP + k] = (k < N) ? f : 0.0;      // padding rows carry no gradient
  }
}

// v[c][i] = sum_k W[i][k] ks[c][k] (rows >= N: 0) — the arithmetic of gemv_small_kernel, storing v instead of v^2
template <int MS, int R>
__global__ __launch_bounds__(256) void gemv_small_v_kernel(const double* __restrict__ W, const double* __restrict__ ks,
                                                           int64_t N, int64_t NP, double* __restrict__ vout) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t i0 = ((int64_t)blockIdx.x * 4 + wave) * R;
  if (i0 >= NP) return;
  ks += (int64_t)blockIdx.y * MS * NP;
  vout += (int64_t)blockIdx.y * MS * NP;
  double acc[R][MS];
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int c = 0; c < MS; ++c) acc[r][c] = 0.0;
  const int64_t kmax = min(NP, i0 + R);
  for (int64_t k = lane; k < kmax; k += 64) {
    double w[R];
#pragma unroll
    for (int r = 0; r < R; ++r) w[r] = W[(i0 + r) * NP + k];
#pragma unroll
    for (int c = 0; c < MS; ++c) {
      const double kv = ks[(int64_t)c * NP + k];
#pragma unroll
      for (int r = 0; r < R; ++r) acc[r][c] = fma(w[r], kv, acc[r][c]);
    }
  }
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int c = 0; c < MS; ++c) {
      double v = acc[r][c];
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
      if (lane == 0) vout[(int64_t)c * NP + i0 + r] = (i0 + r < N) ? v : 0.0;
    }
}

// The same product for long rows (NP >= 2048): the four waves of a workgroup share ONE pair of rows, each taking a quarter
// of the k-range, and their partial sums are added in wave order — a wave of gemv_small_v_kernel walks a 4096-long row
// in 64 dependent turns of ten loads (115 us per pass of 8 candidates at N = 4096, 1.1 TB/s); here in 16.
template <int MS, int R>
__global__ __launch_bounds__(256) void gemv_small_v4_kernel(const double* __restrict__ W, const double* __restrict__ ks,
                                                            int64_t N, int64_t NP, double* __restrict__ vout) {
  __shared__ double red[4][R][MS];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t i0 = (int64_t)blockIdx.x * R;
  if (i0 >= NP) return;
  ks += (int64_t)blockIdx.y * MS * NP;
  vout += (int64_t)blockIdx.y * MS * NP;
  double acc[R][MS];
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int c = 0; c < MS; ++c) acc[r][c] = 0.0;
  const int64_t kmax = min(NP, i0 + R);
  const int64_t quarter = ((kmax + 3) / 4 + 63) / 64 * 64;
  const int64_t kb = (int64_t)wave * quarter, ke = min(kmax, kb + quarter);
#pragma unroll 2
  for (int64_t k = kb + lane; k < ke; k += 64) {
    double w[R];
#pragma unroll
    for (int r = 0; r < R; ++r) w[r] = W[(i0 + r) * NP + k];
#pragma unroll
    for (int c = 0; c < MS; ++c) {
      const double kv = ks[(int64_t)c * NP + k];
#pragma unroll
      for (int r = 0; r < R; ++r) acc[r][c] = fma(w[r], kv, acc[r][c]);
    }
  }
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int c = 0; c < MS; ++c) {
      double v = acc[r][c];
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
      if (lane == 0) red[wave][r][c] = v;
    }
  __syncthreads();
  if (threadIdx.x < R * MS) {
    const int r = threadIdx.x / MS, c = threadIdx.x % MS;
    const double v = ((red[0][r][c] + red[1][r][c]) + red[2][r][c]) + red[3][r][c];
    vout[(int64_t)c * NP + i0 + r] = (i0 + r < N) ? v : 0.0;
  }
}

// partial[split][c][j] = sum over the split's rows i >= j of W[i][j] v[c][i]: 64 columns per workgroup, 4 row lanes,
// GS candidates per pass (blockIdx.z), fixed summation order
constexpr int GRAD_SPLITS = 16;       // row splits for NP < 2048
constexpr int GRAD_SPLITS_BIG = 32;   // ... from NP = 2048 on (with 8 row lanes per workgroup: 16 turns per thread at N = 4096 instead of 64)
constexpr int GRAD_MS = 8;

template <int ROWL>
__global__ __launch_bounds__(64 * ROWL) void gemvt_small_kernel(const double* __restrict__ W, const double* __restrict__ v,
                                                                int64_t NP, int M, int n_splits, double* __restrict__ partial) {
  __shared__ double red[ROWL][GRAD_MS][64];
  const int ig = threadIdx.x >> 6, jl = threadIdx.x & 63;
  const int64_t j0 = (int64_t)blockIdx.x * 64;
  const int c0 = (int)blockIdx.z * GRAD_MS;
  const int64_t rows = NP - j0;
  const int64_t chunk = (rows + n_splits - 1) / n_splits;
  const int64_t r0 = j0 + (int64_t)blockIdx.y * chunk;
  const int64_t r1 = min(NP, r0 + chunk);
  double acc[GRAD_MS];
#pragma unroll
  for (int c = 0; c < GRAD_MS; ++c) acc[c] = 0.0;
  for (int64_t i = r0 + ig; i < r1; i += ROWL) {
    const double w = W[i * NP + j0 + jl];
#pragma unroll
    for (int c = 0; c < GRAD_MS; ++c)
      if (c0 + c < M) acc[c] = fma(w, v[(int64_t)(c0 + c) * NP + i], acc[c]);
  }
#pragma unroll
  for (int c = 0; c < GRAD_MS; ++c) red[ig][c][jl] = acc[c];
  __syncthreads();
  if (ig == 0) {
#pragma unroll
    for (int c = 0; c < GRAD_MS; ++c)
      if (c0 + c < M) {
        double sum = red[0][c][jl];
#pragma unroll
        for (int q = 1; q < ROWL; ++q) sum += red[q][c][jl];
        partial[((int64_t)blockIdx.y * M + c0 + c) * NP + j0 + jl] = sum;
      }
  }
}

// The two k-sums per dimension.  Grid (candidate, k-slice): a slice of NP / GRAD_KSL train points per workgroup, threads =
// DP dimensions x (256 / DP) k-lanes; u_k = sum over the splits (fixed order); per slice the sums are reduced over the
// k-lanes in a fixed order and left in gpart[c][slice][2][DP]; grad_final_kernel adds the slices in order.  (One workgroup
// per candidate until round 3: ten workgroups walking 4096 rows with 19 loads per turn were bound by their ten CUs'
// load issue — 138 us of a 0.4 ms round at N = 4096.)
constexpr int GRAD_KSL = 16;
__global__ __launch_bounds__(256) void grad_small_kernel(const double* __restrict__ Xs, const double* __restrict__ Xcs,
                                                         const double* __restrict__ fs, const double* __restrict__ partial,
                                                         const double* __restrict__ alpha, int DP, int64_t NP, int M,
                                                         int n_splits, double* __restrict__ gpart) {
  __shared__ double gs_smem[2][256];
  const int c = blockIdx.x, sl = blockIdx.y;
  const int t = threadIdx.x % DP, kl = threadIdx.x / DP, nkl = 256 / DP;
  const int64_t per = (NP + GRAD_KSL - 1) / GRAD_KSL;
  const int64_t k0 = (int64_t)sl * per, k1 = min(NP, k0 + per);
  const double xt = Xcs[(int64_t)c * DP + t];
  double gm = 0.0, gv = 0.0;
  for (int64_t k = k0 + kl; k < k1; k += nkl) {
    double u = 0.0;
#pragma unroll 16
    for (int s = 0; s < n_splits; ++s) u += partial[((int64_t)s * M + c) * NP + k];
    const double f = fs[(int64_t)c * NP + k];
    const double df = (xt - Xs[k * DP + t]) * f;
    gm = fma(alpha[k], df, gm);
    gv = fma(u, df, gv);
  }
  gs_smem[0][threadIdx.x] = gm;
  gs_smem[1][threadIdx.x] = gv;
  __syncthreads();
  if ((int)threadIdx.x < DP) {
    double a = 0.0, b = 0.0;
    for (int q = 0; q < nkl; ++q) {
      a += gs_smem[0][q * DP + threadIdx.x];
      b += gs_smem[1][q * DP + threadIdx.x];
    }
    double* o = gpart + (((int64_t)c * GRAD_KSL + sl) * 2) * DP;
    o[threadIdx.x] = a;
    o[DP + threadIdx.x] = b;
  }
}

// One workgroup per candidate: mean and sum of squares (the finalize_small arithmetic), then the slices' sums in order.
__global__ __launch_bounds__(256) void grad_final_kernel(const double* __restrict__ vbuf, const double* __restrict__ ks,
                                                         const double* __restrict__ alpha, const double* __restrict__ ls,
                                                         const double* __restrict__ gpart, int DP, int d, int64_t NP,
                                                         double y_mean, double y_std, double* __restrict__ mu,
                                                         double* __restrict__ sd, double* __restrict__ dmu,
                                                         double* __restrict__ dsd, int* __restrict__ negvar) {
  __shared__ double sh[8];
  const int c = blockIdx.x;
  double s2 = 0.0, mm = 0.0;
  for (int64_t i = threadIdx.x; i < NP; i += 256) {
    const double vi = vbuf[(int64_t)c * NP + i];
    s2 = fma(vi, vi, s2);
    mm = fma(ks[(int64_t)c * NP + i], alpha[i], mm);
  }
  double tot[2] = {s2, mm};
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    double v = tot[q];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    tot[q] = ((sh[0] + sh[1]) + sh[2]) + sh[3];
  }
  double var = 1.0 - tot[0];
  if (var < 0.0) {
    if (threadIdx.x == 0) *negvar = 1;
    var = 0.0;
  }
  const double sdn = sqrt(var);
  if (threadIdx.x == 0) {
    sd[c] = sqrt(var * (y_std * y_std));
    mu[c] = y_std * tot[1] + y_mean;
  }
  if ((int)threadIdx.x < d) {
    double a = 0.0, b = 0.0;
    for (int sl = 0; sl < GRAD_KSL; ++sl) {
      const double* o = gpart + (((int64_t)c * GRAD_KSL + sl) * 2) * DP;
      a += o[threadIdx.x];
      b += o[DP + threadIdx.x];
    }
    const double inv_l = 1.0 / ls[threadIdx.x];
    dmu[(int64_t)c * d + threadIdx.x] = y_std * a * inv_l;
    // d sd / d x = y_std * (-2 b / l) / (2 sqrt(var_n)); a clipped (zero) variance has no slope
    dsd[(int64_t)c * d + threadIdx.x] = (sdn > 0.0) ? -(y_std * b * inv_l) / sdn : 0.0;
  }
}

// Requires ctx->Xcs to hold the scaled candidates; scratch in ctx->part.  Outputs on the device: m.mu, m.sd (M) and
// dmu_dev, dsd_dev (M x d).
int launch_posterior_grad_small(gpbo_ctx* ctx, Model& m, int M, double y_mean, double y_std, double* dmu_dev,
                                double* dsd_dev, double* mu_out, double* sd_out) {
  if (M < 1 || M > GPBO_MAX_SEEDS * 4) GPBO_FAIL(ctx, GPBO_ERR_INVALID, "posterior_grad: M out of range [1, 256]");
  int rc;
  const int64_t rows = (int64_t)M + 16;
  // ks | fs | v | partial[GRAD_SPLITS][M] | gpart[M][GRAD_KSL][2][DP]
  const int64_t gpart_doubles = (int64_t)M * GRAD_KSL * 2 * m.DP;
  const int n_splits = m.NP >= 2048 ? GRAD_SPLITS_BIG : GRAD_SPLITS;
  if ((rc = ensure(ctx, &ctx->part, &ctx->cap_part, std::max<int64_t>((int64_t)2 * (SMALL_MAX + 16) * m.NP,
                                                                       (3 * rows + (int64_t)n_splits * M) * m.NP + gpart_doubles))))
    return rc;
  double* ks = ctx->part;
  double* fs = ks + rows * m.NP;
  double* vb = fs + rows * m.NP;
  double* partial = vb + rows * m.NP;
  double* gpart = partial + (int64_t)n_splits * M * m.NP;
  const dim3 kgrid((unsigned)((m.NP + 255) / 256), (unsigned)((M + 15) / 16));
  if (m.kernel == GPBO_KERNEL_MATERN25)
    kstar_grad_small_kernel<GPBO_KERNEL_MATERN25><<<kgrid, dim3(256), 0, ctx->stream>>>(m.Xs, ctx->Xcs, m.DP, m.NP, m.N, M, ks, fs);
  else
    kstar_grad_small_kernel<GPBO_KERNEL_RBF><<<kgrid, dim3(256), 0, ctx->stream>>>(m.Xs, ctx->Xcs, m.DP, m.NP, m.N, M, ks, fs);
  GPBO_HIP(ctx, hipGetLastError());
  // v = W k*: passes of 8 candidates (a padded last pass reads/writes scratch rows that exist: rows = M + 16)
  {
    const unsigned gb = (unsigned)((m.NP / 2 + 3) / 4);
    if (m.NP >= 2048)
      gemv_small_v4_kernel<8, 2><<<dim3((unsigned)(m.NP / 2), (unsigned)((M + 7) / 8)), dim3(256), 0, ctx->stream>>>(m.W, ks, m.N, m.NP, vb);
    else
      gemv_small_v_kernel<8, 2><<<dim3(gb, (unsigned)((M + 7) / 8)), dim3(256), 0, ctx->stream>>>(m.W, ks, m.N, m.NP, vb);
    GPBO_HIP(ctx, hipGetLastError());
  }
  const dim3 tgrid((unsigned)(m.NP / 64), (unsigned)n_splits, (unsigned)((M + GRAD_MS - 1) / GRAD_MS));
  if (m.NP >= 2048) gemvt_small_kernel<8><<<tgrid, dim3(512), 0, ctx->stream>>>(m.W, vb, m.NP, M, n_splits, partial);
  else gemvt_small_kernel<4><<<tgrid, dim3(256), 0, ctx->stream>>>(m.W, vb, m.NP, M, n_splits, partial);
  GPBO_HIP(ctx, hipGetLastError());
  grad_small_kernel<<<dim3((unsigned)M, GRAD_KSL), dim3(256), 0, ctx->stream>>>(m.Xs, ctx->Xcs, fs, partial, m.alpha, m.DP, m.NP, M, n_splits,
                                                                                gpart);
  GPBO_HIP(ctx, hipGetLastError());
  grad_final_kernel<<<dim3((unsigned)M), dim3(256), 0, ctx->stream>>>(vb, ks, m.alpha, m.ls, gpart, m.DP, m.d, m.NP, y_mean, y_std,
                                                                      mu_out, sd_out, dmu_dev, dsd_dev, ctx->negvar);
  GPBO_HIP(ctx, hipGetLastError());
  return GPBO_OK;
}

}  // namespace gpbo
