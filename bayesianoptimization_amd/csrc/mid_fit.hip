// The fit / log-marginal-likelihood evaluation of a MID-SIZE problem (fused_max_np() < NP <= mid_max_np(), i.e. 65 ... 512
// observations in the product) in ~15 launches instead of ~45 (gfx950).
//
// What it replaces: GaussianProcessRegressor.fit at fixed theta and log_marginal_likelihood(theta, eval_gradient) for the sizes
// a maximize() loop reaches after its first hundred steps and BASELINE config 2 sits at (sklearn _gpr.py:296-364, 575-652).
//
// Where the time of such a fit went (profiles/r04_trace_C2_one_step_kernel_stats.csv, N = 512: 0.27 ms): not into arithmetic — the
// factorisation chain is 4 x 21 us — but into ~45 stream nodes of 4-12 us each: seven copy / fill nodes in front of the first
// kernel, six GEMM launches for W = L^-1 by recursive doubling (12 us each: one 64x64x512 tile is 13.7 us of fp64 MFMA on its CU
// whatever the grid looks like), three launches for alpha, two copy nodes behind.  In-launch grid barriers are no way out: an
// agent-scope release + acquire costs what a kernel boundary costs (MI355X_MICROARCH.md, barrier-counter / barrier-xcd rows).
// So this path changes the ALGORITHMS to ones with fewer dependent phases, one launch per phase:
//   mid_inputs_kernel   length scales, X / length_scale, padded targets and the pivot word straight from pinned host memory
//                       (or the resident device copies of a theta search): no copy or fill nodes
//   kmat_q_kernel       K, a 64x16 quarter tile per workgroup: the arithmetic of kmat_kernel element for element (same bits), a
//                       quarter of its latency (the tile is fp64-VALU-bound: ~155 instructions per element)
//   launch_cholesky128  unchanged (chol_kernels.hip): L and the inverted 64x64 diagonal blocks are bitwise the large path's
//   w_strip_kernel      W = L^-1 by COLUMN STRIPS: a workgroup owns 16 columns and runs the blocked forward substitution
//                       X_c = D_c,  X_r = -D_r sum_{t<r} L_rt X_t  (D = the inverted diagonal blocks) for them from top to bottom
//                       with the strip resident in LDS — no other strip is ever needed, so ONE launch replaces the memset, the
//                       diagonal fill and the 2 log2(NP/64) GEMM launches of the recursive inverse.  The same launch zero-fills
//                       the strip above the diagonal, packs the strip for the posterior kernels (fit) and leaves its
//                       contribution to t = W y.
//   alpha_strip_kernel  t = sum of the strips' contributions (fixed order), alpha = W^T t for the strip's 16 columns; the pivot
//                       word goes to its pinned host word from here
// and the LML tail (lml_kernels.hip) writes its scalars into pinned host memory itself.
//
// Numerics: K and L are the bits of the large path.  W differs from the recursive inverse in rounding only (both are backward
// stable products of the same 64x64 inverses; the forward substitution is the formulation of sklearn's solve_triangular,
// _gpr.py:454-456) and every sum has a fixed order: results are deterministic and identical between gpbo_fit, gpbo_fit_begin,
// gpbo_lml and the lanes of gpbo_lml_batch.  tests/test_gpu_mid_fit.py holds the parity against the oracle and against the
// large path (debug build: GPBO_MID_MAX_NP=0).
#include "fit_bodies.h"
#include "gemm_tile.h"

namespace gpbo {

// ---- inputs ---------------------------------------------------------------------------------------------------------------
// X (N, d) and y (N) are device-visible (pinned host staging or device memory, shared by all lanes); ls_in = [lanes][64] length
// scales in the pinned window.  Element idx of the zero-padded [NP][DP] image per thread: prescale_elem's arithmetic.
__global__ __launch_bounds__(256) void mid_inputs_kernel(const double* __restrict__ X, const double* __restrict__ y,
                                                         const double* __restrict__ ls_in, int64_t N, int64_t NP, int d, int DP,
                                                         double* __restrict__ ls, double* __restrict__ Xs, double* __restrict__ yn,
                                                         int* __restrict__ info, int64_t lane_stride) {
  const int64_t lo = (int64_t)blockIdx.y * lane_stride;
  const double* lsi = ls_in + (int64_t)blockIdx.y * GPBO_MAX_DIM;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx < NP * DP) prescale_elem(X, N, d, DP, lsi, Xs + lo, idx);
  if (idx < NP) yn[lo + idx] = (idx < N) ? y[idx] : 0.0;
  if (idx < GPBO_MAX_DIM) ls[lo + idx] = lsi[idx];
  if (idx == 0) info[lo * 2] = 0;
}

int launch_mid_inputs(gpbo_ctx* ctx, Model& m, const double* X, const double* y, const double* ls_in) {
  const int64_t total = std::max<int64_t>(m.NP * m.DP, GPBO_MAX_DIM);
  mid_inputs_kernel<<<dim3((unsigned)((total + 255) / 256), (unsigned)ctx->lanes), dim3(256), 0, ctx->stream>>>(
      X, y, ls_in, m.N, m.NP, m.d, m.DP, m.ls, m.Xs, m.yn, ctx->info_dev, ctx->lane_stride);
  GPBO_HIP(ctx, hipGetLastError());
  return GPBO_OK;
}

// ---- K: quarter tiles -------------------------------------------------------------------------------------------------------
// blockIdx.x = 4 * (lower tile index) + quarter: rows of tile row bi, columns 16 q ... 16 q + 15 of tile column bj; thread = one
// row x four columns.  Element for element the arithmetic of kmat_tile_body (fit_bodies.h).
template <int KERNEL>
__global__ __launch_bounds__(256) void kmat_q_kernel(const double* __restrict__ Xs, int DP, int64_t N, int64_t NP, double noise,
                                                     double* __restrict__ K, int64_t lane_stride) {
  extern __shared__ __attribute__((aligned(16))) double kq_smem[];
  int bi, bj;
  lower_tile_of((int)(blockIdx.x >> 2), bi, bj);
  const int q = (int)(blockIdx.x & 3);
  Xs += (int64_t)blockIdx.z * lane_stride;
  K += (int64_t)blockIdx.z * lane_stride;
  const int tid = (int)threadIdx.x;
  double* XiT = kq_smem;             // [DP][64]
  double* XjT = kq_smem + DP * 64;   // [DP][16]
  for (int e = tid; e < 64 * DP; e += 256) {
    const int t = e >> 6, r = e & 63;
    XiT[e] = Xs[((int64_t)bi * 64 + r) * DP + t];
  }
  for (int e = tid; e < 16 * DP; e += 256) {
    const int t = e >> 4, r = e & 15;
    XjT[e] = Xs[((int64_t)bj * 64 + 16 * q + r) * DP + t];
  }
  __syncthreads();
  const int r = tid >> 2, c0 = (tid & 3) * 4;
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
  for (int t = 0; t < DP; ++t) {
    const double xi = XiT[t * 64 + r];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const double df = xi - XjT[t * 16 + c0 + b];
      acc[b] = fma(df, df, acc[b]);
    }
  }
  const int64_t i = (int64_t)bi * 64 + r;
  double out[4];
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    const int64_t j = (int64_t)bj * 64 + 16 * q + c0 + b;
    double v;
    if (i >= N || j >= N) v = (i == j) ? 1.0 : 0.0;       // identity padding
    else if (i == j) v = 1.0 + noise;                       // unit diagonal (+ alpha, _gpr.py:347)
    else v = kernel_value<KERNEL>(acc[b]);
    out[b] = v;
  }
  double2* dst = reinterpret_cast<double2*>(K + i * NP + (int64_t)bj * 64 + 16 * q + c0);
  dst[0] = make_double2(out[0], out[1]);
  dst[1] = make_double2(out[2], out[3]);
}

int launch_kmat_q(gpbo_ctx* ctx, Model& m, double noise, double* out) {
  const int64_t nt = m.NP / 64;
  dim3 grid((unsigned)(4 * (nt * (nt + 1) / 2)), 1, (unsigned)ctx->lanes);
  const size_t lds = (size_t)m.DP * 80 * sizeof(double);
  if (m.kernel == GPBO_KERNEL_MATERN25)
    kmat_q_kernel<GPBO_KERNEL_MATERN25><<<grid, dim3(256), lds, ctx->stream>>>(m.Xs, m.DP, m.N, m.NP, noise, out, ctx->lane_stride);
  else
    kmat_q_kernel<GPBO_KERNEL_RBF><<<grid, dim3(256), lds, ctx->stream>>>(m.Xs, m.DP, m.N, m.NP, noise, out, ctx->lane_stride);
  GPBO_HIP(ctx, hipGetLastError());
  return GPBO_OK;
}

// ---- W = L^-1 by column strips -------------------------------------------------------------------------------------------------
// LDS only between the four waves: the wait covers LDS traffic alone, so the global loads requested for the NEXT product stay in
// flight across the barrier (__syncthreads() would drain them: its workgroup-scope release waits for vmcnt too).
__device__ __forceinline__ void ws_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// The strip's LDS image keeps the 64 rows of a block in the order the B fragments are read: a lane fetches its A operand as 16
// bytes = two consecutive k (half the load instructions: the CU's address unit, not the matrix pipe, would bound 8-byte fetches),
// so the k-step (h, e) of a product multiplies k = 8 h + 2 lk + e for lk = 0..3 — these four rows sit side by side at 8 h + 4 e + lk.
__device__ __forceinline__ int ws_perm(const int k) { return (k & ~7) | ((k & 1) << 2) | ((k & 7) >> 1); }

struct WStripArgs {
  const double* L; const double* dinv; const double* y;
  double* W; double* Wp; double* partial;
  int64_t N, NP, lane_stride;
  int pack;
};

// Workgroup (s, lane): strip s = columns 16 s ... 16 s + 15, inside 64-block c = s / 4.  256 threads = 4 waves; wave w owns the
// 16-row tile w of every 64-row block.  Per block row r > c:
//   T_r = sum_{t = c}^{r - 1} L_rt X_t      16 (r - c) v_mfma_f64_16x16x4_f64 per wave, two accumulators (even / odd k);
//                                           A fragments straight from L (16 bytes per lane = two k-steps, requested one product
//                                           ahead), B fragments = the strip's earlier blocks, LDS-resident as [ws_perm(k)][16]
//   X_r = -D_r T_r                          the tiles of T exchanged through LDS, D_r lower triangular by 16-tiles
// X_r goes to LDS (for the rows below) and to W.  LDS: (NP / 64 - c + 1) x 8 KiB.
__global__ __launch_bounds__(256) void w_strip_kernel(WStripArgs a) {
  extern __shared__ __attribute__((aligned(16))) double ws_smem[];
  const int tid = (int)threadIdx.x, lane = tid & 63, lr = lane & 15, lk = lane >> 4;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int s = (int)blockIdx.x, c = s >> 2, q = s & 3;
  const int64_t lo = (int64_t)blockIdx.y * a.lane_stride;
  const int64_t NP = a.NP;
  const int nblk = (int)(NP / 64);
  const double* L = a.L + lo;
  const double* dinv = a.dinv + lo;
  double* W = a.W + lo;
  double* XB = ws_smem;                         // [nblk - c][64][16]: the strip from block row c down, row-major
  double* TB = ws_smem + (nblk - c) * 1024;     // [64][16]
  const int64_t col0 = 16 * (int64_t)s;

  for (int idx = tid; idx < 64 * c * 16; idx += 256) W[(int64_t)(idx >> 4) * NP + col0 + (idx & 15)] = 0.0;   // above block row c
  for (int idx = tid; idx < 1024; idx += 256) {                                                                // X_c = D_c[:, strip]
    const int k = idx >> 4, n = idx & 15;
    const double v = dinv[(int64_t)c * 4096 + k * 64 + 16 * q + n];
    XB[ws_perm(k) * 16 + n] = v;
    W[((int64_t)c * 64 + k) * NP + col0 + n] = v;
  }
  ws_barrier();

  d2v an[8];           // A fragments of the next (r, t) product: L[64 r + 16 w + lr][64 t + 8 h + 2 lk + {0, 1}]
  auto load_a = [&](const int r, const int t) {
    const d2v* p = reinterpret_cast<const d2v*>(L + ((int64_t)64 * r + 16 * w + lr) * NP + 64 * t + 2 * lk);
#pragma unroll
    for (int h = 0; h < 8; ++h) an[h] = p[4 * h];
  };
  if (c + 1 < nblk) load_a(c + 1, c);
  for (int r = c + 1; r < nblk; ++r) {
    double dd[4][4];   // D_r[16 w + lr][16 kt + 4 g + lk], kt <= w
    {
      const double* Dr = dinv + (int64_t)r * 4096 + (16 * w + lr) * 64 + lk;
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int g = 0; g < 4; ++g) dd[kt][g] = (kt <= w) ? Dr[16 * kt + 4 * g] : 0.0;
    }
    d4 acc0 = d4{0.0, 0.0, 0.0, 0.0}, acc1 = d4{0.0, 0.0, 0.0, 0.0};
    for (int t = c; t < r; ++t) {
      d2v ac[8];
#pragma unroll
      for (int h = 0; h < 8; ++h) ac[h] = an[h];
      if (t + 1 < r) load_a(r, t + 1);
      else if (r + 1 < nblk) load_a(r + 1, c);
      const double* xb = XB + (t - c) * 1024 + lk * 16 + lr;
#pragma unroll
      for (int h = 0; h < 8; ++h) {
        acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(ac[h].x, xb[(8 * h) * 16], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(ac[h].y, xb[(8 * h + 4) * 16], acc1, 0, 0, 0);
      }
    }
    const d4 T = acc0 + acc1;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) TB[(16 * w + lk + 4 * rr) * 16 + lr] = T[rr];
    ws_barrier();
    d4 p0 = d4{0.0, 0.0, 0.0, 0.0}, p1 = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
      if (kt <= w) {
#pragma unroll
        for (int g = 0; g < 4; g += 2) {
          p0 = __builtin_amdgcn_mfma_f64_16x16x4f64(dd[kt][g], TB[(16 * kt + 4 * g + lk) * 16 + lr], p0, 0, 0, 0);
          p1 = __builtin_amdgcn_mfma_f64_16x16x4f64(dd[kt][g + 1], TB[(16 * kt + 4 * g + 4 + lk) * 16 + lr], p1, 0, 0, 0);
        }
      }
    const d4 X = -(p0 + p1);
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int row = 16 * w + lk + 4 * rr;
      XB[(r - c) * 1024 + ws_perm(row) * 16 + lr] = X[rr];
      W[((int64_t)64 * r + row) * NP + col0 + lr] = X[rr];
    }
    ws_barrier();      // X_r complete for everyone; TB free again
  }

  // the strip's contribution to t = W y: rows 64 c ... NP - 1, (row group, column) per thread, 16-lane shuffle tree
  const int nrows = (int)(NP - 64 * (int64_t)c);
  {
    const double yv = a.y[lo + col0 + (tid & 15)];
    double* part = a.partial + lo + (int64_t)s * NP + 64 * (int64_t)c;
    for (int i0 = 0; i0 < nrows; i0 += 16) {
      const int i = i0 + (tid >> 4);
      double v = XB[(i & ~63) * 16 + ws_perm(i & 63) * 16 + (tid & 15)] * yv;
      v += __shfl_xor(v, 8);
      v += __shfl_xor(v, 4);
      v += __shfl_xor(v, 2);
      v += __shfl_xor(v, 1);
      if ((tid & 15) == 0) part[i] = v;
    }
  }
  // the strip in the posterior kernels' fragment order (pack_w_elem's layout: fit_bodies.h): k-pairs 2 s and 2 s + 1 of every row slab
  if (a.pack) {
    double* Wp = a.Wp + lo;
    const int64_t pairs = NP / 8;
    for (int j = tid; j < (int)(NP * 16); j += 256) {
      const int e = j & 1, ln = (j >> 1) & 63, t = (j >> 7) & 1, pp = (j >> 8) & 1;
      const int64_t s32 = j >> 9;
      const int64_t p = 2 * (int64_t)s + pp;
      const int64_t row = 32 * s32 + 16 * t + (ln & 15);
      const int cn = 8 * pp + 4 * e + (ln >> 4);
      const int64_t colx = col0 + cn;
      double v = 0.0;
      if (row < a.N && colx < a.N && colx <= row) {
        const int i = (int)(row - 64 * (int64_t)c);
        v = XB[(i & ~63) * 16 + ws_perm(i & 63) * 16 + cn];
      }
      Wp[((s32 * pairs + p) * 2 + t) * 128 + ln * 2 + e] = v;
    }
  }
}

// t_i = sum over the strips left of and in row i's block (ascending) of their contributions; alpha_j = sum_i W_ij t_i for the
// strip's columns: 16 row classes (i mod 16) per column, combined in order.  Workgroup (0, lane) also hands the pivot word to its
// pinned host word (info_out null: the caller copies it).
__global__ __launch_bounds__(256) void alpha_strip_kernel(const double* __restrict__ W, const double* __restrict__ partial,
                                                          double* __restrict__ alpha, int64_t NP, int64_t lane_stride,
                                                          const int* __restrict__ info, int* __restrict__ info_out, int64_t info_pitch) {
  extern __shared__ __attribute__((aligned(16))) double as_smem[];
  double* tv = as_smem;           // [NP]
  double* red = as_smem + NP;     // [16][16]
  const int tid = (int)threadIdx.x;
  const int s = (int)blockIdx.x, c = s >> 2;
  const int64_t lo = (int64_t)blockIdx.y * lane_stride;
  W += lo; partial += lo;
  for (int64_t i = 64 * (int64_t)c + tid; i < NP; i += 256) {
    const int ns = 4 * (int)(i >> 6) + 4;
    double acc = 0.0;
    for (int s2 = 0; s2 < ns; ++s2) acc += partial[(int64_t)s2 * NP + i];
    tv[i] = acc;
  }
  __syncthreads();
  const int n = tid & 15, g = tid >> 4;
  double acc = 0.0;
  for (int64_t i = 64 * (int64_t)c + g; i < NP; i += 16) acc = fma(W[i * NP + 16 * (int64_t)s + n], tv[i], acc);
  red[g * 16 + n] = acc;
  __syncthreads();
  if (tid < 16) {
    double sum = 0.0;
#pragma unroll
    for (int gg = 0; gg < 16; ++gg) sum += red[gg * 16 + tid];
    alpha[lo + 16 * (int64_t)s + tid] = sum;
  }
  if (s == 0 && tid == 0 && info_out) info_out[(int64_t)blockIdx.y * info_pitch] = info[lo * 2];
}

// Largest padded size the strip path serves (fused_max_np() < NP <= mid_max_np()).  The strip's LDS image caps it at 1024; the
// default is where it stops beating the recursive inverse's GEMMs (profiles/r05_mid_fit_timing.json).  (Debug build:
// GPBO_MID_MAX_NP = 0 ... 1024 read per call, for the A/B tests and the crossover measurement.)
int mid_max_np() {
  int v = MID_NP_DEFAULT;
  if (const char* e = dbg_env("GPBO_MID_MAX_NP")) v = atoi(e);
  if (v > MID_NP_CAP) v = MID_NP_CAP;
  return v;
}

int launch_w_strip(gpbo_ctx* ctx, Model& m, bool pack) {
  if (!(ctx->func_attrs & ATTR_MID)) {
    GPBO_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(w_strip_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)((MID_NP_CAP / 64 + 1) * 1024 * sizeof(double))));
    ctx->func_attrs |= ATTR_MID;
  }
  WStripArgs a{};
  a.L = m.L; a.dinv = m.dinv; a.y = m.yn; a.W = m.W; a.Wp = m.Wp; a.partial = m.tmp;
  a.N = m.N; a.NP = m.NP; a.lane_stride = ctx->lane_stride; a.pack = (pack && m.Wp) ? 1 : 0;
  const size_t lds = (size_t)(m.NP / 64 + 1) * 1024 * sizeof(double);
  w_strip_kernel<<<dim3((unsigned)(m.NP / 16), (unsigned)ctx->lanes), dim3(256), lds, ctx->stream>>>(a);
  GPBO_HIP(ctx, hipGetLastError());
  return GPBO_OK;
}

int launch_alpha_strip(gpbo_ctx* ctx, Model& m, int* info_out, int64_t info_pitch) {
  const size_t lds = (size_t)(m.NP + 256) * sizeof(double);
  alpha_strip_kernel<<<dim3((unsigned)(m.NP / 16), (unsigned)ctx->lanes), dim3(256), lds, ctx->stream>>>(
      m.W, m.tmp, m.alpha, m.NP, ctx->lane_stride, ctx->info_dev, info_out, info_pitch);
  GPBO_HIP(ctx, hipGetLastError());
  return GPBO_OK;
}

}  // namespace gpbo
