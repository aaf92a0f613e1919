// The fit / log-marginal-likelihood evaluation of a MID-SIZE problem (fused_max_np() < NP <= mid_max_np(), i.e. 65 ... 768
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
// LDS only between the waves: the wait covers LDS traffic alone, so the global loads requested for the NEXT product stay in
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

// Workgroup (s, lane): strip s = columns 16 s ... 16 s + 15, inside 64-block c = s / 4.  Row r = c + 1 ... nblk - 1 is a sequence of
// r - c + 1 products of a 64x64 block with a 64x16 block: t = c ... r - 1 multiply L_rt (from L) with X_t, the last one (t = r)
// multiplies D_r (from dinv, lower triangular by 16-tiles) with the row's sum T_r and gives X_r = -D_r T_r, which goes to LDS
// (for the rows below) and to W.  LDS: (NP / 64 - c) x 8 KiB for the strip + 2 x 32 KiB of exchange (one area beyond NP = 768).
//
// 1024 threads = 16 waves = FOUR per SIMD: wave w owns the 16-row tile ti = w & 3 of every block and the k-quarter kq = w >> 2
// (k-steps h = 2 kq, 2 kq + 1) of every product — 4 v_mfma_f64_16x16x4_f64 per wave and product, two accumulators (one wave per
// SIMD cannot feed the fp64 matrix pipe: gpbo_mfma_f64_probe, 140 / 102 / 63 cycles per MFMA with 1 / 2 / 4 waves).  The
// k-quarters of a row's sum meet in LDS (fixed order 0..3), so do those of D_r T_r.
//
// All operand addresses are known up front, and L has just been written by other compute units: the A fragments run WS_AHEAD
// products ahead in a ring of register sets (static indices: the sequence is walked in groups of WS_AHEAD + 1), across row ends
// and their LDS barriers.  Every request is the same two 16-byte loads whatever it fetches (a request past the end repeats the
// last one) and sits in straight-line code: the compiler counts the loads in flight exactly and waits for the oldest set only
// (loads issued under a branch make its waitcnt pass fall back to vmcnt(0)).
//
// What the kernel costs and why (NP = 512, the c = 0 strips: 35 products = 15 us of matrix-pipe time; measured 42 us, round 5): a
// product step takes ~1 400 cycles (1 024 of pipe) and a row end ~5 500 cycles of five synchronisation points.  Measured on the way
// and NOT the bound: operand latency (look-ahead 1 / 3 / 5, an up-front touch of everything the strip reads: same time), the wave
// count by itself (4 waves with 16 MFMAs each: same time), the address arithmetic (flat 64-bit addresses: 491 k VALU instructions
// against 26 k MFMAs per launch; buffer descriptors: a tenth of that, same time), stores inside the loop.  docs/LAB_NOTEBOOK.md §9.2.
constexpr int WS_THREADS = 1024;
constexpr int WS_EXCH = 4 * 1024;       // doubles: [k-quarter][64 rows in image order][16]
__global__ __launch_bounds__(WS_THREADS) void w_strip_kernel(WStripArgs a) {
  extern __shared__ __attribute__((aligned(16))) double ws_smem[];
  const int tid = (int)threadIdx.x, lane = tid & 63, lr = lane & 15, lk = lane >> 4;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ti = w & 3, kq = w >> 2;
  const int s = (int)blockIdx.x, c = s >> 2, q = s & 3;
  const int64_t lo = (int64_t)blockIdx.y * a.lane_stride;
  const int64_t NP = a.NP;
  const int nblk = (int)(NP / 64);
  const double* L = a.L + lo;
  const double* dinv = a.dinv + lo;
  double* W = a.W + lo;
  double* XB = ws_smem;                         // [nblk - c][64][16]: the strip from block row c down, rows in image order (ws_perm)
  double* EX = ws_smem + (nblk - c) * 1024;     // [4][64][16]: the k-quarters' partial sums of T_r
  // ... and of D_r T_r: an area of its own where the 160 KiB allow it (NP <= 768), else the same one behind one more barrier
  const bool ex_alias = nblk > 12;
  double* EX2 = ex_alias ? EX : EX + WS_EXCH;
  const int64_t col0 = 16 * (int64_t)s;

  for (int idx = tid; idx < 64 * c * 16; idx += WS_THREADS) W[(int64_t)(idx >> 4) * NP + col0 + (idx & 15)] = 0.0;   // above block row c
  {                                                                                                                  // X_c = D_c[:, strip]
    const int k = tid >> 4, n = tid & 15;
    const double v = dinv[(int64_t)c * 4096 + k * 64 + 16 * q + n];
    XB[ws_perm(k) * 16 + n] = v;
  }
  ws_barrier();

  constexpr int WS_AHEAD = 3, WS_RING = WS_AHEAD + 1;     // (5: no faster — 43.8 vs 42.0 us at NP = 512; 7: spills)
  d2v abuf[WS_RING][2];    // the wave's A fragments of a product: block[16 ti + lr][8 h + 2 lk + {0, 1}], h = 2 kq, 2 kq + 1
  // Both operand sources are read through buffer descriptors — a wave-uniform base in SGPRs, the lane's constant 32-bit offset, the
  // walk over the blocks as the instruction's scalar offset — so a request is a few scalar instructions and two loads (flat 64-bit
  // per-lane addresses: ~70 VALU and ~50 SALU instructions per product next to its 4 MFMAs).
  constexpr int WS_BUF_FLAGS = 0x00020000;       // gfx9 raw buffer descriptor word 3
  const __amdgpu_buffer_rsrc_t rs_l = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(L), 0, 0x7fffffff, WS_BUF_FLAGS);
  const __amdgpu_buffer_rsrc_t rs_d = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(dinv), 0, 0x7fffffff, WS_BUF_FLAGS);
  const unsigned voff_l = (unsigned)(((16 * ti + lr) * (int)NP + 2 * lk + 16 * kq) * 8);
  const unsigned voff_d = (unsigned)(((16 * ti + lr) * 64 + 2 * lk + 16 * kq) * 8);
  int rq_r = c + 1, rq_t = c;                    // the next product to REQUEST
  auto request = [&](d2v (&dst)[2]) {
    const int rr = min(rq_r, nblk - 1), tt = (rq_r < nblk) ? rq_t : nblk - 1;
    const bool is_d = tt == rr;
    const unsigned soff = is_d ? (unsigned)rr * 4096u * 8u : ((unsigned)(64 * rr) * (unsigned)NP + 64u * (unsigned)tt) * 8u;
    const unsigned voff = is_d ? voff_d : voff_l;
    if (is_d) {
      dst[0] = __builtin_bit_cast(d2v, __builtin_amdgcn_raw_buffer_load_b128(rs_d, voff, soff, 0));
      dst[1] = __builtin_bit_cast(d2v, __builtin_amdgcn_raw_buffer_load_b128(rs_d, voff + 64u, soff, 0));
    } else {
      dst[0] = __builtin_bit_cast(d2v, __builtin_amdgcn_raw_buffer_load_b128(rs_l, voff, soff, 0));
      dst[1] = __builtin_bit_cast(d2v, __builtin_amdgcn_raw_buffer_load_b128(rs_l, voff + 64u, soff, 0));
    }
    if (rq_r < nblk && ++rq_t > rq_r) { ++rq_r; rq_t = c; }
  };
#pragma unroll
  for (int j = 0; j < WS_AHEAD; ++j) request(abuf[j]);
  int r = c + 1, t = c;                          // the product being multiplied
  d4 acc0 = d4{0.0, 0.0, 0.0, 0.0}, acc1 = d4{0.0, 0.0, 0.0, 0.0};
  const int boff = (16 * kq + lk) * 16 + lr;     // the wave's B fragment (h, e) of a [64][16] image: boff + (8 (h - 2 kq) + 4 e) * 16
  // A row's end is a dependent chain — the k-quarters of T_r meet in LDS, D_r T_r, its k-quarters meet in LDS, X_r — that only the
  // LAST product of the next row waits for (t = r needs X_r; t < r does not).  So the chain is spread over the next row's first
  // steps instead of standing between the rows with the matrix pipe idle (round 5: 5 000 cycles per row end, 42 % of the loop; spread
  // out: 47.3 -> 42.0 us per launch at NP = 512 — the chain itself, ~4.8 us per row, is what is left):
  //   D step of row r:        T partials -> EX | barrier | D_r T_r by k-quarter -> EX2        (pend_sum: X_r still to be summed)
  //   first product of r + 1: its MFMAs | barrier | waves kq = 0 sum the k-quarters -> X_r    (pend_x: X_r not yet visible to all)
  //   last product of r + 1:  barrier | its MFMAs
  // All flags are the same in every wave (they follow r and t), so every wave meets the same barriers in the same order.
  bool pend_sum = false, pend_x = false;
  int pend_row = 0;
  auto sum_x = [&]() {     // waves kq = 0: row tile ti of X_{pend_row} = -(k-quarters 0 .. ti in order) -> the strip image
    if (kq == 0) {
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int row = 16 * ti + lk + 4 * rr;
        double v = EX2[row * 16 + lr];
        for (int qq = 1; qq <= ti; ++qq) v += EX2[qq * 1024 + row * 16 + lr];
        v = -v;
        XB[(pend_row - c) * 1024 + ws_perm(row) * 16 + lr] = v;
      }
    }
  };
  while (r < nblk) {
#pragma unroll
    for (int j = 0; j < WS_RING; ++j) {
      request(abuf[(j + WS_AHEAD) % WS_RING]);      // unconditional (straight-line code: the loads in flight can be counted)
      if (r < nblk) {
        if (t < r) {       // T_r += L_rt X_t (this wave: its k-quarter)
          if (pend_x && t == r - 1) {              // the one product that multiplies X_{r-1}
            ws_barrier();
            pend_x = false;
          }
          const double* xb = XB + (t - c) * 1024 + boff;
          const double b0 = xb[0], b1 = xb[4 * 16], b2 = xb[8 * 16], b3 = xb[12 * 16];
          acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(abuf[j][0].x, b0, acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(abuf[j][0].y, b1, acc1, 0, 0, 0);
          acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(abuf[j][1].x, b2, acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(abuf[j][1].y, b3, acc1, 0, 0, 0);
          ++t;
          if (pend_sum) {                          // the previous row's partial products are all in EX2
            ws_barrier();
            pend_sum = false;
            sum_x();
            pend_x = true;
          }
        } else {           // X_r = -D_r T_r, first half
          {
            const d4 part = acc0 + acc1;
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) EX[kq * 1024 + ws_perm(16 * ti + lk + 4 * rr) * 16 + lr] = part[rr];
          }
          ws_barrier();
          d4 p0 = d4{0.0, 0.0, 0.0, 0.0}, p1 = d4{0.0, 0.0, 0.0, 0.0};
          if (kq <= ti) {  // (D_r is lower triangular by 16-tiles: k-quarter kq only reaches the row tiles ti >= kq)
            double bt[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const double* ex = EX + boff + (4 * e) * 16;
              bt[e] = ((ex[0] + ex[1024]) + ex[2048]) + ex[3072];       // T_r: the four k-quarters in order
            }
            p0 = __builtin_amdgcn_mfma_f64_16x16x4f64(abuf[j][0].x, bt[0], p0, 0, 0, 0);
            p1 = __builtin_amdgcn_mfma_f64_16x16x4f64(abuf[j][0].y, bt[1], p1, 0, 0, 0);
            p0 = __builtin_amdgcn_mfma_f64_16x16x4f64(abuf[j][1].x, bt[2], p0, 0, 0, 0);
            p1 = __builtin_amdgcn_mfma_f64_16x16x4f64(abuf[j][1].y, bt[3], p1, 0, 0, 0);
          }
          if (ex_alias) ws_barrier();              // (one exchange area only: everybody has read the T partials first)
          if (kq <= ti) {
            const d4 part = p0 + p1;
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) EX2[kq * 1024 + (16 * ti + lk + 4 * rr) * 16 + lr] = part[rr];
          }
          pend_sum = true;
          pend_row = r;
          acc0 = d4{0.0, 0.0, 0.0, 0.0};
          acc1 = d4{0.0, 0.0, 0.0, 0.0};
          ++r;
          t = c;
        }
      }
    }
  }
  if (pend_sum) {          // the last row
    ws_barrier();
    sum_x();
  }
  ws_barrier();            // the strip image is complete for everyone

  // W from the strip image, in one pass at the end.  (No global store inside the product loop: the wave's memory counter is
  // in-order, so a store issued between two operand requests makes the wait for the younger request a wait for the store's
  // acknowledgement — ~2 us per row on the four summing waves, and everybody else meets them at the next barrier.)
  for (int idx = tid; idx < (int)(NP - 64 * (int64_t)c) * 16; idx += WS_THREADS) {
    const int i = idx >> 4, n = idx & 15;
    W[((int64_t)64 * c + i) * NP + col0 + n] = XB[(i & ~63) * 16 + ws_perm(i & 63) * 16 + n];
  }
  // the strip's contribution to t = W y: rows 64 c ... NP - 1, (row group, column) per thread, 16-lane shuffle tree
  const int nrows = (int)(NP - 64 * (int64_t)c);
  {
    const double yv = a.y[lo + col0 + (tid & 15)];
    double* part = a.partial + lo + (int64_t)s * NP + 64 * (int64_t)c;
    for (int i0 = 0; i0 < nrows; i0 += WS_THREADS / 16) {
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
    for (int j = tid; j < (int)(NP * 16); j += WS_THREADS) {
      const int e = j & 1, ln = (j >> 1) & 63, t2 = (j >> 7) & 1, pp = (j >> 8) & 1;
      const int64_t s32 = j >> 9;
      const int64_t p = 2 * (int64_t)s + pp;
      const int64_t row = 32 * s32 + 16 * t2 + (ln & 15);
      const int cn = 8 * pp + 4 * e + (ln >> 4);
      const int64_t colx = col0 + cn;
      double v = 0.0;
      if (row < a.N && colx < a.N && colx <= row) {
        const int i = (int)(row - 64 * (int64_t)c);
        v = XB[(i & ~63) * 16 + ws_perm(i & 63) * 16 + cn];
      }
      Wp[((s32 * pairs + p) * 2 + t2) * 128 + ln * 2 + e] = v;
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
    const int ns = 4 * (int)(i >> 6) + 4;      // a multiple of 4: four loads in flight per step, added in ascending order
    double acc = 0.0;
    for (int s2 = 0; s2 < ns; s2 += 4) {
      const double p0 = partial[(int64_t)s2 * NP + i], p1 = partial[(int64_t)(s2 + 1) * NP + i];
      const double p2 = partial[(int64_t)(s2 + 2) * NP + i], p3 = partial[(int64_t)(s2 + 3) * NP + i];
      acc += p0; acc += p1; acc += p2; acc += p3;
    }
    tv[i] = acc;
  }
  __syncthreads();
  const int n = tid & 15, g = tid >> 4;
  double acc = 0.0;
  {
    // rows in steps of 16 from block row c down: NP - 64 c is a multiple of 64, i.e. the trip count a multiple of 4
    const double* wp = W + 16 * (int64_t)s + n;
    for (int64_t i = 64 * (int64_t)c + g; i < NP; i += 64) {
      const double w0 = wp[i * NP], w1 = wp[(i + 16) * NP], w2 = wp[(i + 32) * NP], w3 = wp[(i + 48) * NP];
      acc = fma(w0, tv[i], acc);
      acc = fma(w1, tv[i + 16], acc);
      acc = fma(w2, tv[i + 32], acc);
      acc = fma(w3, tv[i + 48], acc);
    }
  }
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
// default (768) is where one lane stops beating the multi-launch path (profiles/r05_small_fit_timing.json: fit 0.31 vs 0.39 ms at
// 768, 0.44 vs 0.49 at 1024 but a resident lane 0.52 vs 0.51 there).  (Debug build:
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
                                      (int)(160 * 1024)));
    ctx->func_attrs |= ATTR_MID;
  }
  WStripArgs a{};
  a.L = m.L; a.dinv = m.dinv; a.y = m.yn; a.W = m.W; a.Wp = m.Wp; a.partial = m.tmp;
  a.N = m.N; a.NP = m.NP; a.lane_stride = ctx->lane_stride; a.pack = (pack && m.Wp) ? 1 : 0;

  const size_t lds = (size_t)(m.NP / 64 * 1024 + (m.NP / 64 > 12 ? 1 : 2) * WS_EXCH) * sizeof(double);
  w_strip_kernel<<<dim3((unsigned)(m.NP / 16), (unsigned)ctx->lanes), dim3(WS_THREADS), lds, ctx->stream>>>(a);
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
