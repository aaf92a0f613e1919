// The whole fit (or log-marginal-likelihood evaluation) of a SMALL problem as ONE launch of ONE workgroup per model (gfx950).
//
// What it replaces: GaussianProcessRegressor.fit at fixed theta and log_marginal_likelihood(theta, eval_gradient) for the sizes a
// maximize() loop lives at (sklearn _gpr.py:296-364, 575-652; bayes_opt/bayesian_optimization.py:348-391: N = 5 ... a few hundred).
//
// Why one workgroup and not a grid with barriers: up to NP = 128 the factorisation IS one workgroup (diag128_body), and everything
// around it — X / length_scale, K, W = L^-1, alpha, the W pack or the LML terms, W^T W and the gradient reduction — is a few 64x64
// tiles of work.  As ~20-30 launches of 3-27 us each (every dependent launch costs ~4 us of dispatch / drain plus ~1.5 us of
// boundary, whatever it computes) such a fit took 80 us of device time at N = 25 and an LML evaluation 130 us; an in-launch
// hand-off between workgroups costs what a launch boundary costs (agent-scope release + acquire, MI355X_MICROARCH.md "barrier-xcd":
// 4-5 us), so the phases stay on ONE compute unit, separated by s_barrier only: the CU's own L1 is coherent with its own stores.
//
// Same arithmetic, same order: every phase calls the device body the stand-alone kernel of that phase calls (fit_bodies.h,
// chol_bodies.h, gemm_tile.h, lml_bodies.h), two 256-thread virtual blocks side by side in the 512-thread workgroup, so L, W, alpha,
// the packed W, the LML value and its gradient are BITWISE what the multi-launch path produces (tests/test_gpu_fused_small.py).
// Lane mode (gpbo_lml_batch): blockIdx.x = lane, lane l's buffers l * lane_stride doubles behind lane 0's.
//
// Inputs that come from the host (length scales; X and y of gpbo_fit / gpbo_lml) are read straight out of pinned host memory, and the
// pivot word and the LML scalars are written straight into it: a fit is one launch + one stream synchronisation, no copy nodes.
#include "chol_bodies.h"
#include "fit_bodies.h"
#include "lml_bodies.h"

namespace gpbo {

constexpr int FS_HALF_LDS = 2 * GPBO_MAX_DIM * 64 + 16;       // doubles per 256-thread half: the largest body's need (K / gradient tile at DP = 64)
static_assert(FS_HALF_LDS % 2 == 0, "half regions must stay 16-byte aligned");
static_assert(2 * FS_HALF_LDS <= C128_LDS_DOUBLES, "the halves' scratch must fit the diagonal workgroup's LDS");
static_assert(FS_HALF_LDS >= GT_LDS_DOUBLES && FS_HALF_LDS >= C128_PANEL_LDS_DOUBLES && FS_HALF_LDS >= C128_UPD_LDS_DOUBLES, "half scratch too small");

// Tiles (bm, bn, bz) of one GEMM "launch" by the two halves, two at a time.  Tiles of a triangular product have different k-ranges,
// i.e. different barrier counts: the half that finishes first runs the difference as bare barriers, so that the workgroup's barrier
// count is the same for all its waves at the end of every round (the halves' LDS regions are disjoint: which barrier pairs with
// which does not matter in between).
template <bool BT, bool AT>
__device__ __forceinline__ void fs_run_tiles(const GemmArgs& g, const int tiles_m, const int tiles_n, const int batch, double* lds_half,
                                             const int half, const int t256) {
  const int per = tiles_m * tiles_n, total = per * batch;
  for (int r = 0; r < total; r += 2) {
    int cnt[2], bm[2], bn[2], bz[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int idx = r + h;
      cnt[h] = 0; bm[h] = bn[h] = bz[h] = 0;
      if (idx < total) {
        bz[h] = idx / per;
        const int rem = idx - bz[h] * per;
        bm[h] = rem / tiles_n;
        bn[h] = rem - bm[h] * tiles_n;
        cnt[h] = gemm_tile_barriers(g, bm[h], bn[h]);
      }
    }
    const int mine = half ? cnt[1] : cnt[0];
    const int most = max(cnt[0], cnt[1]);
    if (mine > 0) gemm_tile_body<BT, AT>(g, half ? bm[1] : bm[0], half ? bn[1] : bn[0], 0, half ? bz[1] : bz[0], lds_half, t256);
    for (int i = mine; i < most; ++i) __syncthreads();
  }
  __syncthreads();      // the launch boundary: every tile's stores are visible to the whole workgroup
}

struct FusedArgs {
  int64_t N, NP;
  int d, DP, n_ls, mode;          // mode 0: fit (W packed for the posterior kernels), 1: LML value, 2: LML value + gradient
  int src;                        // 0: raw X / y / length scales given; 1: Xs, yn, ls already resident in the model (refit at the same theta)
  double noise;
  const double* X;                // raw (N, d): device memory, or pinned host memory (device-visible)
  const double* y;                // (N)
  const double* ls_in;            // [lanes][GPBO_MAX_DIM] length scales (pinned host, device-visible)
  double *ls, *Xs, *K, *L, *W, *Wp, *dinv, *tmp, *yn, *tvec, *alpha, *scal;   // lane 0's buffers
  int* info;                      // lane 0's pivot word (device)
  int64_t lane_stride;            // doubles between the lanes' buffers
  int* info_out; int64_t info_pitch;      // pinned host: pivot word per lane (pitch in ints)
  double* out; int64_t out_pitch;         // pinned host: LML scalars per lane (pitch in doubles); null in fit mode
};

// Every phase starts from a FRESH copy of the thread index (an opaque move the optimiser cannot see through): left to itself the
// compiler shares the index arithmetic of all phases (lane, row, fragment offsets ...), keeps those values alive across the diagonal
// workgroup's 156 registers and spills them to scratch (first build: 13 VGPRs, 56 bytes per lane) — recomputing them costs nothing.
__device__ __forceinline__ int fs_tid() {
  int t = (int)threadIdx.x;
  asm volatile("" : "+v"(t));
  return t;
}
#define FS_THREAD()                                                              \
  const int tid = fs_tid();                                                      \
  const int half = __builtin_amdgcn_readfirstlane(tid >> 8);                     \
  const int t256 = tid & 255;                                                    \
  double* lds_half = smem + half * FS_HALF_LDS;                                  \
  (void)t256; (void)lds_half

// ---- inputs: length scales, X / length_scale (zero padded), targets (zero padded)
__device__ __forceinline__ void fs_inputs(const FusedArgs& a, const int64_t lo, const int zl) {
  const int tid = fs_tid();
  double* ls = a.ls + lo;
  if (tid == 0) a.info[lo * 2] = 0;
  if (a.src == 0) {
    if (tid < GPBO_MAX_DIM) ls[tid] = a.ls_in[(int64_t)zl * GPBO_MAX_DIM + tid];
    __syncthreads();
    for (int64_t idx = tid; idx < a.NP * a.DP; idx += 512) prescale_elem(a.X, a.N, a.d, a.DP, ls, a.Xs + lo, idx);
    for (int64_t i = tid; i < a.NP; i += 512) a.yn[lo + i] = (i < a.N) ? a.y[i] : 0.0;
  }
  __syncthreads();
}

// ---- K (lower 64x64 tiles) straight into the buffer the Cholesky factorises in place
template <int KERNEL>
__device__ __forceinline__ void fs_kmat(const FusedArgs& a, const int64_t lo, double* smem) {
  FS_THREAD();
  const int nblk = (int)(a.NP / NB);
  const int ntile = nblk * (nblk + 1) / 2;
  for (int r = 0; r < ntile; r += 2) {
    const int b = r + half;
    if (b < ntile) {
      int bi, bj;
      lower_tile_of(b, bi, bj);
      kmat_tile_body<KERNEL>(a.Xs + lo, a.DP, a.N, a.NP, a.noise, a.L + lo, bi, bj, lds_half, t256);
    } else {
      __syncthreads();
    }
    __syncthreads();     // the staging image is read until the tile's last store: nobody starts the next tile's staging before that
  }
}

// ---- what lies between two diagonal blocks of the Cholesky: panel solve below block pk = kb - 2, the update of diagonal block kb,
// the rank-128 update of the remaining columns
__device__ __forceinline__ void fs_chol_between(const FusedArgs& a, const int64_t lo, double* smem, const int kb, const int nb) {
  FS_THREAD();
  double* L = a.L + lo;
  const int64_t NP = a.NP;
  const int nblk = (int)(NP / NB);
  const int pk = kb - 2;
  const int rem = (int)(NP - (int64_t)kb * NB);               // rows below the previous diagonal block
  const int nrb = rem / 16;
  for (int r = 0; r < nrb; r += 2) {                           // panel solve, 16 rows per group
    const int blk = r + half;
    if (blk < nrb) chol128_panel_body(L, NP, pk, a.dinv + lo, blk, lds_half, t256);
    else { __syncthreads(); __syncthreads(); }
  }
  __syncthreads();
  const int nt = 4 * nb, ntl = nt * (nt + 1) / 2;
  for (int r = 0; r < ntl; r += 2) {                           // the next diagonal block brought up to date
    const int t = r + half;
    if (t < ntl) chol128_diag_update_body(L, NP, pk, t, lds_half, t256);
    else __syncthreads();
    __syncthreads();   // the partial sums are read after the body's barrier: keep the next tile's stores behind those reads
  }
  // rank-128 update of the panel's remaining columns (the tiles chol128_step_kernel runs beside the diagonal workgroup)
  double* panel = L + (int64_t)kb * NB * NP + (int64_t)pk * NB;
  GemmArgs s{};
  s.m = rem; s.n = (nblk - kb) * NB; s.k = 2 * NB; s.alpha = -1.0; s.beta = 1.0;
  s.A = panel; s.lda = NP; s.B = panel; s.ldb = NP; s.b_trans = 1;
  s.C = L + (int64_t)kb * NB * NP + (int64_t)kb * NB; s.ldc = NP;
  s.lower_only = 1; s.skip00 = nb; s.batch = 1; s.lanes = 1; s.lane_stride = 0;
  fs_run_tiles<true, false>(s, s.m / 64, s.n / 64, 1, lds_half, half, t256);
}

// ---- W = L^-1: zero fill, diagonal blocks, then trtri's levels (two GEMMs per level and part)
__device__ __forceinline__ void fs_trtri(const FusedArgs& a, const int64_t lo, double* smem) {
  FS_THREAD();
  double* L = a.L + lo; double* W = a.W + lo; double* tmp = a.tmp + lo;
  const int64_t NP = a.NP;
  const int nblk = (int)(NP / NB);
  double2* W2 = reinterpret_cast<double2*>(W);
  const double2 z = make_double2(0.0, 0.0);
  for (int64_t idx = tid; idx < NP * NP / 2; idx += 512) W2[idx] = z;
  __syncthreads();
  for (int r = 0; r < nblk; r += 2)
    if (r + half < nblk) fill_w_diag_body(a.dinv + lo, W, NP, r + half, t256);
  __syncthreads();
  for (int64_t b = NB; b < NP; b *= 2) {
    const int64_t full = NP / (2 * b);
    const int64_t rag = NP - full * 2 * b;
    for (int part = 0; part < 2; ++part) {
      int64_t npairs, b2, r0;
      if (part == 0) { npairs = full; b2 = b; r0 = 0; }
      else { npairs = (rag > b) ? 1 : 0; b2 = rag - b; r0 = full * 2 * b; }
      if (npairs == 0) continue;
      GemmArgs t{};   // T = L21 * W11
      t.m = (int)b2; t.n = (int)b; t.k = (int)b; t.alpha = 1.0; t.beta = 0.0;
      t.A = L + (r0 + b) * NP + r0; t.lda = NP; t.strideA = 2 * b * NP + 2 * b;
      t.B = W + r0 * NP + r0; t.ldb = NP; t.strideB = 2 * b * NP + 2 * b; t.b_lower = 1;
      t.C = tmp; t.ldc = b; t.strideC = b * b; t.batch = (int)npairs; t.lanes = 1;
      fs_run_tiles<false, false>(t, t.m / 64, t.n / 64, t.batch, lds_half, half, t256);
      GemmArgs w{};   // W21 = -W22 * T
      w.m = (int)b2; w.n = (int)b; w.k = (int)b2; w.alpha = -1.0; w.beta = 0.0;
      w.A = W + (r0 + b) * NP + (r0 + b); w.lda = NP; w.strideA = 2 * b * NP + 2 * b; w.a_lower = 1;
      w.B = tmp; w.ldb = b; w.strideB = b * b;
      w.C = W + (r0 + b) * NP + r0; w.ldc = NP; w.strideC = 2 * b * NP + 2 * b; w.batch = (int)npairs; w.lanes = 1;
      fs_run_tiles<false, false>(w, w.m / 64, w.n / 64, w.batch, lds_half, half, t256);
    }
  }
}

// ---- alpha = W^T (W y)
__device__ __forceinline__ void fs_alpha(const FusedArgs& a, const int64_t lo, double* smem) {
  FS_THREAD();
  const double* W = a.W + lo;
  double* tvec = a.tvec + lo; double* tmp = a.tmp + lo;
  const int64_t NP = a.NP;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  for (int64_t i0 = 0; i0 < NP; i0 += 8) trmv_lower_row(W, a.yn + lo, tvec, NP, i0 + wave, lane);
  __syncthreads();
  const int nbx = (int)(NP / NB), total = nbx * TRMV_SPLITS;       // even
  for (int r = 0; r < total; r += 2) {
    const int idx = r + half;
    trmv_lower_t_body(W, tvec, tmp, NP, idx % nbx, idx / nbx, lds_half, t256);
    __syncthreads();     // (the reduction rows are read after the body's barrier)
  }
  for (int64_t j = tid; j < NP; j += 512) trmv_reduce_elem(tmp, a.alpha + lo, NP, j);
  __syncthreads();
}

// ---- LML terms; K^-1 = W^T W (lower tiles, into the K buffer); the gradient reduction; the scalars to the host words
template <int KERNEL>
__device__ __forceinline__ void fs_lml(const FusedArgs& a, const int64_t lo, const int zl, double* smem) {
  FS_THREAD();
  const int64_t NP = a.NP;
  const int nblk = (int)(NP / NB);
  double* scal = a.scal + lo;
  double* Km = a.K + lo; double* tmp = a.tmp + lo;
  lml_terms_body(a.yn + lo, a.alpha + lo, a.L + lo, a.N, NP, scal, lds_half, t256, half == 0);
  if (a.mode == 2) {
    GemmArgs g{};
    g.m = (int)NP; g.n = (int)NP; g.k = (int)NP; g.alpha = 1.0; g.beta = 0.0;
    g.A = a.W + lo; g.lda = NP; g.a_trans = 1;
    g.B = a.W + lo; g.ldb = NP;
    g.C = Km; g.ldc = NP; g.batch = 1; g.lower_only = 1; g.k_from_tile = 1; g.lanes = 1;
    fs_run_tiles<false, true>(g, nblk, nblk, 1, lds_half, half, t256);
    const int ntile = nblk * (nblk + 1) / 2;
    for (int r = 0; r < ntile; r += 2) {
      const int b = r + half;
      int bi = 0, bj = 0;
      if (b < ntile) lower_tile_of(b, bi, bj);
      lml_grad_tile_body<KERNEL>(a.Xs + lo, a.DP, a.n_ls, a.N, NP, a.alpha + lo, Km, tmp, bi, bj, lds_half, t256, b < ntile);
    }
    __syncthreads();
    for (int r = 0; r < a.n_ls; r += 2) {
      const int t = r + half;
      lml_grad_final_body(tmp, ntile, a.n_ls, scal + 2, t < a.n_ls ? t : 0, lds_half, t256, t < a.n_ls);
    }
  }
  __syncthreads();
  const int nout = 2 + (a.mode == 2 ? a.n_ls : 0);
  if (tid < nout) a.out[(int64_t)zl * a.out_pitch + tid] = scal[tid];
}

template <int KERNEL>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void fused_small_kernel(FusedArgs a) {
  extern __shared__ __attribute__((aligned(16))) double fs_smem[];
  const int zl = (int)blockIdx.x;
  const int64_t lo = (int64_t)zl * a.lane_stride;
  const int nblk = (int)(a.NP / NB);
  fs_inputs(a, lo, zl);
  fs_kmat<KERNEL>(a, lo, fs_smem);
  // Cholesky: launch_cholesky128's schedule with ONE outer panel (what chol_outer_width gives up to NP = 2048), every launch of it a
  // phase of this workgroup
  for (int kb = 0; kb < nblk; kb += 2) {
    const int nb = (nblk - kb >= 2) ? 2 : 1;
    if (kb > 0) fs_chol_between(a, lo, fs_smem, kb, nb);
    diag128_body(a.L + lo, a.NP, kb, nb, a.dinv + lo, a.info + lo * 2, fs_smem, nullptr, fs_tid());
    __syncthreads();
  }
  fs_trtri(a, lo, fs_smem);
  fs_alpha(a, lo, fs_smem);
  if (a.mode == 0) {
    // W in the posterior kernels' fragment order
    const int tid = fs_tid();
    for (int64_t idx = tid; idx < a.NP * a.NP; idx += 512) pack_w_elem(a.W + lo, a.Wp + lo, a.N, a.NP, idx);
  } else {
    fs_lml<KERNEL>(a, lo, zl, fs_smem);
  }
  if (fs_tid() == 0) a.info_out[(int64_t)zl * a.info_pitch] = a.info[lo * 2];
}

// Largest padded size the fused kernel serves: 64 in the product.  (At NP = 128 — still one diagonal workgroup plus a handful of
// tiles — the strip path of mid_fit.hip is faster, 50 against 79 us per fit and 94 against 118 per LML + gradient; at NP = 64 the
// one launch wins the LML evaluation, 67 against 74 us, and loses 5 us on the fit: profiles/r05_small_fit_timing.json.  Debug
// build: GPBO_FUSED_MAX_NP = 0 / 64 / 128 / ... up to FUSED_NP_CAP, read per call, for the bitwise A/B tests and the crossover
// measurement.)
int fused_max_np() {
  int v = FUSED_NP_DEFAULT;
  if (const char* e = dbg_env("GPBO_FUSED_MAX_NP")) v = atoi(e);
  if (v > FUSED_NP_CAP) v = FUSED_NP_CAP;
  return v;
}

int launch_fused_small(gpbo_ctx* ctx, Model& m, int mode, int src, int n_ls, const double* X, const double* y, const double* ls_in,
                       double* scal, int* info_out, int64_t info_pitch, double* out, int64_t out_pitch) {
  if (!(ctx->func_attrs & ATTR_FUSED)) {
    GPBO_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(fused_small_kernel<GPBO_KERNEL_MATERN25>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)C128_LDS_BYTES));
    GPBO_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(fused_small_kernel<GPBO_KERNEL_RBF>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)C128_LDS_BYTES));
    ctx->func_attrs |= ATTR_FUSED;
  }
  FusedArgs a{};
  a.N = m.N; a.NP = m.NP; a.d = m.d; a.DP = m.DP; a.n_ls = n_ls; a.mode = mode; a.src = src; a.noise = m.noise;
  a.X = X; a.y = y; a.ls_in = ls_in;
  a.ls = m.ls; a.Xs = m.Xs; a.K = m.K; a.L = m.L; a.W = m.W; a.Wp = m.Wp; a.dinv = m.dinv; a.tmp = m.tmp; a.yn = m.yn;
  a.tvec = m.tvec; a.alpha = m.alpha; a.scal = scal;
  a.info = ctx->info_dev;
  a.lane_stride = ctx->lane_stride;
  a.info_out = info_out; a.info_pitch = info_pitch; a.out = out; a.out_pitch = out_pitch;
  const dim3 grid((unsigned)ctx->lanes), block(512);
  if (m.kernel == GPBO_KERNEL_MATERN25)
    fused_small_kernel<GPBO_KERNEL_MATERN25><<<grid, block, C128_LDS_BYTES, ctx->stream>>>(a);
  else
    fused_small_kernel<GPBO_KERNEL_RBF><<<grid, block, C128_LDS_BYTES, ctx->stream>>>(a);
  GPBO_HIP(ctx, hipGetLastError());
  return GPBO_OK;
}

}  // namespace gpbo
