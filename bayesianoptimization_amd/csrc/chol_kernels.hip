// Blocked Cholesky, round-3 schedule (gfx950): 128-column steps inside 512-column outer panels.
//
// What it replaces: cholesky(K, lower=True) -> LAPACK dpotrf   (sklearn _gpr.py:349, via scipy.linalg.cholesky)
//
// Why 128: the factorisation is a latency chain, not a flop problem (round 2: 0.68 us per column at every N, the
// matrix pipe 2-5 % busy on the chain kernels).  fp64 runs at 32 flop/clk/SIMD on VALU and MFMA alike, so ONE compute
// unit needs 1.7 us for a 64^3 product: whatever sits on the chain must be small, and everything else must be wide.
// Per 128 columns the chain is now
//     diagonal block (ONE workgroup, no launch inside: potf2 of 128x64 -> SYRK -> potf2 of 64x64 -> the two 64x64 inverses)
//  -> panel solve of ALL rows below (16 rows per workgroup, three chained 16-row MFMA products, no explicit 128x128 inverse)
//  -> rank-128 update of the NEXT diagonal block only (36 small workgroups)
//  -> [next diagonal block || rest of the in-panel update] in one launch
// i.e. three launches per 128 columns where round 2 had two per 64, and the diagonal workgroup no longer applies the
// previous column block's update to its own block by itself (that 64^3 product cost 1.7 us per step on one CU; for a
// 128-block it would be 8.5 us).  The rank-512 trailing update per outer panel is unchanged (gemm128_f64_kernel).
//
// Every kernel here takes a lane index (blockIdx.y / .z) so that gpbo_lml_batch's theta lanes share the launches.
#include <cstdlib>

#include "chol_bodies.h"
#include "fit_bodies.h"

namespace gpbo {

// One launch = diagonal block(s) kb (workgroup 0) || the 64x64 tiles of the previous column block's in-panel update, two per
// 512-thread workgroup (every tile but the leading skip00 x skip00 ones, which chol128_diag_update_kernel has already
// brought up to date).
// (waves_per_eu(2, 2): the launch's dynamic LDS gives every workgroup a CU to itself, so 256 VGPRs are there for the taking — 156 used)
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void chol128_step_kernel(double* L, int64_t ld, int kb, int nblk, double* dinv, int* info,
                                                            GemmArgs g, int tiles_n, int tiles, int live, int64_t lane_stride, long long* stamps) {
  extern __shared__ __attribute__((aligned(16))) double c128_smem[];
  const int zl = (int)blockIdx.y;
  if (blockIdx.x == 0) {
    diag128_body(L + (int64_t)zl * lane_stride, ld, kb, nblk, dinv + (int64_t)zl * lane_stride, info + (int64_t)zl * lane_stride * 2,
                 c128_smem, stamps, (int)threadIdx.x);
    return;
  }
  const int half = (int)(threadIdx.x >> 8);
  const int t = 2 * ((int)blockIdx.x - 1) + half;
  int bm, bn;
  bool mine;
  if (live) {
    // LIVE tiles only (round 6), in the order [lower triangle of the leading tiles_n x tiles_n tiles, by rows | the full rows below it],
    // the leading skip00 (skip00 + 1) / 2 of which belong to chol128_diag_update_kernel.  The square grid's dead upper tiles were
    // not free: a dead tile goes through the loads and barriers of a live one beside its partner half, so a one-panel factorisation
    // (NP <= 2048) ran twice the workgroups it needed — 392 for 203 live pairs at the second step of N = 2048: two rounds over the
    // chip's 256 CUs (the launch's dynamic LDS admits one workgroup per CU), 39 us against the diagonal workgroup's 23.
    const int tt = t + g.skip00 * (g.skip00 + 1) / 2;
    const int tri = tiles_n * (tiles_n + 1) / 2;
    if (tt < tri) lower_tile_of(tt, bm, bn);
    else { bm = tiles_n + (tt - tri) / tiles_n; bn = (tt - tri) - (bm - tiles_n) * tiles_n; }
    mine = t < tiles;
  } else {
    bm = t / tiles_n; bn = t - bm * tiles_n;
    mine = t < tiles && !(g.lower_only && bn > bm) && !(bn < g.skip00 && bm < g.skip00);
  }
  if (!mine) { bm = g.m / 64 - 1; bn = 0; }   // a tile that exists (the last row tile): loads only, same barrier count
  GemmArgs h = g;
  h.lower_only = 0; h.skip00 = 0;            // decided above
  gemm_tile_body<true, false>(h, bm, bn, zl, 0, c128_smem + half * GT_LDS_DOUBLES, (int)(threadIdx.x & 255), mine);
}

// Panel solve below a 128-column diagonal block (16 rows per workgroup) and the rank-128 update of the NEXT diagonal block (one
// 16x16 tile per workgroup): the bodies live in chol_bodies.h (the fused small-problem kernel runs them too).
__global__ __launch_bounds__(256) void chol128_panel_kernel(double* L, int64_t ld, int kb, const double* __restrict__ dinv,
                                                             int64_t lane_stride) {
  __shared__ __attribute__((aligned(16))) double XT[C128_PANEL_LDS_DOUBLES];
  chol128_panel_body(L + (int64_t)blockIdx.y * lane_stride, ld, kb, dinv + (int64_t)blockIdx.y * lane_stride, (int)blockIdx.x, XT,
                     (int)threadIdx.x);
}

__global__ __launch_bounds__(256) void chol128_diag_update_kernel(double* L, int64_t ld, int kb, int64_t lane_stride) {
  __shared__ __attribute__((aligned(16))) double Ps[C128_UPD_LDS_DOUBLES];
  chol128_diag_update_body(L + (int64_t)blockIdx.y * lane_stride, ld, kb, (int)blockIdx.x, Ps, (int)threadIdx.x);
}

// ---- the 128-column step as ONE launch (round 6 experiment; debug build: GPBO_CHOL_FUSED_STEP=1) -----------------------------
// [diagonal block kb | the previous step's update tiles | panel solve of all rows below | update of the next diagonal block] in one
// grid, block ids in exactly this order, the stages handed on through counters in memory instead of kernel boundaries:
//   flags[0]      the diagonal workgroup has stored L_kk and its two inverses                      (-> every panel group)
//   flags[8 + bm] update tiles (bm, 0) and (bm, 1) of this launch are stored                        (-> the panel groups of row tile bm)
//   flags[2]      update tiles of the next diagonal block ((2,2), (3,2), (3,3)) are stored          (-> the next-diagonal update)
//   flags[1]      panel groups 0 .. 7 (the next diagonal block's rows) are stored                   (-> the next-diagonal update)
// A producer stores, __syncthreads, one lane: release fence (agent) + s_waitcnt vmcnt(0) + relaxed atomic add.  A consumer: one lane
// polls with relaxed agent loads + s_sleep, acquire fence (agent), __syncthreads (MI355X_MICROARCH.md, valid forms).  Every wait
// points at LOWER block ids, which are dispatched first and never wait upwards; every spin is bounded (a broken hand-off surfaces
// as info = -2000 - kb, never as a hang).  The bodies are the three launches' bodies: same bits.
// What it was built to find out (VERDICT r5 next #1b): whether consumers that are already resident — operands
// that do not depend on the diagonal block prefetched, no launch ramp — beat the two kernel boundaries of a step.
constexpr int CF_HEAD = 8;      // flag words in front of the row counters
__device__ __forceinline__ void cf_publish(int* word) {        // one lane, behind a __syncthreads() that follows the stores
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __hip_atomic_fetch_add(word, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bool cf_wait(const int* word, int want) {      // one lane; false: gave up (~30 ms)
  int spins = 0;
  while (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
    if (++spins > (1 << 19)) return false;
    __builtin_amdgcn_s_sleep(2);
  }
  return true;
}

__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void chol128_fused_step_kernel(
    double* L, int64_t ld, int kb, int nblk, double* dinv, int* info, GemmArgs g, int tiles_n, int tiles, int n_tile_wgs, int n_groups,
    int n_panel_wgs, int nt_next, int64_t lane_stride, int* flags0, long long* stamps) {
  extern __shared__ __attribute__((aligned(16))) double c128_smem[];
  __shared__ int cf_ok;
  const int zl = (int)blockIdx.y;
  double* Lz = L + (int64_t)zl * lane_stride;
  double* dz = dinv + (int64_t)zl * lane_stride;
  int* iz = info + (int64_t)zl * lane_stride * 2;
  int* flags = flags0 + (int64_t)zl * lane_stride * 2;
  const int bid = (int)blockIdx.x, tid = (int)threadIdx.x;
  if (bid == 0) {
    // (stamps 8 .. 13: the first step's hand-off times on the 100 MHz wall clock every CU shares — s_memtime counters differ between CUs)
    diag128_body(Lz, ld, kb, nblk, dz, iz, c128_smem, stamps, tid);
    __syncthreads();
    if (tid == 0) {
      if (stamps) stamps[14] = wall_clock64();
      cf_publish(flags + 0);
      if (stamps) stamps[8] = wall_clock64();
    }
    return;
  }
  const int half = tid >> 8, t256 = tid & 255;
  if (bid <= n_tile_wgs) {
    const int t = 2 * (bid - 1) + half;
    const int tt = t + g.skip00 * (g.skip00 + 1) / 2;
    const int tri = tiles_n * (tiles_n + 1) / 2;
    int bm, bn;
    if (tt < tri) lower_tile_of(tt, bm, bn);
    else { bm = tiles_n + (tt - tri) / tiles_n; bn = (tt - tri) - (bm - tiles_n) * tiles_n; }
    const bool mine = t < tiles;
    const int bm0 = bm, bn0 = bn;
    if (!mine) { bm = g.m / 64 - 1; bn = 0; }
    GemmArgs h = g;
    h.lower_only = 0; h.skip00 = 0;
    gemm_tile_body<true, false>(h, bm, bn, zl, 0, c128_smem + half * GT_LDS_DOUBLES, t256, mine);
    __syncthreads();
    if (t256 == 0 && mine) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (bn0 < 2) __hip_atomic_fetch_add(flags + CF_HEAD + bm0, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (bm0 >= 2 && bm0 < 2 + nt_next / 4 && bn0 >= 2) __hip_atomic_fetch_add(flags + 2, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return;
  }
  if (bid <= n_tile_wgs + n_panel_wgs) {
    // two 16-row panel groups side by side (the rows below a diagonal block are a multiple of 64: the count is even)
    const int grp = 2 * (bid - 1 - n_tile_wgs) + half;
    if (tid == 0) {
      bool ok = cf_wait(flags + 0, 1);
      if (tiles > 0) ok = ok && cf_wait(flags + CF_HEAD + 2 + (grp & ~1) / 4, 2);      // (both groups sit in one 64-row tile)
      cf_ok = ok ? 1 : 0;
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    if (!cf_ok && tid == 0 && *iz == 0) *iz = -2000 - kb;
    if (stamps && tid == 0 && grp == 0) stamps[9] = wall_clock64();
    chol128_panel_body(Lz, ld, kb, dz, grp, c128_smem + half * C128_PANEL_LDS_DOUBLES, t256);
    __syncthreads();
    if (stamps && tid == 0 && grp == 0) stamps[13] = wall_clock64();
    if (t256 == 0 && grp < nt_next) cf_publish(flags + 1);
    if (stamps && tid == 0 && grp == 0) stamps[10] = wall_clock64();
    return;
  }
  {
    // two 16x16 tiles of the next diagonal block side by side (36 or 10 of them: even)
    const int t = 2 * (bid - 1 - n_tile_wgs - n_panel_wgs) + half;
    if (tid == 0) {
      bool ok = cf_wait(flags + 1, nt_next);
      if (tiles > 0 && tiles_n > 2) ok = ok && cf_wait(flags + 2, nt_next == 8 ? 3 : 1);
      cf_ok = ok ? 1 : 0;
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    if (!cf_ok && tid == 0 && *iz == 0) *iz = -2000 - kb;
    if (stamps && tid == 0 && t == 0) stamps[11] = wall_clock64();
    chol128_diag_update_body(Lz, ld, kb, t, c128_smem + half * C128_UPD_LDS_DOUBLES, t256);
    if (stamps && tid == 0 && t == 0) stamps[12] = wall_clock64();
  }
}

// ---- launchers -------------------------------------------------------------------------------------------------------
static int launch_step(gpbo_ctx* ctx, Model& m, int kb, int nblk, const GemmArgs* upd, long long* stamps) {
  if (!(ctx->func_attrs & ATTR_CHOL128)) {
    GPBO_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(chol128_step_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)C128_LDS_BYTES));
    ctx->func_attrs |= ATTR_CHOL128;
  }
  GemmArgs g{};
  int tiles = 0, tiles_n = 1, live = 0;
  if (upd) {
    g = *upd;
    g.lanes = ctx->lanes; g.lane_stride = ctx->lane_stride; g.batch = 1;
    tiles_n = g.n / 64;
    const int tiles_m = g.m / 64;
    const char* sq = dbg_env("GPBO_CHOL_SQUARE_TILES");     // debug A/B: 1 = the square enumeration of rounds 3-5
    live = g.lower_only && tiles_m >= tiles_n && !(sq && sq[0] == '1');
    // live tiles: the lower triangle of the leading tiles_n rows + the full rows below, less the leading skip00-block's
    tiles = live ? tiles_n * (tiles_n + 1) / 2 + (tiles_m - tiles_n) * tiles_n - g.skip00 * (g.skip00 + 1) / 2 : tiles_m * tiles_n;
    if (tiles_m <= g.skip00 && tiles_n <= g.skip00) tiles = 0;   // nothing but the block the diagonal workgroup owns
  }
  chol128_step_kernel<<<dim3((unsigned)(1 + (tiles + 1) / 2), (unsigned)ctx->lanes), dim3(512), C128_LDS_BYTES, ctx->stream>>>(
      m.L, m.NP, kb, nblk, m.dinv, ctx->info_dev, g, tiles_n, tiles, live, ctx->lane_stride, stamps);
  GPBO_HIP(ctx, hipGetLastError());
  return GPBO_OK;
}

// ---- look-ahead over the outer panels ----------------------------------------------------------------------------
// The rank-`outer` trailing update of panel p only has to reach the NEXT panel's columns before that panel's chain can
// start; the rest of it (T_b) is independent of the chain and runs on a second ("bulk") stream meanwhile:
//     main:  chain(p) | [wait T_b(p-1)] T_a(p) | chain(p+1) | ...
//     bulk:                             [wait T_a(p)] T_b(p)
// (T_b starts behind T_a, not beside it: side by side the two share the chip and T_a, which the chain waits for, takes
// 106 us instead of ~45 at N = 4096)
// T_a(p) waits for T_b(p-1) because both accumulate into the next panel's columns and the order of the two rank-512
// contributions is part of the result (bitwise the one-stream factor: same launches, same kernels, same order per tile).
// Round 2 measured this with a plain second stream and dropped it: the bulk GEMM's workgroups hold every CU, so the
// chain's small kernels queue behind them.  Here the bulk stream may be confined to a subset of the CUs
// (hipExtStreamCreateWithCUMask; GPBO_CHOL_LA_CUS = how many of the device's CUs it may use, 0 = no mask), which keeps
// the rest free for the chain at any moment.  GPBO_CHOL_LA=0 turns the look-ahead off (A/B runs).
static int lookahead_min_np() {
  static const int v = [] {
    const char* e = dbg_env("GPBO_CHOL_LA");
    if (e && e[0] == '0') return 1 << 30;
    const char* f = dbg_env("GPBO_CHOL_LA_MIN_NP");
    return f ? atoi(f) : 4096;     // measured (scripts/archive/r03_la_probe.py): 2048 0.709 -> 0.721 ms, 4096 1.82 -> 1.75, 8192 6.42 -> 6.17
  }();
  return v;
}

static LookAhead* lookahead_for(gpbo_ctx* ctx, int n_events) {
  LookAhead* la = nullptr;
  for (auto& l : ctx->lookahead)
    if (l.main == ctx->stream) la = &l;
  if (!la) {
    LookAhead l;
    l.main = ctx->stream;
    static const int cus = dbg_env("GPBO_CHOL_LA_CUS") ? atoi(dbg_env("GPBO_CHOL_LA_CUS")) : 0;
    hipError_t e = hipErrorUnknown;
    if (cus > 0) {
      hipDeviceProp_t prop;
      if (hipGetDeviceProperties(&prop, ctx->device) == hipSuccess && cus < prop.multiProcessorCount) {
        std::vector<uint32_t> mask((size_t)(prop.multiProcessorCount + 31) / 32, 0u);
        for (int b = 0; b < cus; ++b) mask[(size_t)b / 32] |= 1u << (b % 32);
        e = hipExtStreamCreateWithCUMask(&l.bulk, (uint32_t)mask.size(), mask.data());
      }
    }
    if (e != hipSuccess) e = hipStreamCreateWithFlags(&l.bulk, hipStreamNonBlocking);
    if (e != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    ctx->lookahead.push_back(l);
    la = &ctx->lookahead.back();
  }
  while ((int)la->ev.size() < n_events) {
    hipEvent_t ev = nullptr;
    if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    la->ev.push_back(ev);
  }
  return la;
}

// One fused launch for step kb (see chol128_fused_step_kernel): `upd` the previous step's in-panel update (null: none), the panel solve
// of the `rem` rows below the block, the update of the next diagonal block of nb2 64-blocks (0: none).  flags: this step's words.
static int launch_fused_step(gpbo_ctx* ctx, Model& m, int kb, int nblk, const GemmArgs* upd, int rem, int nb2, int* flags, long long* stamps) {
  if (!(ctx->func_attrs & ATTR_CHOL_FUSED)) {
    GPBO_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(chol128_fused_step_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)C128_LDS_BYTES));
    ctx->func_attrs |= ATTR_CHOL_FUSED;
  }
  GemmArgs g{};
  int tiles = 0, tiles_n = 1;
  if (upd) {
    g = *upd;
    g.lanes = ctx->lanes; g.lane_stride = ctx->lane_stride; g.batch = 1;
    tiles_n = g.n / 64;
    const int tiles_m = g.m / 64;
    tiles = tiles_n * (tiles_n + 1) / 2 + (tiles_m - tiles_n) * tiles_n - g.skip00 * (g.skip00 + 1) / 2;
    if (tiles_m <= g.skip00 && tiles_n <= g.skip00) tiles = 0;
  }
  const int n_tile_wgs = (tiles + 1) / 2, n_groups = rem / 16, n_panel_wgs = n_groups / 2, nt_next = 4 * nb2;
  const int n_upd_wgs = nt_next * (nt_next + 1) / 4;
  chol128_fused_step_kernel<<<dim3((unsigned)(1 + n_tile_wgs + n_panel_wgs + n_upd_wgs), (unsigned)ctx->lanes), dim3(512), C128_LDS_BYTES,
                              ctx->stream>>>(m.L, m.NP, kb, nblk, m.dinv, ctx->info_dev, g, tiles_n, tiles, n_tile_wgs, n_groups, n_panel_wgs,
                                             nt_next, ctx->lane_stride, flags, stamps);
  GPBO_HIP(ctx, hipGetLastError());
  return GPBO_OK;
}

// Blocked Cholesky of m.L (lower, in place), inverted 64x64 diagonal blocks to m.dinv: 128-column steps inside
// `outer`-column panels (outer a multiple of 128), one rank-`outer` trailing update per panel.
int launch_cholesky128(gpbo_ctx* ctx, Model& m, int outer, long long* stamps) {
  const int nblk = (int)(m.NP / NB);
  const int per_outer = outer / NB;
  const unsigned lanes = (unsigned)ctx->lanes;
  int rc;
  const int n_panels = (nblk + per_outer - 1) / per_outer;
  LookAhead* la = (ctx->lanes == 1 && !ctx->no_lookahead && m.NP >= lookahead_min_np() && n_panels >= 3) ? lookahead_for(ctx, 2 * n_panels) : nullptr;
  bool la_joined = true;
  int pidx = 0;     // outer panel index
  // debug build, read per call: the step as ONE launch with in-launch hand-offs (the round-6 experiment; m.tmp holds the flag words)
  const char* fe = dbg_env("GPBO_CHOL_FUSED_STEP");
  const bool fused_step = fe && fe[0] == '1';
  const int FS = CF_HEAD + (int)(m.NP / 64);
  int* flag_words = reinterpret_cast<int*>(m.tmp);
  if (fused_step) {
    const size_t bytes = (size_t)(nblk / 2 + 1) * FS * sizeof(int);
    for (int l = 0; l < ctx->lanes; ++l)
      GPBO_HIP(ctx, hipMemsetAsync(m.tmp + (int64_t)l * ctx->lane_stride, 0, bytes, ctx->stream));
  }
  for (int ob = 0; ob < nblk; ob += per_outer) {
    const int oe = (ob + per_outer < nblk) ? ob + per_outer : nblk;
    if (fused_step) {
      // step kb = [diagonal block kb | update from panel kb - 2 | panel kb | update of diagonal block kb + 2] in one launch
      for (int kb = ob; kb < oe; kb += 2) {
        const int wb = (oe - kb >= 2) ? 2 : 1;
        const int rem = (int)(m.NP - (int64_t)(kb + wb) * NB);
        if (rem > 0 && wb != 2) GPBO_FAIL(ctx, GPBO_ERR_INVALID, "cholesky128: a single 64-block can only end the matrix");
        GemmArgs s{};
        if (kb > ob) {
          double* panel = m.L + (int64_t)kb * NB * m.NP + (int64_t)(kb - 2) * NB;
          s.m = (int)(m.NP - (int64_t)kb * NB); s.n = (oe - kb) * NB; s.k = 2 * NB; s.alpha = -1.0; s.beta = 1.0;
          s.A = panel; s.lda = m.NP; s.B = panel; s.ldb = m.NP; s.b_trans = 1;
          s.C = m.L + (int64_t)kb * NB * m.NP + (int64_t)kb * NB; s.ldc = m.NP;
          s.lower_only = 1; s.skip00 = wb;
        }
        const int next = kb + 2;
        const int nb2 = (rem > 0 && next < oe) ? ((oe - next >= 2) ? 2 : 1) : 0;
        if ((rc = launch_fused_step(ctx, m, kb, wb, kb > ob ? &s : nullptr, rem, nb2, flag_words + (kb / 2) * FS, kb == 0 ? stamps : nullptr)))
          return rc;
      }
    } else {
    if ((rc = launch_step(ctx, m, ob, (oe - ob >= 2) ? 2 : 1, nullptr, stamps))) return rc;   // everything before the panel is applied
    for (int kb = ob; kb < oe; kb += 2) {
      const int wb = (oe - kb >= 2) ? 2 : 1;
      const int rem = (int)(m.NP - (int64_t)(kb + wb) * NB);      // rows below the diagonal block
      if (rem == 0) break;
      if (wb != 2) GPBO_FAIL(ctx, GPBO_ERR_INVALID, "cholesky128: a single 64-block can only end the matrix");
      chol128_panel_kernel<<<dim3((unsigned)(rem / 16), lanes), dim3(256), 0, ctx->stream>>>(m.L, m.NP, kb, m.dinv, ctx->lane_stride);
      GPBO_HIP(ctx, hipGetLastError());
      const int next = kb + 2;
      if (next < oe) {
        const int nb2 = (oe - next >= 2) ? 2 : 1;
        const int nt = 4 * nb2;
        chol128_diag_update_kernel<<<dim3((unsigned)(nt * (nt + 1) / 2), lanes), dim3(256), 0, ctx->stream>>>(m.L, m.NP, kb,
                                                                                                            ctx->lane_stride);
        GPBO_HIP(ctx, hipGetLastError());
        double* panel = m.L + (int64_t)next * NB * m.NP + (int64_t)kb * NB;
        GemmArgs s{};    // rank-128 update of the panel's remaining columns; the next diagonal block is already done
        s.m = rem; s.n = (oe - next) * NB; s.k = 2 * NB; s.alpha = -1.0; s.beta = 1.0;
        s.A = panel; s.lda = m.NP; s.B = panel; s.ldb = m.NP; s.b_trans = 1;
        s.C = m.L + (int64_t)next * NB * m.NP + (int64_t)next * NB; s.ldc = m.NP;
        s.lower_only = 1; s.skip00 = nb2;
        if ((rc = launch_step(ctx, m, next, nb2, &s, nullptr))) return rc;
      }
    }
    }
    const int rem2 = (int)(m.NP - (int64_t)oe * NB);
    if (rem2 > 0) {
      const double* P = m.L + (int64_t)oe * NB * m.NP + (int64_t)ob * NB;
      GemmArgs t{};
      t.k = (oe - ob) * NB; t.alpha = -1.0; t.beta = 1.0;
      t.lda = m.NP; t.ldb = m.NP; t.b_trans = 1; t.ldc = m.NP;
      t.batch = 1; t.lower_only = 1;
      // one decision for the whole update, whichever way it is launched below (whole, or the next panel's columns + the rest): the
      // sixteen-wave tile kernel sums k in another order, and L must not depend on the look-ahead
      t.m = rem2; t.n = rem2;
      t.fat = gemm_fat_rule(t) ? 1 : -1;
      const int nw = rem2 < outer ? rem2 : outer;       // the next panel's columns
      if (!la || rem2 <= nw) {
        if (la && pidx > 0) GPBO_HIP(ctx, hipStreamWaitEvent(ctx->stream, la->ev[2 * (pidx - 1) + 1], 0));
        t.m = rem2; t.n = rem2; t.A = P; t.B = P;
        t.C = m.L + (int64_t)oe * NB * m.NP + (int64_t)oe * NB;
        if ((rc = launch_gemm(ctx, t))) return rc;
        la_joined = true;
      } else {
        if (pidx > 0) GPBO_HIP(ctx, hipStreamWaitEvent(ctx->stream, la->ev[2 * (pidx - 1) + 1], 0));
        t.m = rem2; t.n = nw; t.A = P; t.B = P;
        t.C = m.L + (int64_t)oe * NB * m.NP + (int64_t)oe * NB;
        if ((rc = launch_gemm(ctx, t))) return rc;        // T_a on main, with the whole chip to itself:
        GPBO_HIP(ctx, hipEventRecord(la->ev[2 * pidx], ctx->stream));   // the bulk stream starts behind it
        hipStream_t main_stream = ctx->stream;
        ctx->stream = la->bulk;
        hipError_t e = hipStreamWaitEvent(la->bulk, la->ev[2 * pidx], 0);
        if (e == hipSuccess) {
          t.m = rem2 - nw; t.n = rem2 - nw; t.A = P + (int64_t)nw * m.NP; t.B = t.A;
          t.C = m.L + ((int64_t)oe * NB + nw) * m.NP + ((int64_t)oe * NB + nw);
          rc = launch_gemm(ctx, t);                          // T_b on bulk
          if (rc == GPBO_OK) e = hipEventRecord(la->ev[2 * pidx + 1], la->bulk);
        }
        ctx->stream = main_stream;
        if (rc) return rc;
        GPBO_HIP(ctx, e);
        la_joined = false;
      }
    }
    ++pidx;
  }
  // nothing of the factorisation may still be in flight on the bulk stream when main goes on (it always joined above:
  // the last update has no far part)
  if (la && !la_joined && pidx > 0) GPBO_HIP(ctx, hipStreamWaitEvent(ctx->stream, la->ev[2 * (pidx - 1) + 1], 0));
  return GPBO_OK;
}

}  // namespace gpbo
