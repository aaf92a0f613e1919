// posterior_kernel_v2 — the posterior MFMA kernel tiled for FOUR waves per SIMD (v1, 2 waves/SIMD, is in the git
// history), with the k* operand either generated in-kernel (GEN = 1) or read from a slab (GEN = 2, "v3").
//
// Why: the in-tree probe (gpbo_mfma_f64_probe) shows that on gfx950 a SIMD only reaches the
// 64-cycle issue cadence of v_mfma_f64_16x16x4_f64 when >= 4 waves feed it (1 wave: 140 cycles per
// MFMA, 2 waves: 102, 4 waves: 63), and rocprofv3 puts v1 (256 VGPRs -> 2 waves/SIMD) at 56 % matrix-pipe
// busy.  v2 halves the per-wave accumulator tile so that a wave fits 128 VGPRs:
//
//   workgroup = 8 waves, 256 rows of W x 64 candidates, two workgroups resident per CU (4 waves/SIMD);
//   wave w    = 32 rows x 64 candidates = 2 x 4 MFMA tiles = 64 accumulator VGPRs;
//   k* stage  = 16 train points x 64 candidates, 2 elements per thread (lane = candidate, wave = k pair),
//               candidate coordinates read from an LDS image (no registers to spare for them);
//   W slab    = streamed HBM/L2 -> registers in fragment order, prefetched one k-pair (8 columns) ahead.
//
// Same math, same reduction order per row chunk as v1 up to the candidate-tile width; results are
// deterministic.  Replaces the same reference lines as posterior_kernel.hip (_gpr.py:443-494).
#include <cstdlib>
#include <type_traits>

#include "gpbo_internal.h"
#include "fit_bodies.h"

namespace gpbo {

typedef double d4 __attribute__((ext_vector_type(4)));

constexpr int V2_CANDS = 64;
// Stage tile in LDS: [k-quad q][candidate group jt (16)][k in quad (4)][candidate in group (16)] doubles, i.e. blocks of
// 64 doubles = 512 B holding exactly what ONE B-fragment read of v_mfma_f64_16x16x4_f64 takes (lane l -> element l of
// block (q, jt)).  Every read of a stage — and of both stage buffers — is then the lane's own constant address plus a
// multiple of 512 B, which ds_read2st64_b64 carries as an immediate: no address arithmetic per read, and a wave reads
// 512 contiguous bytes (no bank conflicts).  (Round 1: [k][80] rows: 12 VALU instructions per stage for addresses.)
constexpr int V2_STRIDE = 64;  // doubles per k-row equivalent: a stage of BK points takes BK * 64 doubles

struct PostArgs2 {
  const double* Wp;
  const double* Xs;
  const double* alpha;
  const double* Xcs;
  double* part;
  double* mu_part;
  int NP;
  int64_t Mp;
  int nchunks;
  int n_ctiles;
  const double* Kst;   // GEN == 2: materialised k* slab [NP][ldk], candidate-contiguous
  int64_t ldk;
  int64_t m0;          // first candidate of the slab (outputs are indexed m0 + local)
  int fuse_ends = 0;   // GEN == 1, one row chunk: raw candidates in, mu / sd out (PostEnds, gpbo_internal.h)
  PostEnds ends = {};
};

constexpr int BUF_FLAGS = 0x00020000;   // gfx9 buffer descriptor word 3: raw buffer, 32-bit data format

// GEN = 1: k* generated in the kernel (fused).  GEN = 2: k* read from a slab materialised by
// kstar_gen_kernel (the fp64 VALU work of the generation shares the FP64 datapath with the MFMAs — measured:
// 31 % of the fused kernel's time at C3 — so paying it once per candidate instead of once per row chunk wins).
// GEN = 0 is a timing-only ablation (k* replaced by a constant; results are wrong): GPBO_POST_ABLATE_GEN=1.
// BK = train points per LDS stage (one s_barrier per stage): 16, or 32 for the slab kernel (half the barriers; the
// triangular cut-off of a 16-row tile then rounds up to 32 columns — zeros of the packed W, a few per cent more MFMAs
// in the diagonal chunk only).
// WAVES = 8: 256-row chunks, two workgroups per CU.  WAVES = 16 (round 4, GEN = 1 only, "v4"): 512-row chunks, ONE 1024-thread
// workgroup per CU — for NP <= 1024 a candidate tile's k* is then generated once (NP <= 512) or 1.5 times (NP <= 1024)
// instead of once per 256-row chunk (1.5 / 2.5 times) and never crosses HBM: the slab route generates it once too, but pays a
// 268 MB round trip at N = 512, M = 65 536 and a second launch (C2: 0.09 + 0.29 ms).
template <int DP, int KERNEL, int GEN, int BK = POST_BK, int WAVES = 8>
__global__ __launch_bounds__(WAVES * 64, 4) void posterior_kernel_v2(PostArgs2 p) {
  static_assert(BK == 16 || BK == 32 || BK == 64, "stages of 16, 32 or 64 train points");
  static_assert(WAVES == 8 || (WAVES == 16 && GEN == 1 && (BK == 32 || BK == 64)), "the 16-wave form is the fused kernel with 32- or 64-point stages");
  constexpr int NT = WAVES * 64;                   // threads
  constexpr int ROWS = WAVES * 32;                 // rows of W per workgroup (wave = two 16-row tiles)
  constexpr int E = BK / WAVES;                    // stage elements per thread (lane = candidate, wave = E train points)
  static_assert(GEN == 2 || E == 2 || E == 4, "the in-kernel generation computes train points in pairs: one or two pairs per thread and stage");
  constexpr int KP = BK / 8;                       // k-pairs (8 columns of W) per stage
  extern __shared__ __attribute__((aligned(16))) double smem2[];
  double* Ks = smem2;                              // [2][BK][V2_STRIDE]
  double* Xl = smem2 + 2 * BK * V2_STRIDE;         // [DP][64] candidate coordinates, dimension-major

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  // Heaviest row chunks first; the workgroups resident at any time share a chunk, so its rows of W come out of L2.
  // (Round-2 A/B: mappings that put the two chunks of a candidate tile 1 ... 64 block ids apart, hoping the second
  // chunk's k* reads would hit the first one's in L2, never lowered FETCH_SIZE — 3.5e11 ... 6.6e11 B against 3.3e11 —
  // and cost up to 6 % of the time; removed.)
  const int bid = blockIdx.x;
  const int r = p.nchunks - 1 - bid / p.n_ctiles;
  const int ct = bid - (bid / p.n_ctiles) * p.n_ctiles;
  const bool last = (r == p.nchunks - 1);
  const int NP = p.NP;
  const int k_end = min(NP, (r + 1) * ROWS);
  const int n_stages = k_end / BK;

  // candidate tile -> LDS (thread t loads candidate t>>3, dims (t&7)*DP/8 ...)
  if constexpr (GEN != 2) {
    if (p.fuse_ends) {
      // the raw tile, scaled on the way in: prescale_elem's element (row < M and t < d: X / length_scale, else 0), into LDS instead of memory
      for (int e = tid; e < V2_CANDS * DP; e += NT) {
        const int cnd = e / DP, t = e - cnd * DP;
        const int64_t row = (int64_t)ct * V2_CANDS + cnd;
        double v = 0.0;
        if (row < p.ends.M && t < p.ends.d) v = p.ends.Xc[row * p.ends.d + t] / p.ends.ls[t];
        Xl[t * V2_CANDS + cnd] = v;
      }
    } else {
      const double* src = p.Xcs + (int64_t)ct * V2_CANDS * DP;
      for (int e = tid; e < V2_CANDS * DP; e += NT) {
        const int cnd = e / DP, t = e - cnd * DP;
        Xl[t * V2_CANDS + cnd] = src[e];
      }
    }
  }

  // MFMA role.  A chunk holds 16 tiles of 16 rows; wave w owns tiles w and 15 - w (not two adjacent ones):
  // in the chunk's diagonal block a tile t only needs the stages up to its own rows, so the pairing gives
  // every wave the same (t+1) + (16-t) = 17 tile-stages of work instead of 3 ... 31.
  const int tileA = r * (ROWS / 16) + wave;                     // global 16-row tile index (the earlier one)
  const int tileB = r * (ROWS / 16) + (2 * WAVES - 1) - wave;   // the later one
  const int rowA0 = tileA * 16, rowB0 = tileB * 16;
  const bool activeA = rowA0 < NP, activeB = rowB0 < NP;    // false only in a ragged last chunk
  const int64_t pairs = NP / 8;
  // packed W: [slab of 32 rows][k-pair][tile of 16 rows][lane] double2 ; inactive tiles stream tile 0 (dropped later)
  const int tA = activeA ? tileA : 0, tB = activeB ? tileB : 0;
  // Both operand streams are read through buffer descriptors: wave-uniform base in SGPRs, the lane as a constant 32-bit
  // offset, the walk along k as the instruction's scalar offset — no per-lane 64-bit address arithmetic in the MFMA loop
  // (12 v_lshl_add_u64 per stage with flat global loads; fp64 MFMAs and other VALU work do not overlap on this part).
  const double2* wpA = reinterpret_cast<const double2*>(p.Wp) + ((int64_t)(tA >> 1) * pairs * 2 + (tA & 1)) * 64;
  const double2* wpB = reinterpret_cast<const double2*>(p.Wp) + ((int64_t)(tB >> 1) * pairs * 2 + (tB & 1)) * 64;
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<double2*>(wpA), 0, 0x7fffffff, BUF_FLAGS);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<double2*>(wpB), 0, 0x7fffffff, BUF_FLAGS);
  const unsigned voff16 = (unsigned)lane * 16u, voff8 = (unsigned)lane * 8u;

  d4 acc[2][4];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[t][j] = d4{0.0, 0.0, 0.0, 0.0};
  double mu_acc = 0.0;

  // generation role: candidate = lane, train points E*wave .. E*wave + E - 1 of the stage
  auto gen_compute = [&](int stage, double (&kv)[E], double mu_weight) {
    const int j0 = stage * BK + wave * E;
    if constexpr (GEN == 0) {
      kv[0] = 1e-3 * lane;
      kv[1] = 2e-3 * lane + stage;
      return;
    }
    if constexpr (GEN == 2) {
      const double* src = p.Kst + (int64_t)j0 * p.ldk + (int64_t)ct * V2_CANDS;       // wave-uniform
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(src), 0, 0x7fffffff, BUF_FLAGS);
      const unsigned row = (unsigned)p.ldk * 8u;                                       // ldk * 8 * (E - 1) < 2^31 (slab budget)
#pragma unroll
      for (int e = 0; e < E; ++e)
        kv[e] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rs, voff8, (unsigned)e * row, 0));
      return;
    }
#pragma unroll
    for (int e2 = 0; e2 < E; e2 += 2) {
      const double* xr = p.Xs + (int64_t)(j0 + e2) * DP;  // wave-uniform -> scalar loads
      double d2a = 0.0, d2b = 0.0;
      if constexpr (DP <= 8) {
#pragma unroll
        for (int t = 0; t < DP; ++t) {
          const double x = Xl[t * V2_CANDS + lane];
          const double da = x - xr[t], db = x - xr[DP + t];
          d2a = fma(da, da, d2a);
          d2b = fma(db, db, d2b);
        }
      } else {
#pragma unroll 1
        for (int tb = 0; tb < DP; tb += 8) {
#pragma unroll
          for (int tt = 0; tt < 8; ++tt) {
            const double x = Xl[(tb + tt) * V2_CANDS + lane];
            const double da = x - xr[tb + tt], db = x - xr[DP + tb + tt];
            d2a = fma(da, da, d2a);
            d2b = fma(db, db, d2b);
          }
        }
      }
      kv[e2] = gpbo_kernel_value<KERNEL>(d2a);
      // (16 waves = 128 VGPRs: with DP = 4 the two inlined evaluations, interleaved, spilled 15-29 registers; one after the other fits)
      if constexpr (WAVES == 16 && (DP <= 4 || E > 2)) __builtin_amdgcn_sched_barrier(0);
      kv[e2 + 1] = gpbo_kernel_value<KERNEL>(d2b);
      if constexpr (WAVES == 16 && (DP <= 4 || E > 2)) __builtin_amdgcn_sched_barrier(0);
      // mu_weight = 0 for the clamped (repeated) look-ahead of the last stage: it must not be counted twice
      mu_acc = fma(kv[e2] * mu_weight, p.alpha[j0 + e2], mu_acc);
      mu_acc = fma(kv[e2 + 1] * mu_weight, p.alpha[j0 + e2 + 1], mu_acc);
    }
  };
  auto gen_store = [&](const double (&kv)[E], int buf) {
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const int k = wave * E + e;     // train point of the stage this thread generated / fetched for candidate `lane`
      Ks[((buf * (BK / 4) + (k >> 2)) * 4 + (lane >> 4)) * 64 + (k & 3) * 16 + (lane & 15)] = kv[e];
    }
  };

  // A fragments for one k-pair (8 columns): [tile] double2 = 8 VGPRs
  auto loadA = [&](int kpair, double2(&a)[2]) {
    a[0] = __builtin_bit_cast(double2, __builtin_amdgcn_raw_buffer_load_b128(rsA, voff16, (unsigned)kpair * 2048u, 0));
    a[1] = __builtin_bit_cast(double2, __builtin_amdgcn_raw_buffer_load_b128(rsB, voff16, (unsigned)kpair * 2048u, 0));
  };
  // MODE 2: both tiles, 1: only the later tile (B), compile-time so that the hot loop stays one basic block
  auto mma_pair = [&](int buf, int pp, const double2(&a)[2], auto mode) {
    constexpr int MODE = decltype(mode)::value;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int q = pp * 2 + e;
      const double a0 = e ? a[0].y : a[0].x;
      const double a1 = e ? a[1].y : a[1].x;
      const double* kb = Ks + (buf * (BK / 4) + q) * 256 + lane;
#pragma unroll
      for (int jt = 0; jt < 4; ++jt) {
        const double b = kb[jt * 64];
        if constexpr (MODE == 2) acc[0][jt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b, acc[0][jt], 0, 0, 0);
        acc[1][jt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b, acc[1][jt], 0, 0, 0);
      }
    }
  };
  using both_t = std::integral_constant<int, 2>;
  using later_t = std::integral_constant<int, 1>;

  __syncthreads();   // Xl visible
  {
    double kv0[E];
    gen_compute(0, kv0, 1.0);
    gen_store(kv0, 0);
  }
  double2 aA[2], aB[2];
  loadA(0, aA);
  __syncthreads();

  // One stage = BK train points.  MODE 2: both tiles multiply, 1: only the later tile, 0: none (the wave only
  // feeds the stage tile).  The body is branch-free (the look-ahead indices are clamped instead of tested) so that
  // loads, MFMAs and the LDS store of a stage stay in one basic block.
  const int last_stage = n_stages - 1;
  const int last_pair = (int)pairs - 1;
  auto stage = [&](int st, auto mode) {
    constexpr int MODE = decltype(mode)::value;
    const int buf = st & 1;
    double kv[E];
    if constexpr (MODE > 0) loadA(min(KP * st + 1, last_pair), aB);
    gen_compute(min(st + 1, last_stage), kv, st < last_stage ? 1.0 : 0.0);
    // slab mode: these are plain global loads — keep them at the top of the stage (a full stage of MFMAs hides
    // their latency); without the fence hipcc sinks them next to their first use
    if constexpr (GEN == 2) __builtin_amdgcn_sched_barrier(0);
    // k-pairs alternate between the two fragment registers; each is refilled (two pairs ahead) right after its use —
    // the last one of the stage at the top of the next stage
    if constexpr (MODE > 0) {
#pragma unroll
      for (int pp = 0; pp < KP; ++pp) {
        if (pp & 1) mma_pair(buf, pp, aB, mode);
        else mma_pair(buf, pp, aA, mode);
        if (pp + 1 < KP || (pp & 1) == 0) {
          if (pp & 1) loadA(min(KP * st + pp + 2, last_pair), aB);
          else loadA(min(KP * st + pp + 2, last_pair), aA);
        }
      }
    }
    gen_store(kv, buf ^ 1);   // after the last LDS read of this stage (for st == last_stage nobody reads it)
    __syncthreads();
  };
  // stages 0 .. sA: both tiles; sA+1 .. sB: the later tile only; beyond: none (W is lower triangular)
  const int sA = min(last_stage, (rowA0 + 15) / BK);
  const int sB = min(last_stage, (rowB0 + 15) / BK);
  int s = 0;
  for (; s <= sA; ++s) stage(s, both_t{});
  for (; s <= sB; ++s) stage(s, later_t{});
  for (; s <= last_stage; ++s) stage(s, std::integral_constant<int, 0>{});

  // epilogue: per-candidate sum of squares over this chunk's rows, fixed order
  static_assert(2 * WAVES * V2_CANDS <= 2 * BK * V2_STRIDE, "the epilogue's exchange area aliases the stage tiles");
  double* red = Ks;                           // [WAVES][64]
  double* mured = Ks + WAVES * V2_CANDS;      // [WAVES][64]
#pragma unroll
  for (int jt = 0; jt < 4; ++jt) {
    double v = 0.0;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) v = fma(acc[0][jt][rr], acc[0][jt][rr], v);
    if (!activeA) v = 0.0;
    double vb = 0.0;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) vb = fma(acc[1][jt][rr], acc[1][jt][rr], vb);
    if (activeB) v += vb;
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    if (lane < 16) red[wave * V2_CANDS + jt * 16 + lane] = v;
  }
  mured[wave * V2_CANDS + lane] = mu_acc;
  __syncthreads();
  if (tid < V2_CANDS) {
    double v = 0.0;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) v += red[w * V2_CANDS + tid];
    const int64_t m = p.m0 + (int64_t)ct * V2_CANDS + tid;
    if (GEN != 2 && p.fuse_ends) {
      // this workgroup holds every row of its candidates: mu and sd from here (what posterior_finalize_kernel makes of ONE partial
      // each: 0.0 + v and 0.0 + u are v and u)
      double u = 0.0;
#pragma unroll
      for (int w = 0; w < WAVES; ++w) u += mured[w * V2_CANDS + tid];
      if (m < p.ends.M)
        posterior_finalize_elem(0.0 + v, 0.0 + u, p.ends.y_mean, p.ends.y_std, p.ends.mu + m, p.ends.sd + m, p.ends.negvar);
      return;
    }
    p.part[(int64_t)r * p.Mp + m] = v;
    if (GEN != 2 && last) {
      double u = 0.0;
#pragma unroll
      for (int w = 0; w < WAVES; ++w) u += mured[w * V2_CANDS + tid];
      p.mu_part[m] = u;
    }
  }
}

// k* slab generation: thread = candidate (coordinates in registers), loop over a chunk of 256 train points, coalesced
// stores to Kst[k][m]; also the partial means sum_k k*[k] alpha[k] per chunk.  The train points of the chunk are staged
// through LDS 64 at a time and read back as broadcasts (every lane the same address): as scalar loads straight from
// memory each pair of points cost the wave four s_load_dwordx16 + s_waitcnt round trips per iteration, which left the
// fp64 VALU — the unit this kernel is bound by — idle about 40 % of the time (14.2 ms per C3 pass).
constexpr int GEN_CH = 64;
template <int DP, int KERNEL>
__global__ __launch_bounds__(256) void kstar_gen_kernel(const double* __restrict__ Xs, const double* __restrict__ alpha,
                                                        const double* __restrict__ Xcs, double* __restrict__ Kst,
                                                        int64_t ldk, int NP, double* __restrict__ mu_part,
                                                        int64_t Mp, int64_t m0) {
  __shared__ __attribute__((aligned(16))) double xs[GEN_CH * DP];
  __shared__ double al[GEN_CH];
  const int64_t ml = (int64_t)blockIdx.x * 256 + threadIdx.x;   // slab-local candidate
  const bool live = ml < ldk;
  const int k0 = blockIdx.y * POST_ROWS, k1 = min(NP, k0 + POST_ROWS);
  double xc[DP];
  {
    const double* xcp = Xcs + (m0 + (live ? ml : 0)) * DP;
#pragma unroll
    for (int t = 0; t < DP; t += 2) {
      const double2 v = *reinterpret_cast<const double2*>(xcp + t);
      xc[t] = v.x;
      xc[t + 1] = v.y;
    }
  }
  double mu = 0.0;
  for (int kc = k0; kc < k1; kc += GEN_CH) {      // NP is a multiple of 64: every refill is full
    __syncthreads();
    {
      const double2* src = reinterpret_cast<const double2*>(Xs + (int64_t)kc * DP);
      double2* dst = reinterpret_cast<double2*>(xs);
      for (int e = threadIdx.x; e < GEN_CH * DP / 2; e += 256) dst[e] = src[e];
      if (threadIdx.x < GEN_CH) al[threadIdx.x] = alpha[kc + threadIdx.x];
    }
    __syncthreads();
    if (live) {
#pragma unroll 2
      for (int kk = 0; kk < GEN_CH; kk += 2) {
        const double* xr = xs + kk * DP;           // the same address in every lane: LDS broadcast
        double d2a = 0.0, d2b = 0.0;
#pragma unroll
        for (int t = 0; t < DP; ++t) {
          const double da = xc[t] - xr[t], db = xc[t] - xr[DP + t];
          d2a = fma(da, da, d2a);
          d2b = fma(db, db, d2b);
        }
        const double ka = gpbo_kernel_value<KERNEL>(d2a), kb = gpbo_kernel_value<KERNEL>(d2b);
        const int k = kc + kk;
        Kst[(int64_t)k * ldk + ml] = ka;
        Kst[(int64_t)(k + 1) * ldk + ml] = kb;
        mu = fma(ka, al[kk], mu);
        mu = fma(kb, al[kk + 1], mu);
      }
    }
  }
  if (live) mu_part[(int64_t)blockIdx.y * Mp + m0 + ml] = mu;
}

template <int DP, int KERNEL>
static int launch_gen_t(gpbo_ctx* ctx, Model& m, double* Kst, int64_t ldk, int64_t Mp, int64_t m0, int nchunks) {
  dim3 grid((unsigned)((ldk + 255) / 256), (unsigned)nchunks);
  kstar_gen_kernel<DP, KERNEL><<<grid, dim3(256), 0, ctx->stream>>>(m.Xs, m.alpha, ctx->Xcs, Kst, ldk, (int)m.NP,
                                                                      ctx->mu_part, Mp, m0);
  GPBO_HIP(ctx, hipGetLastError());
  return GPBO_OK;
}

template <int KERNEL>
static int launch_gen_k(gpbo_ctx* ctx, Model& m, double* Kst, int64_t ldk, int64_t Mp, int64_t m0, int nchunks) {
  switch (m.DP) {
    case 4: return launch_gen_t<4, KERNEL>(ctx, m, Kst, ldk, Mp, m0, nchunks);
    case 8: return launch_gen_t<8, KERNEL>(ctx, m, Kst, ldk, Mp, m0, nchunks);
    case 16: return launch_gen_t<16, KERNEL>(ctx, m, Kst, ldk, Mp, m0, nchunks);
    case 32: return launch_gen_t<32, KERNEL>(ctx, m, Kst, ldk, Mp, m0, nchunks);
    case 64: return launch_gen_t<64, KERNEL>(ctx, m, Kst, ldk, Mp, m0, nchunks);
  }
  GPBO_FAIL(ctx, GPBO_ERR_UNSUPPORTED, "posterior: unsupported padded dimension");
}

// k* slab [NP][ldk] (+ partial means) for candidates [m0, m0 + ldk) of the scaled set ctx->Xcs — also used by the
// covariance path (posterior_cov.hip)
int launch_kstar_slab(gpbo_ctx* ctx, Model& m, double* Kst, int64_t ldk, int64_t Mp, int64_t m0, int nchunks) {
  if (m.kernel == GPBO_KERNEL_MATERN25) return launch_gen_k<GPBO_KERNEL_MATERN25>(ctx, m, Kst, ldk, Mp, m0, nchunks);
  return launch_gen_k<GPBO_KERNEL_RBF>(ctx, m, Kst, ldk, Mp, m0, nchunks);
}

// Two-kernel pipeline per candidate slab: kstar_gen_kernel -> posterior_kernel_v2<.., GEN = 2>.
// The slab width is bounded by a workspace budget (default 4 GB, GPBO_KSTAR_GB to override); mu partials
// need nchunks x Mp doubles in ctx->mu_part (allocated by the caller).
int launch_posterior_v3(gpbo_ctx* ctx, Model& m, int64_t Mp, int nchunks) {
  // k* workspace: the candidate set is walked slab by slab; a slab only has to be wide enough to fill the chip
  // (4e9 B = 121 984 candidates at N = 4096 = 1906 candidate tiles x 16 row chunks per launch); measured at C3: one 34 GB
  // slab 263.7 ms, eight 4 GB slabs 264.4 ms (round 1 A/B) — the big workspace bought nothing.  GPBO_KSTAR_GB overrides.
  // 32 train points per stage: 262.9 vs 264.0 ms per C3 launch (round-2 A/B, same box, same run); GPBO_POST_BK=16 restores 16
  static const int post_bk = (dbg_env("GPBO_POST_BK") && atoi(dbg_env("GPBO_POST_BK")) == 16) ? 16 : 32;
  const int64_t budget = kstar_slab_budget_bytes(ctx, Mp * m.NP * 8);
  int64_t ms = budget / (m.NP * 8);
  // the slab kernel addresses a stage's rows as 32-bit buffer offsets: 3 rows of ldk doubles must stay below 2^31 bytes
  if (ms > (int64_t)80 * 1000 * 1000) ms = (int64_t)80 * 1000 * 1000;
  ms = ms / 128 * 128;
  if (ms < 128) GPBO_FAIL(ctx, GPBO_ERR_HIP, "posterior: not enough device memory for one k* slab");
  if (ms > Mp) ms = Mp;
  int rc;
  if ((rc = ensure(ctx, &ctx->kst, &ctx->cap_kst, ms * m.NP))) return rc;
  for (int64_t m0 = 0; m0 < Mp; m0 += ms) {
    const int64_t ldk = (Mp - m0 < ms) ? (Mp - m0) : ms;
    if (m.kernel == GPBO_KERNEL_MATERN25) rc = launch_gen_k<GPBO_KERNEL_MATERN25>(ctx, m, ctx->kst, ldk, Mp, m0, nchunks);
    else rc = launch_gen_k<GPBO_KERNEL_RBF>(ctx, m, ctx->kst, ldk, Mp, m0, nchunks);
    if (rc) return rc;
    PostArgs2 a;
    a.Wp = m.Wp; a.Xs = m.Xs; a.alpha = m.alpha; a.Xcs = ctx->Xcs; a.part = ctx->part;
    a.mu_part = ctx->mu_part; a.NP = (int)m.NP; a.Mp = Mp; a.nchunks = nchunks;
    a.n_ctiles = (int)(ldk / V2_CANDS); a.Kst = ctx->kst; a.ldk = ldk; a.m0 = m0;
    const int64_t nblocks = (int64_t)a.n_ctiles * nchunks;
    if (nblocks > 0x7fffffffLL) GPBO_FAIL(ctx, GPBO_ERR_UNSUPPORTED, "posterior: grid too large; shard the candidates");
    if (post_bk == 32) {
      const size_t lds = (size_t)(2 * 32 * V2_STRIDE) * sizeof(double);
      posterior_kernel_v2<4, 0, 2, 32><<<dim3((unsigned)nblocks), dim3(512), lds, ctx->stream>>>(a);
    } else {
      const size_t lds = (size_t)(2 * POST_BK * V2_STRIDE) * sizeof(double);
      posterior_kernel_v2<4, 0, 2><<<dim3((unsigned)nblocks), dim3(512), lds, ctx->stream>>>(a);
    }
    GPBO_HIP(ctx, hipGetLastError());
  }
  return GPBO_OK;
}

template <int DP, int KERNEL>
static int launch_v2_t(gpbo_ctx* ctx, const PostArgs2& a, int64_t nblocks) {
  const size_t lds = (size_t)(2 * POST_BK * V2_STRIDE + DP * V2_CANDS) * sizeof(double);
#ifdef GPBO_DEBUG   // timing-only ablation (k* replaced by a constant: WRONG results) — scripts/ablate_gen.py, debug build only
  const char* ab = dbg_env("GPBO_POST_ABLATE_GEN");
  if (ab && ab[0] == '1') {
    posterior_kernel_v2<DP, KERNEL, 0><<<dim3((unsigned)nblocks), dim3(512), lds, ctx->stream>>>(a);
    GPBO_HIP(ctx, hipGetLastError());
    return GPBO_OK;
  }
#endif
  posterior_kernel_v2<DP, KERNEL, 1><<<dim3((unsigned)nblocks), dim3(512), lds, ctx->stream>>>(a);
  GPBO_HIP(ctx, hipGetLastError());
  return GPBO_OK;
}

template <int KERNEL>
static int launch_v2_k(gpbo_ctx* ctx, int DP, const PostArgs2& a, int64_t nblocks) {
  switch (DP) {
    case 4: return launch_v2_t<4, KERNEL>(ctx, a, nblocks);
    case 8: return launch_v2_t<8, KERNEL>(ctx, a, nblocks);
    case 16: return launch_v2_t<16, KERNEL>(ctx, a, nblocks);
    case 32: return launch_v2_t<32, KERNEL>(ctx, a, nblocks);
    case 64: return launch_v2_t<64, KERNEL>(ctx, a, nblocks);
  }
  GPBO_FAIL(ctx, GPBO_ERR_UNSUPPORTED, "posterior: unsupported padded dimension");
}

// "v4": the fused kernel in its 16-wave form, 512-row chunks (NP <= 1024: at most two).  *part_chunks = the number of row
// chunks the sum-of-squares partials are split into (what posterior_finalize_kernel sums over).
template <int DP, int KERNEL>
static int launch_v4_t(gpbo_ctx* ctx, const PostArgs2& a, int64_t nblocks) {
#ifdef GPBO_DEBUG   // experiment (round 5): 64-point stages = half the workgroup barriers of the 1024-thread workgroup; GPBO_POST_V4_BK=64
  if (dbg_env("GPBO_POST_V4_BK") && atoi(dbg_env("GPBO_POST_V4_BK")) == 64) {
    const size_t lds64 = (size_t)(2 * 64 * V2_STRIDE + DP * V2_CANDS) * sizeof(double);
    static bool attr[6] = {false, false, false, false, false, false};
    const int ai = KERNEL * 3 + (DP <= 8 ? 0 : DP <= 16 ? 1 : 2);
    if (!attr[ai]) {
      GPBO_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(posterior_kernel_v2<DP, KERNEL, 1, 64, 16>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * 64 * V2_STRIDE + 64 * V2_CANDS) * 8));
      attr[ai] = true;
    }
    posterior_kernel_v2<DP, KERNEL, 1, 64, 16><<<dim3((unsigned)nblocks), dim3(1024), lds64, ctx->stream>>>(a);
    GPBO_HIP(ctx, hipGetLastError());
    return GPBO_OK;
  }
#endif
  const size_t lds = (size_t)(2 * 32 * V2_STRIDE + DP * V2_CANDS) * sizeof(double);
  posterior_kernel_v2<DP, KERNEL, 1, 32, 16><<<dim3((unsigned)nblocks), dim3(1024), lds, ctx->stream>>>(a);
  GPBO_HIP(ctx, hipGetLastError());
  return GPBO_OK;
}
template <int KERNEL>
static int launch_v4_k(gpbo_ctx* ctx, int DP, const PostArgs2& a, int64_t nblocks) {
  switch (DP) {
    case 4: return launch_v4_t<4, KERNEL>(ctx, a, nblocks);
    case 8: return launch_v4_t<8, KERNEL>(ctx, a, nblocks);
    case 16: return launch_v4_t<16, KERNEL>(ctx, a, nblocks);
    case 32: return launch_v4_t<32, KERNEL>(ctx, a, nblocks);
    case 64: return launch_v4_t<64, KERNEL>(ctx, a, nblocks);
  }
  GPBO_FAIL(ctx, GPBO_ERR_UNSUPPORTED, "posterior: unsupported padded dimension");
}
int launch_posterior_v4(gpbo_ctx* ctx, Model& m, int64_t Mp, int* part_chunks, const PostEnds* ends) {
  const int nch = (int)((m.NP + 511) / 512);
  if (ends && nch != 1) GPBO_FAIL(ctx, GPBO_ERR_STATE, "posterior: fused ends need one row chunk");
  PostArgs2 a;
  if (ends) { a.fuse_ends = 1; a.ends = *ends; }
  a.Wp = m.Wp; a.Xs = m.Xs; a.alpha = m.alpha; a.Xcs = ctx->Xcs; a.part = ctx->part;
  a.mu_part = ctx->mu_part; a.NP = (int)m.NP; a.Mp = Mp; a.nchunks = nch;
  a.n_ctiles = (int)(Mp / V2_CANDS);
  a.Kst = nullptr; a.ldk = 0; a.m0 = 0;
  const int64_t nblocks = (int64_t)a.n_ctiles * nch;
  if (nblocks > 0x7fffffffLL) GPBO_FAIL(ctx, GPBO_ERR_UNSUPPORTED, "posterior: grid too large; shard the candidates");
  *part_chunks = nch;
  if (m.kernel == GPBO_KERNEL_MATERN25) return launch_v4_k<GPBO_KERNEL_MATERN25>(ctx, m.DP, a, nblocks);
  return launch_v4_k<GPBO_KERNEL_RBF>(ctx, m.DP, a, nblocks);
}

// Mp must be a multiple of 128 (the v1 tile) — also a multiple of 64.
int launch_posterior_v2(gpbo_ctx* ctx, Model& m, int64_t Mp, int nchunks, const PostEnds* ends) {
  if (ends && nchunks != 1) GPBO_FAIL(ctx, GPBO_ERR_STATE, "posterior: fused ends need one row chunk");
  PostArgs2 a;
  if (ends) { a.fuse_ends = 1; a.ends = *ends; }
  a.Wp = m.Wp; a.Xs = m.Xs; a.alpha = m.alpha; a.Xcs = ctx->Xcs; a.part = ctx->part;
  a.mu_part = ctx->mu_part; a.NP = (int)m.NP; a.Mp = Mp; a.nchunks = nchunks;
  a.n_ctiles = (int)(Mp / V2_CANDS);
  a.Kst = nullptr; a.ldk = 0; a.m0 = 0;
  const int64_t nblocks = (int64_t)a.n_ctiles * nchunks;
  if (nblocks > 0x7fffffffLL) GPBO_FAIL(ctx, GPBO_ERR_UNSUPPORTED, "posterior: grid too large; shard the candidates");
  if (m.kernel == GPBO_KERNEL_MATERN25) return launch_v2_k<GPBO_KERNEL_MATERN25>(ctx, m.DP, a, nblocks);
  return launch_v2_k<GPBO_KERNEL_RBF>(ctx, m.DP, a, nblocks);
}

}  // namespace gpbo
