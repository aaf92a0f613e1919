// fp32 posterior pipeline (precision = GPBO_F32; BASELINE.json configs[4] is quoted in fp32).
//
// The factorisation (K, L, W = L^-1, alpha) stays in fp64 — it is O(N^3) once per fit and its accuracy
// decides everything downstream.  What runs in fp32 is the M-scaled part:
//   kstar_gen_f32_kernel : k* evaluated in fp64 (distance, sqrt, exp) and ROUNDED to fp32 into the slab
//                          [NP][slab] (half the HBM traffic of the fp64 slab); the means k*.alpha are
//                          accumulated in fp64 before rounding, so mu keeps fp64 accuracy;
//   posterior_kernel_f32 : V = W K*^T on v_mfma_f32_16x16x4_f32 (exact f32, 2x the fp64 matrix rate) with W
//                          rounded to fp32 in fragment order; sum of squares in fp32 inside a 256-row chunk,
//                          fp64 across lanes / waves / chunks.
// Same decomposition as posterior_kernel_v2<GEN=2> (8 waves, wave = 32 rows x 64 candidates, 4 waves/SIMD),
// with 32 train points per stage.  The reference has no fp32 path (everything is float64,
// bayes_opt/target_space.py:95-96): this mode trades ~1e-3 relative accuracy on sigma for throughput and is
// checked against the fp64 goldens with that tolerance (tests/test_gpu_f32.py).
#include <cstdlib>
#include <type_traits>

#include "gpbo_internal.h"

namespace gpbo {

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

// Which f32 MFMA the big chunks (NP >= 512) run on: v_mfma_f32_32x32x2_f32 (default) or v_mfma_f32_16x16x4_f32
// (GPBO_F32_MFMA=16; NP < 512 always).  Same rate, same operand bytes per flop — the 32x32 form is HALF the MFMA
// instructions (64 cycles each instead of 32), which leaves the issue slots the LDS reads, the slab / W loads and the
// barrier need: the 16x16 kernel ran its matrix pipe 81 % busy.  The two forms want W packed differently.
static bool f32_use_mfma32(int64_t NP) {
  static const bool off = dbg_env("GPBO_F32_MFMA") && atoi(dbg_env("GPBO_F32_MFMA")) == 16;
  static const bool rt2 = dbg_env("GPBO_F32_RT") && dbg_env("GPBO_F32_RT")[0] == '2';
  return !off && !rt2 && NP >= 512;
}

constexpr int F32_CANDS = 64;
constexpr int F32_BK = 32;
constexpr int F32_STRIDE = 80;   // floats per k-row of the stage tile: 64 + 16 -> the two k-rows of a 32-lane group 16 banks apart

struct PostArgsF32 {
  const float* Wp;      // packed (pack_w32_kernel)
  const float* Kst;     // [NP][ldk]
  double* part;         // [nchunks][Mp]
  int NP;
  int64_t Mp;
  int nchunks;
  int n_ctiles;
  int64_t ldk;
  int64_t m0;
};

// W -> fp32 A fragments of v_mfma_f32_16x16x4_f32: for slab s (32 rows), k-quad q (16 columns), tile t (16 rows):
// 64 lanes x 4 floats contiguous; lane l, element e holds W[32 s + 16 t + (l & 15)][16 q + 4 e + (l >> 4)].
__global__ __launch_bounds__(256) void pack_w32_kernel(const double* __restrict__ W, float* __restrict__ Wp,
                                                       int64_t N, int64_t NP) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= NP * NP) return;
  const int e = (int)(idx & 3);
  const int lane = (int)((idx >> 2) & 63);
  const int t = (int)((idx >> 8) & 1);
  const int64_t sq = idx >> 9;
  const int64_t quads = NP / 16;
  const int64_t s = sq / quads, q = sq - s * quads;
  const int64_t row = 32 * s + 16 * t + (lane & 15);
  const int64_t colx = 16 * q + 4 * e + (lane >> 4);
  float v = 0.f;
  if (row < N && colx < N && colx <= row) v = (float)W[row * NP + colx];
  Wp[idx] = v;
}

// W -> fp32 A fragments of v_mfma_f32_32x32x2_f32: for slab s (64 rows), k-quad q (16 columns), tile t (32 rows), half h
// (8 columns): 64 lanes x 4 floats contiguous; lane l, element e holds W[64 s + 32 t + (l & 31)][16 q + 8 h + 2 e + (l >> 5)].
__global__ __launch_bounds__(256) void pack_w32x_kernel(const double* __restrict__ W, float* __restrict__ Wp,
                                                        int64_t N, int64_t NP) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= NP * NP) return;
  const int e = (int)(idx & 3);
  const int lane = (int)((idx >> 2) & 63);
  const int h = (int)((idx >> 8) & 1);
  const int t = (int)((idx >> 9) & 1);
  const int64_t sq = idx >> 10;
  const int64_t quads = NP / 16;
  const int64_t s = sq / quads, q = sq - s * quads;
  const int64_t row = 64 * s + 32 * t + (lane & 31);
  const int64_t colx = 16 * q + 8 * h + 2 * e + (lane >> 5);
  float v = 0.f;
  if (row < N && colx < N && colx <= row) v = (float)W[row * NP + colx];
  Wp[idx] = v;
}

int launch_pack_w32(gpbo_ctx* ctx, Model& m) {
  const int64_t total = m.NP * m.NP;
  if (f32_use_mfma32(m.NP)) {
    pack_w32x_kernel<<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream>>>(m.W, m.Wp32, m.N, m.NP);
    GPBO_HIP(ctx, hipGetLastError());
    return GPBO_OK;
  }
  pack_w32_kernel<<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream>>>(m.W, m.Wp32, m.N, m.NP);
  GPBO_HIP(ctx, hipGetLastError());
  return GPBO_OK;
}

// (train points staged through LDS and read back as broadcasts: see kstar_gen_kernel in posterior_kernel_v2.hip)
constexpr int GEN32_CH = 64;
template <int DP, int KERNEL>
__global__ __launch_bounds__(256) void kstar_gen_f32_kernel(const double* __restrict__ Xs, const double* __restrict__ alpha,
                                                            const double* __restrict__ Xcs, float* __restrict__ Kst,
                                                            int64_t ldk, int NP, double* __restrict__ mu_part,
                                                            int64_t Mp, int64_t m0) {
  __shared__ __attribute__((aligned(16))) double xs[GEN32_CH * DP];
  __shared__ double al[GEN32_CH];
  const int64_t ml = (int64_t)blockIdx.x * 256 + threadIdx.x;
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
  for (int kc = k0; kc < k1; kc += GEN32_CH) {
    __syncthreads();
    {
      const double2* src = reinterpret_cast<const double2*>(Xs + (int64_t)kc * DP);
      double2* dst = reinterpret_cast<double2*>(xs);
      for (int e = threadIdx.x; e < GEN32_CH * DP / 2; e += 256) dst[e] = src[e];
      if (threadIdx.x < GEN32_CH) al[threadIdx.x] = alpha[kc + threadIdx.x];
    }
    __syncthreads();
    if (live) {
#pragma unroll 2
      for (int kk = 0; kk < GEN32_CH; kk += 2) {
        const double* xr = xs + kk * DP;
        double d2a = 0.0, d2b = 0.0;
#pragma unroll
        for (int t = 0; t < DP; ++t) {
          const double da = xc[t] - xr[t], db = xc[t] - xr[DP + t];
          d2a = fma(da, da, d2a);
          d2b = fma(db, db, d2b);
        }
        const double ka = gpbo_kernel_value<KERNEL>(d2a), kb = gpbo_kernel_value<KERNEL>(d2b);
        const int k = kc + kk;
        Kst[(int64_t)k * ldk + ml] = (float)ka;
        Kst[(int64_t)(k + 1) * ldk + ml] = (float)kb;
        mu = fma(ka, al[kk], mu);
        mu = fma(kb, al[kk + 1], mu);
      }
    }
  }
  if (live) mu_part[(int64_t)blockIdx.y * Mp + m0 + ml] = mu;
}

// RT = 16-row MFMA tiles per wave: 2 (wave = 32 rows, workgroup chunk = 256 rows) or 4 (64 rows / 512 rows: half
// the LDS B-fragment reads and slab re-reads per MFMA).  p.nchunks counts chunks of 8 * 16 * RT rows.
template <int RT>
__global__ __launch_bounds__(512, 4) void posterior_kernel_f32(PostArgsF32 p) {
  __shared__ __attribute__((aligned(16))) float Ks[2 * F32_BK * F32_STRIDE];   // 20 KiB
  constexpr int WROWS = 16 * RT;          // rows per wave
  constexpr int CROWS = 8 * WROWS;        // rows per workgroup chunk

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int bid = blockIdx.x;
  const int r = p.nchunks - 1 - bid / p.n_ctiles;
  const int ct = bid - (bid / p.n_ctiles) * p.n_ctiles;
  const int NP = p.NP;
  const int k_end = min(NP, (r + 1) * CROWS);
  const int n_stages = (k_end + F32_BK - 1) / F32_BK;   // NP is a multiple of 64, so k_end is a multiple of 32

  const int wrow0 = r * CROWS + wave * WROWS;           // first row of this wave
  const bool active = wrow0 < NP;
  const int64_t quads = NP / 16;
  const int wrow_ld = active ? wrow0 : (NP - WROWS);    // inactive waves stream valid rows; their sums are dropped
  // packed layout is per 32-row slab: [slab32][quad][tile2][lane] float4
  const f4* wp0 = reinterpret_cast<const f4*>(p.Wp) + (int64_t)(wrow_ld / 32) * quads * 128 + lane;
  const f4* wp1 = wp0 + quads * 128;                    // second 32-row slab (RT == 4)

  f4 acc[RT][4];
#pragma unroll
  for (int t = 0; t < RT; ++t)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[t][j] = f4{0.f, 0.f, 0.f, 0.f};

  // this thread's 4 slab elements of a stage: train points 4*wave .. 4*wave+3, candidate = lane
  auto ld_stage = [&](int stage, float (&kv)[4]) {
    const float* src = p.Kst + (int64_t)(stage * F32_BK + wave * 4) * p.ldk + (int64_t)ct * F32_CANDS + lane;
#pragma unroll
    for (int e = 0; e < 4; ++e) kv[e] = src[(int64_t)e * p.ldk];
  };
  auto st_stage = [&](const float (&kv)[4], int buf) {
#pragma unroll
    for (int e = 0; e < 4; ++e) Ks[(buf * F32_BK + wave * 4 + e) * F32_STRIDE + lane] = kv[e];
  };
  // A fragments of one k-quad (16 columns): [tile] float4
  auto loadA = [&](int kquad, f4(&a)[RT]) {
    a[0] = wp0[((int64_t)kquad * 2 + 0) * 64];
    a[1] = wp0[((int64_t)kquad * 2 + 1) * 64];
    if constexpr (RT == 4) {
      a[2] = wp1[((int64_t)kquad * 2 + 0) * 64];
      a[3] = wp1[((int64_t)kquad * 2 + 1) * 64];
    }
  };
  auto mma_quad = [&](int buf, int qq, const f4(&a)[RT]) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float* kb = Ks + (buf * F32_BK + qq * 16 + e * 4 + (lane >> 4)) * F32_STRIDE + (lane & 15);
#pragma unroll
      for (int jt = 0; jt < 4; ++jt) {
        const float b = kb[jt * 16];
#pragma unroll
        for (int t = 0; t < RT; ++t) acc[t][jt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t][e], b, acc[t][jt], 0, 0, 0);
      }
    }
  };

  {
    float kv0[4];
    ld_stage(0, kv0);
    st_stage(kv0, 0);
  }
  f4 aA[RT], aB[RT];
  loadA(0, aA);
  __syncthreads();

  const int n_full = r * (CROWS / F32_BK);
  int s = 0;
  for (; s < n_full; ++s) {
    const int buf = s & 1;
    float kv[4];
    loadA(2 * s + 1, aB);
    ld_stage(s + 1, kv);
    mma_quad(buf, 0, aA);
    loadA(2 * s + 2, aA);
    mma_quad(buf, 1, aB);
    st_stage(kv, buf ^ 1);
    __syncthreads();
  }
  for (; s < n_stages; ++s) {
    const int buf = s & 1;
    const bool has_next = (s + 1 < n_stages);
    const bool domma = (s * F32_BK <= wrow0 + WROWS - 1);
    const bool domma_next = has_next && ((s + 1) * F32_BK <= wrow0 + WROWS - 1);
    if (domma) loadA(2 * s + 1, aB);
    float kv[4];
    if (has_next) ld_stage(s + 1, kv);
    if (domma) {
      mma_quad(buf, 0, aA);
      mma_quad(buf, 1, aB);
    }
    if (has_next) st_stage(kv, buf ^ 1);
    if (domma_next) loadA(2 * s + 2, aA);
    __syncthreads();
  }

  // epilogue: squares in fp32 per lane, everything beyond that in fp64, fixed order
  double* red = reinterpret_cast<double*>(Ks);   // [8][64] doubles = 4 KiB
#pragma unroll
  for (int jt = 0; jt < 4; ++jt) {
    float vs = 0.f;
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) vs = fmaf(acc[t][jt][rr], acc[t][jt][rr], vs);
    double v = (double)vs;
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    if (lane < 16) red[wave * F32_CANDS + jt * 16 + lane] = active ? v : 0.0;
  }
  __syncthreads();
  if (tid < F32_CANDS) {
    double v = 0.0;
#pragma unroll
    for (int w = 0; w < 8; ++w) v += red[w * F32_CANDS + tid];
    p.part[(int64_t)r * p.Mp + p.m0 + (int64_t)ct * F32_CANDS + tid] = v;
  }
}

// The same pipeline on v_mfma_f32_32x32x2_f32: workgroup chunk = 512 rows = 16 tiles of 32 rows, 8 waves, 32 train points
// per stage.  Wave w owns tiles w and 15 - w (2 x 2 tiles of 32 x 32 with the 64 candidates = 64 accumulator VGPRs, as the
// 4 x 4 tiles of 16 x 16): W is lower triangular, so inside the chunk's diagonal block a tile only needs the stages up to
// its own rows, and the pairing gives every wave the same 17 tile-stages there (64 contiguous rows per wave gave wave 0 two
// stages and wave 7 sixteen: the workgroup ran at the pace of its last wave).  The stage loop is three branch-free loops
// (both tiles / later tile / none), look-ahead indices clamped instead of tested — the structure of posterior_kernel_v2.
// A fragment: lane l holds W[row l & 31][k l >> 5]; B fragment: k*[k l >> 5][candidate l & 31]; C/D: column l & 31, rows
// (reg & 3) + 8 (reg >> 2) + 4 (l >> 5) (cdna_hip_programming.md §3) — so a lane already holds 16 rows of ONE candidate and
// the sum of squares needs a single cross-lane add (lane ^ 32).
// BKX train points per stage (one workgroup barrier per stage).  Measured on the C5 shard (two GPs): BKX = 32 276.3 ms,
// BKX = 64 286.4 ms (twice the slab loads in flight per thread, 122 VGPRs) — 32 it is.
template <int BKX>
__global__ __launch_bounds__(512, 4) void posterior_kernel_f32x(PostArgsF32 p) {
  // Stage tile: [k-pair][candidate half (32)][k in pair (2)][candidate (32)] floats = blocks of 64 floats (256 B) holding
  // exactly one B-fragment read of v_mfma_f32_32x32x2_f32 (lane l -> element l), so every read of both stage buffers is the
  // lane's constant address + an immediate multiple of 256 B (ds_read2st64_b32) — as in posterior_kernel_v2.
  __shared__ __attribute__((aligned(16))) float Ks[2 * BKX * 64];   // 16 KiB at BKX = 32
  constexpr int EX = BKX / 8;      // slab elements per thread per stage
  constexpr int QX = BKX / 16;     // k-quads (16 columns of W) per stage
  constexpr int CROWS = 512, CT = CROWS / 32;
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int bid = blockIdx.x;
  const int r = p.nchunks - 1 - bid / p.n_ctiles;
  const int ct = bid - (bid / p.n_ctiles) * p.n_ctiles;
  const int NP = p.NP;
  const int k_end = min(NP, (r + 1) * CROWS);
  const int n_stages = k_end / BKX;                       // NP is a multiple of 64
  const int tileA = r * CT + wave, tileB = r * CT + CT - 1 - wave;    // global 32-row tiles (earlier / later)
  const int rowA0 = tileA * 32, rowB0 = tileB * 32;
  const bool activeA = rowA0 < NP, activeB = rowB0 < NP;   // false only in a ragged last chunk
  const int64_t quads = NP / 16;
  const int tA = activeA ? tileA : 0, tB = activeB ? tileB : 0;        // inactive tiles stream tile 0 (sums dropped)
  // packed: [slab64][quad][tile2][half2][lane] float4
  // operands through buffer descriptors: wave-uniform bases, the lane as a constant offset, the walk along k as scalar
  // offsets (no per-lane 64-bit address arithmetic between the MFMAs; see posterior_kernel_v2.hip)
  const f4* wpA = reinterpret_cast<const f4*>(p.Wp) + ((int64_t)(tA >> 1) * quads * 4 + (tA & 1) * 2) * 64;
  const f4* wpB = reinterpret_cast<const f4*>(p.Wp) + ((int64_t)(tB >> 1) * quads * 4 + (tB & 1) * 2) * 64;
  constexpr int BUF_FLAGS = 0x00020000;
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<f4*>(wpA), 0, 0x7fffffff, BUF_FLAGS);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<f4*>(wpB), 0, 0x7fffffff, BUF_FLAGS);
  const unsigned voff16 = (unsigned)lane * 16u, voff4 = (unsigned)lane * 4u;

  f16v acc[2][2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[t][u][e] = 0.f;

  auto ld_stage = [&](int stage, float (&kv)[EX]) {
    const float* src = p.Kst + (int64_t)(stage * BKX + wave * EX) * p.ldk + (int64_t)ct * F32_CANDS;   // wave-uniform
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, 0x7fffffff, BUF_FLAGS);
    const unsigned row = (unsigned)p.ldk * 4u;
#pragma unroll
    for (int e = 0; e < EX; ++e)
      kv[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff4, (unsigned)e * row, 0));
  };
  auto st_stage = [&](const float (&kv)[EX], int buf) {
#pragma unroll
    for (int e = 0; e < EX; ++e) {
      const int k = wave * EX + e;
      Ks[((buf * (BKX / 2) + (k >> 1)) * 2 + (lane >> 5)) * 64 + (k & 1) * 32 + (lane & 31)] = kv[e];
    }
  };
  auto loadA = [&](int kquad, f4(&a)[2][2]) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const unsigned so = (unsigned)kquad * 4096u + (unsigned)h * 1024u;
      a[0][h] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rsA, voff16, so, 0));
      a[1][h] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rsB, voff16, so, 0));
    }
  };
  // MODE 2: both tiles, 1: the later tile only
  auto mma_quad = [&](int buf, int qq, const f4(&a)[2][2], auto mode) {
    constexpr int MODE = decltype(mode)::value;
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float* kb = Ks + ((buf * (BKX / 2) + qq * 8 + h * 4 + e) * 2) * 64 + lane;
        const float b0 = kb[0], b1 = kb[64];
        if constexpr (MODE == 2) {
          acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0][h][e], b0, acc[0][0], 0, 0, 0);
          acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0][h][e], b1, acc[0][1], 0, 0, 0);
        }
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1][h][e], b0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1][h][e], b1, acc[1][1], 0, 0, 0);
      }
  };

  {
    float kv0[EX];
    ld_stage(0, kv0);
    st_stage(kv0, 0);
  }
  f4 aA[2][2], aB[2][2];
  loadA(0, aA);
  __syncthreads();

  const int last_stage = n_stages - 1;
  const int last_quad = (int)quads - 1;
  auto stage = [&](int st, auto mode) {
    constexpr int MODE = decltype(mode)::value;
    const int buf = st & 1;
    float kv[EX];
    if constexpr (MODE > 0) loadA(min(QX * st + 1, last_quad), aB);
    ld_stage(min(st + 1, last_stage), kv);
    __builtin_amdgcn_sched_barrier(0);
    // k-quads alternate between the two fragment registers; each is refilled (two quads ahead) right after its use — the
    // last one of the stage at the top of the next stage
    if constexpr (MODE > 0) {
#pragma unroll
      for (int qq = 0; qq < QX; ++qq) {
        if (qq & 1) mma_quad(buf, qq, aB, mode);
        else mma_quad(buf, qq, aA, mode);
        if (qq + 1 < QX || (qq & 1) == 0) {
          if (qq & 1) loadA(min(QX * st + qq + 2, last_quad), aB);
          else loadA(min(QX * st + qq + 2, last_quad), aA);
        }
      }
    }
    st_stage(kv, buf ^ 1);      // after the last LDS read of this stage (for st == last_stage nobody reads it)
    __syncthreads();
  };
  using both_t = std::integral_constant<int, 2>;
  using later_t = std::integral_constant<int, 1>;
  // stages 0 .. sA: both tiles; sA + 1 .. sB: the later tile only; beyond: none (W is lower triangular)
  const int sA = min(last_stage, (rowA0 + 31) / BKX);
  const int sB = min(last_stage, (rowB0 + 31) / BKX);
  int s = 0;
  for (; s <= sA; ++s) stage(s, both_t{});
  for (; s <= sB; ++s) stage(s, later_t{});
  for (; s <= last_stage; ++s) stage(s, std::integral_constant<int, 0>{});

  // epilogue: squares in fp32 per lane (16 rows of one candidate per tile), everything beyond that in fp64, fixed order
  double* red = reinterpret_cast<double*>(Ks);   // [8][64] doubles = 4 KiB
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    float va = 0.f, vb = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      va = fmaf(acc[0][u][e], acc[0][u][e], va);
      vb = fmaf(acc[1][u][e], acc[1][u][e], vb);
    }
    double v = (activeA ? (double)va : 0.0) + (activeB ? (double)vb : 0.0);
    v += __shfl_xor(v, 32);
    if (lane < 32) red[wave * F32_CANDS + u * 32 + lane] = v;
  }
  __syncthreads();
  if (tid < F32_CANDS) {
    double v = 0.0;
#pragma unroll
    for (int w = 0; w < 8; ++w) v += red[w * F32_CANDS + tid];
    p.part[(int64_t)r * p.Mp + p.m0 + (int64_t)ct * F32_CANDS + tid] = v;
  }
}

template <int DP, int KERNEL>
static int launch_gen32_t(gpbo_ctx* ctx, Model& m, float* Kst, int64_t ldk, int64_t Mp, int64_t m0, int nchunks) {
  dim3 grid((unsigned)((ldk + 255) / 256), (unsigned)nchunks);
  kstar_gen_f32_kernel<DP, KERNEL><<<grid, dim3(256), 0, ctx->stream>>>(m.Xs, m.alpha, ctx->Xcs, Kst, ldk, (int)m.NP,
                                                                          ctx->mu_part, Mp, m0);
  GPBO_HIP(ctx, hipGetLastError());
  return GPBO_OK;
}

template <int KERNEL>
static int launch_gen32_k(gpbo_ctx* ctx, Model& m, float* Kst, int64_t ldk, int64_t Mp, int64_t m0, int nchunks) {
  switch (m.DP) {
    case 4: return launch_gen32_t<4, KERNEL>(ctx, m, Kst, ldk, Mp, m0, nchunks);
    case 8: return launch_gen32_t<8, KERNEL>(ctx, m, Kst, ldk, Mp, m0, nchunks);
    case 16: return launch_gen32_t<16, KERNEL>(ctx, m, Kst, ldk, Mp, m0, nchunks);
    case 32: return launch_gen32_t<32, KERNEL>(ctx, m, Kst, ldk, Mp, m0, nchunks);
    case 64: return launch_gen32_t<64, KERNEL>(ctx, m, Kst, ldk, Mp, m0, nchunks);
  }
  GPBO_FAIL(ctx, GPBO_ERR_UNSUPPORTED, "posterior: unsupported padded dimension");
}

// fp32 pipeline per candidate slab; the slab buffer (ctx->kst, sized in doubles) is shared with the fp64 path.
int launch_posterior_f32(gpbo_ctx* ctx, Model& m, int64_t Mp, int nchunks, int* part_chunks) {
  *part_chunks = nchunks;
  const int64_t budget = kstar_slab_budget_bytes(ctx, Mp * m.NP * 4);   // as the fp64 path (posterior_kernel_v2.hip)
  int64_t ms = budget / (m.NP * 4);
  if (ms > (int64_t)160 * 1000 * 1000) ms = (int64_t)160 * 1000 * 1000;   // 32-bit buffer offsets of a stage's rows (f32x kernel)
  ms = ms / 128 * 128;
  if (ms < 128) GPBO_FAIL(ctx, GPBO_ERR_HIP, "posterior: not enough device memory for one k* slab");
  if (ms > Mp) ms = Mp;
  int rc;
  if ((rc = ensure(ctx, &ctx->kst, &ctx->cap_kst, (ms * m.NP + 1) / 2))) return rc;
  float* kst = reinterpret_cast<float*>(ctx->kst);
  for (int64_t m0 = 0; m0 < Mp; m0 += ms) {
    const int64_t ldk = (Mp - m0 < ms) ? (Mp - m0) : ms;
    if (m.kernel == GPBO_KERNEL_MATERN25) rc = launch_gen32_k<GPBO_KERNEL_MATERN25>(ctx, m, kst, ldk, Mp, m0, nchunks);
    else rc = launch_gen32_k<GPBO_KERNEL_RBF>(ctx, m, kst, ldk, Mp, m0, nchunks);
    if (rc) return rc;
    PostArgsF32 a;
    a.Wp = m.Wp32; a.Kst = kst; a.part = ctx->part; a.NP = (int)m.NP; a.Mp = Mp;
    a.n_ctiles = (int)(ldk / F32_CANDS); a.ldk = ldk; a.m0 = m0;
    // wave tile: 64 rows (chunks of 512 rows) by default; GPBO_F32_RT=2 selects 32 rows (chunks of 256)
    const char* e = dbg_env("GPBO_F32_RT");
    const bool rt2 = (e && e[0] == '2') || m.NP < 512;
    const bool mf32 = f32_use_mfma32(m.NP);     // (the packed W of this fit was laid out for the same choice)
    a.nchunks = rt2 ? nchunks : (int)((m.NP + 511) / 512);
    const int64_t nblocks = (int64_t)a.n_ctiles * a.nchunks;
    if (nblocks > 0x7fffffffLL) GPBO_FAIL(ctx, GPBO_ERR_UNSUPPORTED, "posterior: grid too large; shard the candidates");
    if (mf32) posterior_kernel_f32x<32><<<dim3((unsigned)nblocks), dim3(512), 0, ctx->stream>>>(a);
    else if (rt2) posterior_kernel_f32<2><<<dim3((unsigned)nblocks), dim3(512), 0, ctx->stream>>>(a);
    else posterior_kernel_f32<4><<<dim3((unsigned)nblocks), dim3(512), 0, ctx->stream>>>(a);
    GPBO_HIP(ctx, hipGetLastError());
    *part_chunks = a.nchunks;
  }
  return GPBO_OK;
}

}  // namespace gpbo
