// On-device candidate generation in INDEX-PARITY mode (first option): the (M, d) matrix
// TargetSpace.random_sample (bayes_opt/target_space.py:593-600) fills column by column from NumPy's legacy
// RandomState — FloatParameter.random_sample = random_state.uniform(lo, hi, M) (bayes_opt/parameter.py:86-87) —
// is produced here from the SAME MT19937 state, bit for bit, and the advanced state goes back to the caller, so
// the host RandomState continues exactly where the reference's would.  Removes the host sampling (91 ms at
// M = 2^20, d = 16 on the GPU box's EPYC) and the 134 MB upload from a suggest() without leaving the reference's
// candidate stream (candidates.hip's Philox generator is the other, non-parity, throughput mode).
//
// NumPy arithmetic restated (numpy/random/src/mt19937/mt19937.c, legacy distributions):
//   regeneration  mt[k] = mt[(k+397) mod 624] ^ (y >> 1) ^ (y & 1 ? 0x9908b0df : 0),  y = (mt[k] & 0x80000000) | (mt[k+1] & 0x7fffffff)
//   tempering     y ^= y >> 11;  y ^= (y << 7) & 0x9d2c5680;  y ^= (y << 15) & 0xefc60000;  y ^= y >> 18
//   next_double   ((a >> 5) * 2^26 + (b >> 6)) / 2^53  from two consecutive outputs a, b
//   uniform       lo + (hi - lo) * next_double          (no fused multiply-add: this unit is built -ffp-contract=off)
//
// MT19937 is a 624-word shift register with taps 0, 1 and 397: word k of the next block needs words at least 227
// positions back, so a block regenerates in three data-parallel phases (k < 227, 227 <= k < 454, k >= 454).  The chain
// over blocks is sequential — ONE workgroup walks it, double-buffered in LDS (the previous block stays readable while
// later ones are written): wave 0 regenerates (the three phases chained through registers, see mt_regenerate), waves
// 1..7 meanwhile temper the previous group's words into doubles and store them (312 per block); one workgroup
// barrier per group of four blocks.  The kernel is latency-bound by construction (53 k dependent blocks for 2^20 x 16 candidates).
#include <cstdint>

#include "gpbo_internal.h"

namespace gpbo {

constexpr int MT_N = 624, MT_M = 397;

__device__ __forceinline__ unsigned mt_twist(unsigned cur, unsigned nxt, unsigned far) {
  const unsigned y = (cur & 0x80000000u) | (nxt & 0x7fffffffu);
  return far ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
}

__device__ __forceinline__ unsigned mt_temper(unsigned y) {
  y ^= y >> 11;
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= y >> 18;
  return y;
}

// Ordering of ONE wave's LDS traffic across its lanes: DS operations of a wave execute in issue order, so the hardware
// needs nothing; the fences only stop the compiler from moving a lane's loads above the previous block's stores.
#define GPBO_WAVE_SYNC()                                   \
  do {                                                     \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); \
    __builtin_amdgcn_wave_barrier();                       \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); \
  } while (0)

constexpr int MT_RING = 16;    // block buffers in LDS (power of two; the previous group and the last block before it stay readable)
constexpr int MT_GROUP = 4;    // blocks regenerated (wave 0) / emitted (waves 1..7) between two workgroup barriers

// Block `from` -> block `to` (both 624 words in LDS), by one wave.  Lane l owns words l + 64 s of each phase, so the
// far tap of phase B (word k - 227) is the word the SAME lane produced in phase A, and phase C's is its phase-B word:
// the three phases chain through registers and only the previous block is read from LDS — all loads first, one LDS
// round trip per block.
__device__ __forceinline__ void mt_regenerate(const unsigned* __restrict__ cur, unsigned* __restrict__ nxt, int lane) {
  constexpr int LAG = MT_N - MT_M;      // 227
  unsigned a0[4], a1[4], af[4], b0[4], b1[4], c0[3], c1[3];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int k = min(lane + 64 * s, LAG - 1);
    a0[s] = cur[k]; a1[s] = cur[k + 1]; af[s] = cur[k + MT_M];
    b0[s] = cur[LAG + k]; b1[s] = cur[LAG + k + 1];
  }
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    const int k = min(2 * LAG + lane + 64 * s, MT_N - 1);
    c0[s] = cur[k]; c1[s] = cur[min(k + 1, MT_N - 1)];
  }
  unsigned vA[4], vB[4], vC[3];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    vA[s] = mt_twist(a0[s], a1[s], af[s]);     // k in [0, 227): all taps in the previous block
    vB[s] = mt_twist(b0[s], b1[s], vA[s]);     // k in [227, 454): far tap = new word k - 227
  }
  const unsigned first = __builtin_amdgcn_readlane(vA[0], 0);   // the last word's "next" is new word 0
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    const bool is_last = 2 * LAG + lane + 64 * s == MT_N - 1;
    vC[s] = mt_twist(c0[s], is_last ? first : c1[s], vB[s]);   // k in [454, 624): far tap = new word k - 227
  }
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    nxt[lane + 64 * s] = vA[s];
    nxt[LAG + lane + 64 * s] = vB[s];
  }
  if (lane + 192 < LAG) {
    nxt[lane + 192] = vA[3];
    nxt[LAG + lane + 192] = vB[3];
  }
  nxt[2 * LAG + lane] = vC[0];
  nxt[2 * LAG + lane + 64] = vC[1];
  if (2 * LAG + lane + 128 < MT_N) nxt[2 * LAG + lane + 128] = vC[2];
}

// key_io: 624 state words (in: the caller's state = block 0, out: the last block touched); pos0: words of key_io already
// consumed (0..624); T = M * d doubles written in STREAM order (out[t], i.e. the column-major [d][M] image of the
// candidate matrix: coalesced stores; transpose_stream_kernel turns it into the row-major matrix afterwards);
// n_blocks = regenerations needed.
// SUB-STREAMS (mt_jump.hip): workgroup s walks blocks (bs, be] with bs = 0 for s = 0 and 1 + s * stride otherwise,
// be = min(n_blocks, 1 + (s + 1) * stride); its start block comes from states[s] (jump-ahead), the caller's key for s = 0.
// A workgroup emits the doubles whose SECOND word lies in its blocks (s = 0 also those of block 0), so a double that
// straddles two sub-streams belongs to the later one — whose start block holds the first word.  gridDim.x = 1 is the
// sequential walk.  Block b lives in ring slot b % 16; per step wave 0 regenerates the next group of four blocks while
// waves 1..7 temper and store the doubles of the previous group; one workgroup barrier per step.
__global__ __launch_bounds__(512) void mt19937_uniform_kernel(unsigned* __restrict__ key_io, int pos0, int64_t T,
                                                              int64_t M, int d, int64_t n_blocks,
                                                              const double* __restrict__ lohi, double* __restrict__ out,
                                                              int skip,   // 1 = no emission, 2 = no regeneration (timing probes)
                                                              const unsigned* __restrict__ states, int64_t stride,
                                                              unsigned* __restrict__ key_out) {
  __shared__ unsigned ring[MT_RING][MT_N];
  __shared__ double lohi_s[2 * GPBO_MAX_DIM];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const bool generator = tid < 64;
  const int etid = tid - 64;            // 0..447 for the emitting waves: one double each per block (<= 312)
  const int64_t sidx = blockIdx.x;
  const int64_t bs = (sidx == 0) ? 0 : 1 + sidx * stride;                      // start block (given)
  int64_t be = (gridDim.x == 1) ? n_blocks : 1 + (sidx + 1) * stride;         // last block this workgroup produces
  if (be > n_blocks || sidx == (int64_t)gridDim.x - 1) be = n_blocks;
  if (bs > n_blocks) return;                                                   // nothing left for this sub-stream
  const unsigned* src = (sidx == 0) ? key_io : states + sidx * MT_N;
  for (int k = tid; k < MT_N; k += 512) ring[bs & (MT_RING - 1)][k] = src[k];
  if (tid < 2 * GPBO_MAX_DIM) lohi_s[tid] = lohi[tid];
  __syncthreads();
  const int64_t my_blocks = be - bs;
  const int64_t n_steps = (my_blocks + MT_GROUP - 1) / MT_GROUP;
  for (int64_t step = 0; step <= n_steps; ++step) {
    if (generator) {
      if (!(skip & 2)) {
        for (int j = 1; j <= MT_GROUP; ++j) {
          const int64_t b = bs + step * MT_GROUP + j;          // block to produce, from block b - 1
          if (b > be) break;
          mt_regenerate(ring[(b - 1) & (MT_RING - 1)], ring[b & (MT_RING - 1)], lane);
          GPBO_WAVE_SYNC();
        }
      }
    } else if (!(skip & 1) && !(step == 0 && sidx > 0)) {
      // Doubles whose SECOND word lies in the previous group of blocks (block 0, the caller's state, in the first
      // step of the first sub-stream).  Their stream indices t are contiguous, so the emitting lanes simply split
      // [t_lo, t_hi); a double's two words sit at virtual positions v1 = pos0 + 2t and v1 + 1, located in the ring
      // relative to b_first.
      const int64_t b_first = (step == 0) ? bs : bs + (step - 1) * MT_GROUP + 1;
      const int64_t b_last = (step == 0) ? bs : min(be, bs + (step - 1) * MT_GROUP + MT_GROUP);
      const int64_t v_lo = (int64_t)MT_N * b_first, v_hi = (int64_t)MT_N * (b_last + 1);   // [v_lo, v_hi)
      const int64_t t_lo = (v_lo - pos0 - 1 >= 0) ? (v_lo - pos0) / 2 : 0;                 // ceil((v_lo - pos0 - 1) / 2)
      int64_t t_hi = (v_hi - pos0 - 2 >= 0) ? (v_hi - pos0 - 2) / 2 + 1 : 0;               // exclusive
      if (t_hi > T) t_hi = T;
      const int cnt = (t_hi > t_lo) ? (int)(t_hi - t_lo) : 0;                              // <= 4 * 312 + 1
      const int rel0 = (int)(pos0 + 2 * t_lo - v_lo);                                      // first word of double t_lo: >= -1
      const int slot0 = (int)(b_first & (MT_RING - 1));
      const int64_t col = t_lo / M, row = t_lo - col * M;          // position of double t_lo in the [d][M] image
      const int c0 = (int)min(col, (int64_t)d - 1);
      const int cn = (c0 + 1 < d) ? c0 + 1 : c0;                   // a group spans at most two columns when M >= 1280
      const double lo0 = lohi_s[c0], rg0 = lohi_s[GPBO_MAX_DIM + c0];
      const double lo1 = lohi_s[cn], rg1 = lohi_s[GPBO_MAX_DIM + cn];
      for (int o = etid; o < cnt; o += 448) {
        const int rel1 = rel0 + 2 * o, rel2 = rel1 + 1;
        const int q2 = rel2 / MT_N;
        const unsigned w2 = ring[(slot0 + q2) & (MT_RING - 1)][rel2 - q2 * MT_N];
        unsigned w1;
        if (rel1 < 0) {
          w1 = ring[(slot0 + MT_RING - 1) & (MT_RING - 1)][MT_N - 1];
        } else {
          const int q1 = rel1 / MT_N;
          w1 = ring[(slot0 + q1) & (MT_RING - 1)][rel1 - q1 * MT_N];
        }
        const unsigned a = mt_temper(w1) >> 5, bb = mt_temper(w2) >> 6;
        const double u = ((double)a * 67108864.0 + (double)bb) / 9007199254740992.0;
        double lo_c, rg_c;
        if (M >= 1280) {
          const bool nextcol = row + o >= M;
          lo_c = nextcol ? lo1 : lo0;
          rg_c = nextcol ? rg1 : rg0;
        } else {
          const int64_t c = col + (row + o) / M;
          lo_c = lohi_s[c];
          rg_c = lohi_s[GPBO_MAX_DIM + c];
        }
        out[t_lo + o] = lo_c + rg_c * u;                                      // lo + (hi - lo) * u
      }
    }
    __syncthreads();
  }
  // the workgroup that produced the last block hands the state back (exactly one: the one with bs < n_blocks <= be,
  // or the first when nothing had to be regenerated); key_out is not key_io — other workgroups may still be reading that
  if (be == n_blocks && (bs < n_blocks || sidx == 0)) {
    const unsigned* last = ring[n_blocks & (MT_RING - 1)];
    for (int k = tid; k < MT_N; k += 512) key_out[k] = last[k];
  }
}

// stream image S[c][r] (d x M) -> candidate matrix Xc[r][c] (M x d)
__global__ __launch_bounds__(256) void transpose_stream_kernel(const double* __restrict__ S, int64_t M, int d,
                                                               double* __restrict__ Xc) {
  const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (r >= M) return;
  for (int c = 0; c < d; ++c) Xc[r * d + c] = S[(int64_t)c * M + r];
}

}  // namespace gpbo

using namespace gpbo;

extern "C" int gpbo_generate_candidates_mt19937(gpbo_ctx* ctx, int64_t M, int d, const double* lo, const double* hi,
                                                uint32_t* key, int* pos) {
  if (!ctx) return GPBO_ERR_INVALID;
  if (!lo || !hi || !key || !pos || M < 1 || d < 1 || d > GPBO_MAX_DIM || *pos < 0 || *pos > MT_N)
    GPBO_FAIL(ctx, GPBO_ERR_INVALID, "generate_candidates_mt19937: bad arguments");
  GPBO_HIP(ctx, hipSetDevice(ctx->device));
  int rc;
  if ((rc = ensure(ctx, &ctx->Xc, &ctx->cap_Xc, M * d))) return rc;
  if ((rc = ensure(ctx, &ctx->stage, &ctx->cap_stage, M * d))) return rc;
  {
    char* p = (char*)ctx->red;
    int64_t cap = ctx->cap_red;
    if ((rc = ensure(ctx, &p, &cap, (int64_t)2 * GPBO_MAX_DIM * 8 + MT_N * 4 + 4096))) return rc;
    ctx->red = p;
    ctx->cap_red = cap;
  }
  double* h = (double*)((char*)ctx->pinned_aux + PIN_AUX_CAND);   // [lo | hi - lo | key]
  for (int t = 0; t < d; ++t) { h[t] = lo[t]; h[GPBO_MAX_DIM + t] = hi[t] - lo[t]; }
  unsigned* hkey = (unsigned*)(h + 2 * GPBO_MAX_DIM);
  for (int k = 0; k < MT_N; ++k) hkey[k] = key[k];
  const size_t head = 2 * GPBO_MAX_DIM * sizeof(double);
  GPBO_HIP(ctx, hipMemcpyAsync(ctx->red, h, head + MT_N * sizeof(unsigned), hipMemcpyHostToDevice, ctx->stream));
  unsigned* dkey = (unsigned*)((char*)ctx->red + head);
  const int64_t T = M * d, words = 2 * T, avail = MT_N - *pos;
  const int64_t n_blocks = (words > avail) ? (words - avail + MT_N - 1) / MT_N : 0;
  // Sub-streams by jump-ahead (mt_jump.hip): the block chain is sequential, so a long stream is cut into S pieces whose
  // start states are computed side by side from the polynomial table of (stride, S - 1) — cached per process, hence a
  // stride that depends on (M, d) only, not on the caller's position in its block.  Short streams keep the single walk.
  int S = 1;
  int64_t stride = 0;
  {
    const char* e = getenv("GPBO_MT_STREAMS");               // 64 sub-streams by default; 1 = the sequential walk
    const int want = e ? atoi(e) : 64;
    const int64_t upper = words / MT_N + 2;                  // >= n_blocks for any pos
    if (want > 1 && upper >= (int64_t)want * (e ? 4 : 16)) {  // (short streams are not worth the jump kernels)
      S = want > 256 ? 256 : want;
      stride = (upper + S - 1) / S;
    }
  }
  // device work area: [key_out 624 | seq 34 x 624 | states S x 624] words
  if ((rc = ensure(ctx, &ctx->mt_work, &ctx->cap_mt_work, (int64_t)MT_N * (1 + 34 + (S > 1 ? S : 1))))) return rc;
  unsigned* key_out = ctx->mt_work;
  unsigned* seq_dev = key_out + MT_N;
  unsigned* states_dev = seq_dev + 34 * MT_N;
  if (S > 1) {
    if ((rc = mt_jump_states(ctx, dkey, stride, S - 1, seq_dev, states_dev, &ctx->mt_bits, &ctx->mt_offset, &ctx->mt_table_key)))
      return rc;
  }
  mt19937_uniform_kernel<<<dim3((unsigned)S), dim3(512), 0, ctx->stream>>>(dkey, *pos, T, M, d, n_blocks,
                                                                          (const double*)ctx->red, ctx->stage,
                                                                          getenv("GPBO_MT_PROBE") ? atoi(getenv("GPBO_MT_PROBE")) : 0,
                                                                          states_dev, stride, key_out);
  GPBO_HIP(ctx, hipGetLastError());
  transpose_stream_kernel<<<dim3((unsigned)((M + 255) / 256)), dim3(256), 0, ctx->stream>>>(ctx->stage, M, d, ctx->Xc);
  GPBO_HIP(ctx, hipGetLastError());
  GPBO_HIP(ctx, hipMemcpyAsync(hkey, key_out, MT_N * sizeof(unsigned), hipMemcpyDeviceToHost, ctx->stream));
  GPBO_HIP(ctx, hipStreamSynchronize(ctx->stream));
  for (int k = 0; k < MT_N; ++k) key[k] = hkey[k];
  *pos = (words <= avail) ? (int)(*pos + words) : (int)((words - avail - 1) % MT_N + 1);
  ctx->M = M;
  ctx->d_c = d;
  for (auto& m : ctx->models) m.M_post = -1;
  return GPBO_OK;
}
