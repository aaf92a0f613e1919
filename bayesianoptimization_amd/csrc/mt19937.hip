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
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

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

// One sub-stream = one workgroup = one descriptor: a run of doubles [t_begin, t_end) of ONE column of the reference's
// candidate matrix (stream index t = column * M + row; double t is made of the 32-bit outputs pos0 + 2t and pos0 + 2t + 1,
// counted from the start of the caller's current 624-word block).  The workgroup starts from block `bs` — the caller's
// state (state_idx < 0, bs = 0) or a jump-ahead state (mt_jump.hip) at or before the block that holds the first word —
// walks the chain to the block that holds its last word, and stores lo + (hi - lo) * u at out[t + out_base], i.e. into the
// column-major image of the rows this context keeps (transpose_stream_kernel makes the row-major matrix of it).  Block b
// lives in ring slot b % 16; per step wave 0 regenerates the next group of four blocks while waves 1..7 temper and store the
// doubles whose second word lies in the previous group; one workgroup barrier per step.  The descriptor flagged `is_last`
// (it ends at the last double of the whole matrix) also returns the last block: the state the caller's RandomState must
// continue from.
struct MtChunk {
  int64_t bs;          // block the walk starts from
  int64_t t_begin, t_end;
  int64_t out_base;    // out index = t + out_base
  int state_idx;       // < 0: the caller's state (block 0); else index into states[]
  int col;
  int is_last;
  int pad;
};

__global__ __launch_bounds__(512) void mt19937_uniform_kernel(const unsigned* __restrict__ key_in, int pos0,
                                                              const MtChunk* __restrict__ chunks,
                                                              const unsigned* __restrict__ states,
                                                              const double* __restrict__ lohi, double* __restrict__ out,
                                                              unsigned* __restrict__ key_out,
                                                              int skip) {   // 1 = no emission, 2 = no regeneration (timing probes)
  __shared__ unsigned ring[MT_RING][MT_N];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const bool generator = tid < 64;
  const int etid = tid - 64;            // 0..447 for the emitting waves: one double each per block (<= 312)
  const MtChunk c = chunks[blockIdx.x];
  const int64_t bs = c.bs;
  // block that holds the last word of the last double (>= bs by construction); an empty run still returns the state
  const int64_t v_last = (c.t_end > c.t_begin) ? (int64_t)pos0 + 2 * c.t_end - 1 : (int64_t)MT_N * bs;
  const int64_t be = max(bs, v_last / MT_N);
  const unsigned* src = (c.state_idx < 0) ? key_in : states + (int64_t)c.state_idx * MT_N;
  for (int k = tid; k < MT_N; k += 512) ring[bs & (MT_RING - 1)][k] = src[k];
  const double lo_c = lohi[c.col], rg_c = lohi[GPBO_MAX_DIM + c.col];
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
    } else if (!(skip & 1)) {
      // Doubles whose SECOND word lies in the previous group of blocks (the start block itself in the first step), cut to
      // [t_begin, t_end).  Their stream indices are contiguous, so the emitting lanes simply split [t_lo, t_hi); the two
      // words of double t sit at virtual positions v1 = pos0 + 2t and v1 + 1, located in the ring relative to b_first.
      // (t >= t_begin has its first word at or after block bs, so every word read here has been produced.)
      const int64_t b_first = (step == 0) ? bs : bs + (step - 1) * MT_GROUP + 1;
      const int64_t b_last = (step == 0) ? bs : min(be, bs + (step - 1) * MT_GROUP + MT_GROUP);
      const int64_t v_lo = (int64_t)MT_N * b_first, v_hi = (int64_t)MT_N * (b_last + 1);   // [v_lo, v_hi)
      int64_t t_lo = (v_lo - pos0 - 1 >= 0) ? (v_lo - pos0) / 2 : 0;                       // ceil((v_lo - pos0 - 1) / 2)
      int64_t t_hi = (v_hi - pos0 - 2 >= 0) ? (v_hi - pos0 - 2) / 2 + 1 : 0;               // exclusive
      if (t_lo < c.t_begin) t_lo = c.t_begin;
      if (t_hi > c.t_end) t_hi = c.t_end;
      const int cnt = (t_hi > t_lo) ? (int)(t_hi - t_lo) : 0;                              // <= 4 * 312 + 1
      const int rel0 = (int)(pos0 + 2 * t_lo - v_lo);                                      // first word of double t_lo: >= -1
      const int slot0 = (int)(b_first & (MT_RING - 1));
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
        out[t_lo + o + c.out_base] = lo_c + rg_c * u;                                     // lo + (hi - lo) * u
      }
    }
    __syncthreads();
  }
  if (c.is_last) {
    const unsigned* last = ring[be & (MT_RING - 1)];
    for (int k = tid; k < MT_N; k += 512) key_out[k] = last[k];
  }
}

// stream image S[c][r] (d x M) -> candidate matrix Xc[r][c] (M x d)
// (columns [col0, col0 + d) of a d_total-wide matrix: a mixed space assembles its matrix from several column groups)
__global__ __launch_bounds__(256) void transpose_stream_kernel(const double* __restrict__ S, int64_t M, int d,
                                                               double* __restrict__ Xc, int d_total, int col0) {
  const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (r >= M) return;
  for (int c = 0; c < d; ++c) Xc[r * d_total + col0 + c] = S[(int64_t)c * M + r];
}

}  // namespace gpbo

using namespace gpbo;

// Rows [r0, r1) of the (M, d) candidate matrix TargetSpace.random_sample(M, random_state) would return, generated from the
// state (key, pos) and left resident as this context's candidate matrix ((r1 - r0) x d).  Sub-streams: every column's run
// of rows is cut into pieces; each piece starts from a jump-ahead state on a uniform grid of block offsets (stride chosen
// from (M, d, pieces) only, so the polynomial table is built once per process) and walks at most `stride` blocks before
// its first word.  key_out / pos_out (may be NULL): the state after the WHOLE matrix — available only when the last row
// is generated here (r1 == M); *has_state says so.
// d_total / col0: the d generated columns are columns [col0, col0 + d) of a d_total-wide resident matrix (a space with
// non-float parameters is assembled group by group, gpbo_generate_candidate_columns_mt19937); d_total = d, col0 = 0
// otherwise.
static int mt_generate_rows(gpbo_ctx* ctx, int64_t M, int d, int64_t r0, int64_t r1, const double* lo, const double* hi,
                            const uint32_t* key, int pos, uint32_t* key_out, int* pos_out, int* has_state, int d_total = 0,
                            int col0 = 0) {
  GPBO_HIP(ctx, hipSetDevice(ctx->device));
  if (d_total <= 0) d_total = d;
  const int64_t Mloc = r1 - r0;
  int rc;
  ctx->raw_valid = false;
  if ((rc = ensure(ctx, &ctx->Xc, &ctx->cap_Xc, Mloc * d_total))) return rc;
  if ((rc = ensure(ctx, &ctx->stage, &ctx->cap_stage, Mloc * d))) return rc;
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
  const int64_t T = M * d, words = 2 * T, avail = MT_N - pos;
  const int64_t n_blocks = (words > avail) ? (words - avail + MT_N - 1) / MT_N : 0;   // block that holds the last word

  // ---- plan: d columns x P pieces --------------------------------------------------------------------------------------
  int want = 64;                                            // sub-streams aimed at (GPBO_MT_STREAMS); at least one per column
  if (const char* e = dbg_env("GPBO_MT_STREAMS")) want = atoi(e) > 0 ? atoi(e) : 1;
  if (want > 1024) want = 1024;
  const int64_t my_blocks = 2 * Mloc * d / MT_N + 1;        // blocks' worth of words generated here
  // a tiny stream: one run per column, each walking from the caller's block (the prefix it re-walks is tiny by definition)
  const bool tiny = 2 * M * d / MT_N < 64;
  int P = 1;                                                // pieces per column
  if (!tiny && want > d && my_blocks >= (int64_t)want * 4) P = (int)((want + d - 1) / d);
  // uniform grid of jump targets: block 1 + k * stride.  A quarter of a piece: a piece walks < stride idle blocks.
  const int64_t piece_blocks = std::max<int64_t>(1, 2 * ((Mloc + P - 1) / P) / MT_N);
  const int64_t stride = std::max<int64_t>(1, piece_blocks / 4);
  std::vector<MtChunk> chunks;
  std::vector<int> polys;        // polynomial index per state slot
  int max_k = 0;
  for (int col = 0; col < d; ++col) {
    for (int p = 0; p < P; ++p) {
      MtChunk c{};
      const int64_t ra = r0 + Mloc * p / P, rb = r0 + Mloc * (p + 1) / P;
      c.t_begin = (int64_t)col * M + ra;
      c.t_end = (int64_t)col * M + rb;
      c.out_base = (int64_t)col * Mloc - ((int64_t)col * M + r0);
      c.col = col;
      c.is_last = (c.t_end == T) ? 1 : 0;
      const int64_t bneed = ((int64_t)pos + 2 * c.t_begin) / MT_N;       // block of the first word
      if (bneed == 0 || tiny) {
        c.bs = 0;
        c.state_idx = -1;
      } else {
        const int64_t k = (bneed - 1) / stride;
        c.bs = 1 + k * stride;
        c.state_idx = (int)polys.size();
        polys.push_back((int)k);
        if (k > max_k) max_k = (int)k;
      }
      chunks.push_back(c);
    }
  }
  if (M * (int64_t)d >= 1 && chunks.empty()) GPBO_FAIL(ctx, GPBO_ERR_INVALID, "generate_candidates_mt19937: empty plan");
  // when the last double is not generated here, nobody returns the state
  const bool owns_end = (r1 == M);
  // device work area: [key_out 624 | seq 34 x 624 | states n x 624 | windows n x 624] words ; descriptors + polynomial indices
  const int n_states = (int)polys.size();
  if ((rc = ensure(ctx, &ctx->mt_work, &ctx->cap_mt_work, (int64_t)MT_N * (1 + 34 + 2 * std::max(1, n_states))))) return rc;
  unsigned* key_out_dev = ctx->mt_work;
  unsigned* seq_dev = key_out_dev + MT_N;
  unsigned* states_dev = seq_dev + 34 * MT_N;
  unsigned* windows_dev = states_dev + (int64_t)MT_N * std::max(1, n_states);
  const size_t desc_bytes = chunks.size() * sizeof(MtChunk), idx_bytes = (size_t)std::max(1, n_states) * sizeof(int);
  {
    char* p = (char*)ctx->mt_desc;
    int64_t cap = ctx->cap_mt_desc;
    if ((rc = ensure(ctx, &p, &cap, (int64_t)(desc_bytes + idx_bytes + 64)))) return rc;
    ctx->mt_desc = p;
    ctx->cap_mt_desc = cap;
  }
  MtChunk* chunks_dev = (MtChunk*)ctx->mt_desc;
  int* polys_dev = (int*)((char*)ctx->mt_desc + ((desc_bytes + 63) / 64) * 64);
  GPBO_HIP(ctx, hipMemcpyAsync(chunks_dev, chunks.data(), desc_bytes, hipMemcpyHostToDevice, ctx->stream));
  if (n_states > 0) {
    GPBO_HIP(ctx, hipMemcpyAsync(polys_dev, polys.data(), (size_t)n_states * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    if ((rc = mt_jump_states(ctx, dkey, stride, max_k, polys_dev, n_states, seq_dev, states_dev, windows_dev))) return rc;
  }
  mt19937_uniform_kernel<<<dim3((unsigned)chunks.size()), dim3(512), 0, ctx->stream>>>(
      dkey, pos, chunks_dev, states_dev, (const double*)ctx->red, ctx->stage, key_out_dev,
      dbg_env("GPBO_MT_PROBE") ? atoi(dbg_env("GPBO_MT_PROBE")) : 0);
  GPBO_HIP(ctx, hipGetLastError());
  transpose_stream_kernel<<<dim3((unsigned)((Mloc + 255) / 256)), dim3(256), 0, ctx->stream>>>(ctx->stage, Mloc, d, ctx->Xc, d_total, col0);
  GPBO_HIP(ctx, hipGetLastError());
  if (owns_end) GPBO_HIP(ctx, hipMemcpyAsync(hkey, key_out_dev, MT_N * sizeof(unsigned), hipMemcpyDeviceToHost, ctx->stream));
  GPBO_HIP(ctx, hipStreamSynchronize(ctx->stream));
  // (the pageable vectors `chunks` / `polys` were consumed by the synchronisation above)
  if (has_state) *has_state = owns_end ? 1 : 0;
  if (owns_end && key_out && pos_out) {
    for (int k = 0; k < MT_N; ++k) key_out[k] = hkey[k];
    *pos_out = (words <= avail) ? (int)(pos + words) : (int)((words - avail - 1) % MT_N + 1);
  }
  (void)n_blocks;
  ctx->M = Mloc;
  ctx->d_c = d_total;
  for (auto& m : ctx->models) m.M_post = -1;
  return GPBO_OK;
}

extern "C" int gpbo_generate_candidate_columns_mt19937(gpbo_ctx* ctx, int64_t M, int d_total, int col0, int ncols, const double* lo,
                                                       const double* hi, uint32_t* key, int* pos) {
  if (!ctx) return GPBO_ERR_INVALID;
  if (!lo || !hi || !key || !pos || M < 1 || d_total < 1 || d_total > GPBO_MAX_DIM || ncols < 1 || col0 < 0 ||
      col0 + ncols > d_total || *pos < 0 || *pos > MT_N)
    GPBO_FAIL(ctx, GPBO_ERR_INVALID, "generate_candidate_columns_mt19937: bad arguments");
  uint32_t key_new[MT_N];
  int pos_new = 0, has = 0;
  int rc = mt_generate_rows(ctx, M, ncols, 0, M, lo, hi, key, *pos, key_new, &pos_new, &has, d_total, col0);
  if (rc) return rc;
  memcpy(key, key_new, sizeof(key_new));
  *pos = pos_new;
  return GPBO_OK;
}

extern "C" int gpbo_generate_candidates_mt19937(gpbo_ctx* ctx, int64_t M, int d, const double* lo, const double* hi,
                                                uint32_t* key, int* pos) {
  if (!ctx) return GPBO_ERR_INVALID;
  if (!lo || !hi || !key || !pos || M < 1 || d < 1 || d > GPBO_MAX_DIM || *pos < 0 || *pos > MT_N)
    GPBO_FAIL(ctx, GPBO_ERR_INVALID, "generate_candidates_mt19937: bad arguments");
  uint32_t key_new[MT_N];
  int pos_new = 0, has = 0;
  int rc = mt_generate_rows(ctx, M, d, 0, M, lo, hi, key, *pos, key_new, &pos_new, &has);
  if (rc) return rc;
  memcpy(key, key_new, sizeof(key_new));
  *pos = pos_new;
  return GPBO_OK;
}

extern "C" int gpbo_generate_candidate_rows_mt19937(gpbo_ctx* ctx, int64_t M, int d, int64_t row_begin, int64_t row_end,
                                                    const double* lo, const double* hi, const uint32_t* key, int pos,
                                                    uint32_t* key_out, int* pos_out) {
  if (!ctx) return GPBO_ERR_INVALID;
  if (!lo || !hi || !key || M < 1 || d < 1 || d > GPBO_MAX_DIM || pos < 0 || pos > MT_N || row_begin < 0 ||
      row_end <= row_begin || row_end > M)
    GPBO_FAIL(ctx, GPBO_ERR_INVALID, "generate_candidate_rows_mt19937: bad arguments");
  int has = 0;
  return mt_generate_rows(ctx, M, d, row_begin, row_end, lo, hi, key, pos, key_out, pos_out, &has);
}
