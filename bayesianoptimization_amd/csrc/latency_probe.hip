// Single-wave latency / issue-cost micro-probe (gfx950): what one step of a dependent fp64 chain costs — the numbers
// the diagonal-block factorisation (chol_kernels.hip) is designed around.  Every test is 64 copies of one instruction
// pattern between two s_memtime reads; out[t] = shader cycles for the 64 copies (the empty bracket is test 0).
#include "gpbo_internal.h"

namespace gpbo {

#define R4(x) x x x x
#define R16(x) R4(x) R4(x) R4(x) R4(x)
#define R64(x) R16(x) R16(x) R16(x) R16(x)

__device__ __forceinline__ long long tick() {
  long long t;
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}

__global__ __launch_bounds__(64) void latency_probe_kernel(long long* out, double* sink, double seed) {
  __shared__ double lds[4096];
  const int lane = threadIdx.x;
  double a = seed + lane * 1e-3, b = 1.0 + 1e-9 * lane, c = 1e-12, d0 = a, d1 = a + 1, d2 = a + 2, d3 = a + 3, d4 = a + 4, d5 = a + 5, d6 = a + 6, d7 = a + 7;
  lds[lane] = a;
  __syncthreads();
  long long t[40];
  int n = 0;
  t[n++] = tick();
  t[n++] = tick();                                                                             // 0: empty bracket
  asm volatile(R64("v_fma_f64 %0, %0, %1, %2\n\t") : "+v"(a) : "v"(b), "v"(c));
  t[n++] = tick();                                                                             // 1: dependent v_fma_f64
  asm volatile(R64("v_mul_f64 %0, %0, %1\n\t") : "+v"(a) : "v"(b));
  t[n++] = tick();                                                                             // 2: dependent v_mul_f64
  asm volatile(R16("v_fma_f64 %0, %0, %4, %5\n\tv_fma_f64 %1, %1, %4, %5\n\tv_fma_f64 %2, %2, %4, %5\n\tv_fma_f64 %3, %3, %4, %5\n\t")
               : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(b), "v"(c));
  t[n++] = tick();                                                                             // 3: 4 independent chains of v_fma_f64 (64 total)
  asm volatile(R4("v_fma_f64 %0, %0, %8, %9\n\tv_fma_f64 %1, %1, %8, %9\n\tv_fma_f64 %2, %2, %8, %9\n\tv_fma_f64 %3, %3, %8, %9\n\t"
                  "v_fma_f64 %4, %4, %8, %9\n\tv_fma_f64 %5, %5, %8, %9\n\tv_fma_f64 %6, %6, %8, %9\n\tv_fma_f64 %7, %7, %8, %9\n\t")
               R4("v_fma_f64 %0, %0, %8, %9\n\tv_fma_f64 %1, %1, %8, %9\n\tv_fma_f64 %2, %2, %8, %9\n\tv_fma_f64 %3, %3, %8, %9\n\t"
                  "v_fma_f64 %4, %4, %8, %9\n\tv_fma_f64 %5, %5, %8, %9\n\tv_fma_f64 %6, %6, %8, %9\n\tv_fma_f64 %7, %7, %8, %9\n\t")
               : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(b), "v"(c));
  t[n++] = tick();                                                                             // 4: 8 independent chains (64 total)
  asm volatile(R64("v_rsq_f64 %0, %0\n\t") : "+v"(a));
  t[n++] = tick();                                                                             // 5: dependent v_rsq_f64
  {
    int s0, li = lane * 3 + 1;
    asm volatile(R64("v_readlane_b32 %0, %1, 3\n\t") : "=s"(s0) : "v"(li));
    t[n++] = tick();                                                                           // 6: 64 v_readlane_b32 (same destination, independent)
    c += s0 * 1e-300;
  }
  {
#pragma unroll
    for (int k = 0; k < 64; ++k) {
      const unsigned long long u = (unsigned long long)__double_as_longlong(a);
      const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)u, 5);
      const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), 5);
      a = fma(a, b, __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo)));
    }
    t[n++] = tick();                                                                           // 7: dependent: 2 readlanes -> v_fma_f64 reading that SGPR pair -> ...
  }
  {
    int x = lane, sacc = 0;
#pragma unroll
    for (int k = 0; k < 64; ++k) {
      const int s1 = __builtin_amdgcn_readlane(x, 7);
      x = x * 3 + s1;                                                                          // 32-bit: readlane -> v_mad using the SGPR -> readlane ...
    }
    sacc = x;
    t[n++] = tick();                                                                           // 8: dependent: readlane -> 32-bit VALU op reading that SGPR -> ...
    c += sacc * 1e-300;
  }
  {
    const unsigned addr = (unsigned)(lane * 8);
    asm volatile(R64("ds_write_b64 %1, %0\n\tds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)\n\t") : "+v"(a) : "v"(addr) : "memory");
    t[n++] = tick();                                                                           // 9: LDS write -> read -> wait round trip
    asm volatile(R64("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)\n\tv_add_u32 %1, %1, 0\n\t") : "+v"(a) : "v"(addr) : "memory");
    t[n++] = tick();                                                                           // 10: LDS read -> wait
  }
  int wl = lane;
  asm volatile(R64("v_writelane_b32 %0, s20, 3\n\t") : "+v"(wl) :: "s20");
  c += wl * 1e-300;
  t[n++] = tick();                                                                             // 11: v_writelane_b32
  int cm = lane;
  asm volatile(R64("v_cndmask_b32 %0, %0, %1, vcc\n\t") : "+v"(cm) : "v"(lane));
  c += cm * 1e-300;
  t[n++] = tick();                                                                             // 12: dependent v_cndmask_b32
  asm volatile(R64("v_fma_f64 %0, %0, %1, %2\n\tv_mov_b32 %3, %3\n\tv_mov_b32 %3, %3\n\t") : "+v"(a) : "v"(b), "v"(c), "v"(lane));
  t[n++] = tick();                                                                             // 13: dependent v_fma_f64 with two independent 32-bit VALU ops between
  asm volatile(R64("v_fma_f64 %0, %0, %2, %3\n\tv_fma_f64 %1, %1, %2, %3\n\t") : "+v"(a), "+v"(d0) : "v"(b), "v"(c));
  t[n++] = tick();                                                                             // 14: 2 independent chains (128 total)
  // ---- cross-lane broadcast through an SGPR pair, written out in registers (the compiler folds the intrinsic version)
  asm volatile("v_mov_b32 v10, %0\n\tv_mov_b32 v11, %1\n\tv_mov_b32 v12, %2\n\tv_mov_b32 v13, %3\n\t"
               "v_mov_b32 v14, %0\n\tv_mov_b32 v15, %1\n\tv_mov_b32 v16, %0\n\tv_mov_b32 v17, %1\n\t"
               :: "v"((int)__double_as_longlong(b)), "v"((int)(__double_as_longlong(b) >> 32)), "v"((int)__double_as_longlong(c)),
                  "v"((int)(__double_as_longlong(c) >> 32)) : "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17");
  t[n++] = tick();
  asm volatile(R64("v_readlane_b32 s20, v10, 5\n\tv_readlane_b32 s21, v11, 5\n\ts_nop 1\n\tv_fma_f64 v[10:11], v[10:11], v[12:13], s[20:21]\n\t")
               ::: "v10", "v11", "s20", "s21");
  t[n++] = tick();                                                                             // 16: dependent [2 readlane -> s_nop 1 -> v_fma_f64 on the pair -> readlane of the result]
  asm volatile(R64("v_readlane_b32 s20, v10, 5\n\tv_readlane_b32 s21, v11, 5\n\ts_nop 1\n\tv_fma_f64 v[14:15], v[14:15], v[12:13], s[20:21]\n\t")
               ::: "v14", "v15", "s20", "s21");
  t[n++] = tick();                                                                             // 17: [2 readlane -> s_nop 1 -> v_fma_f64], source lanes fixed: the group's issue cost
  asm volatile(R16("v_readlane_b32 s20, v10, 5\n\tv_readlane_b32 s21, v11, 5\n\tv_readlane_b32 s22, v10, 6\n\tv_readlane_b32 s23, v11, 6\n\t"
                   "v_readlane_b32 s24, v10, 7\n\tv_readlane_b32 s25, v11, 7\n\tv_readlane_b32 s26, v10, 8\n\tv_readlane_b32 s27, v11, 8\n\t"
                   "v_fma_f64 v[14:15], v[12:13], s[20:21], v[14:15]\n\tv_fma_f64 v[16:17], v[12:13], s[22:23], v[16:17]\n\t"
                   "v_fma_f64 v[14:15], v[12:13], s[24:25], v[14:15]\n\tv_fma_f64 v[16:17], v[12:13], s[26:27], v[16:17]\n\t")
               ::: "v14", "v15", "v16", "v17", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");
  t[n++] = tick();                                                                             // 18: 16 x [8 readlanes, then 4 v_fma_f64 on the four pairs] (64 fma)
  {
    const unsigned addr = 64;     // every lane the same address: broadcast read
    asm volatile(R16("ds_read_b64 v[20:21], %0\n\tds_read_b64 v[22:23], %0 offset:8\n\tds_read_b64 v[24:25], %0 offset:16\n\tds_read_b64 v[26:27], %0 offset:24\n\t"
                     "s_waitcnt lgkmcnt(0)\n\t"
                     "v_fma_f64 v[14:15], v[12:13], v[20:21], v[14:15]\n\tv_fma_f64 v[16:17], v[12:13], v[22:23], v[16:17]\n\t"
                     "v_fma_f64 v[14:15], v[12:13], v[24:25], v[14:15]\n\tv_fma_f64 v[16:17], v[12:13], v[26:27], v[16:17]\n\t")
                 :: "v"(addr) : "v14", "v15", "v16", "v17", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "memory");
    t[n++] = tick();                                                                           // 19: 16 x [4 broadcast ds_read_b64, wait, 4 v_fma_f64] (64 fma)
    asm volatile(R16("ds_read_b128 v[20:23], %0\n\tds_read_b128 v[24:27], %0 offset:16\n\t"
                     "s_waitcnt lgkmcnt(0)\n\t"
                     "v_fma_f64 v[14:15], v[12:13], v[20:21], v[14:15]\n\tv_fma_f64 v[16:17], v[12:13], v[22:23], v[16:17]\n\t"
                     "v_fma_f64 v[14:15], v[12:13], v[24:25], v[14:15]\n\tv_fma_f64 v[16:17], v[12:13], v[26:27], v[16:17]\n\t")
                 :: "v"(addr) : "v14", "v15", "v16", "v17", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "memory");
    t[n++] = tick();                                                                           // 20: 16 x [2 broadcast ds_read_b128, wait, 4 v_fma_f64]
  }
  asm volatile(R64("v_readlane_b32 s20, v10, 5\n\tv_readlane_b32 s21, v11, 5\n\ts_nop 1\n\tv_rsq_f64 v[14:15], s[20:21]\n\tv_mul_f64 v[10:11], v[14:15], v[12:13]\n\t")
               ::: "v10", "v11", "v14", "v15", "s20", "s21");
  t[n++] = tick();                                                                             // 21: dependent [2 readlane -> v_rsq_f64 of the pair -> v_mul_f64 -> readlane ...]
  asm volatile(R64("v_mul_f64 v[10:11], v[10:11], v[12:13]\n\tds_write_b64 %0, v[10:11]\n\t") :: "v"((unsigned)(lane * 8)) : "v10", "v11", "memory");
  t[n++] = tick();                                                                             // 22: dependent v_mul_f64 each followed by a ds_write_b64 of the result (no wait)
  // ---- the factorisation's per-column chain as the compiler emits it (chol_kernels.hip, last columns of a quarter), 16 copies
  asm volatile("v_mov_b32 v10, 0\n\tv_mov_b32 v11, 0x40080000\n\tv_mov_b32 v20, 0\n\tv_mov_b32 v21, 0x3ff00000\n\t"
               "v_mov_b32 v24, 0\n\tv_mov_b32 v25, 0x40100000\n\tv_mov_b32 v28, %0\n\t"
               :: "v"((unsigned)(lane * 8)) : "v10", "v11", "v20", "v21", "v24", "v25", "v28");
  t[n++] = tick();
#pragma nounroll
  for (int rep = 0; rep < 2; ++rep) {     // the same code twice: first pass (instruction cache as the launch found it), second pass (warm)
  asm volatile(R16("v_readlane_b32 s21, v11, 5\n\tv_readlane_b32 s20, v10, 5\n\ts_nop 1\n\t"
                   "v_rsq_f64_e32 v[14:15], s[20:21]\n\t"
                   "v_mul_f64 v[16:17], s[20:21], -0.5\n\tv_mul_f64 v[18:19], v[14:15], v[14:15]\n\tv_fma_f64 v[18:19], v[16:17], v[18:19], 0.5\n\t"
                   "v_fmac_f64_e32 v[14:15], v[14:15], v[18:19]\n\tv_mul_f64 v[18:19], v[14:15], v[14:15]\n\tv_fma_f64 v[16:17], v[16:17], v[18:19], 0.5\n\t"
                   "v_fmac_f64_e32 v[14:15], v[14:15], v[16:17]\n\tv_mul_f64 v[22:23], v[20:21], v[14:15]\n\t"
                   "ds_write2st64_b64 v28, v[22:23], v[14:15] offset0:16 offset1:17\n\t""v_fma_f64 v[10:11], -v[22:23], v[22:23], v[24:25]\n\t") ::: "v10", "v11", "v14", "v15", "v16", "v17", "v18", "v19", "v22", "v23", "s20", "s21", "memory");
    t[n++] = tick();
  }
  // 24, 25: 16 x column chain with ds_write2st64_b64, first and second pass over the same code
  asm volatile(R16("v_readlane_b32 s21, v11, 5\n\tv_readlane_b32 s20, v10, 5\n\ts_nop 1\n\t"
                   "v_rsq_f64_e32 v[14:15], s[20:21]\n\t"
                   "v_mul_f64 v[16:17], s[20:21], -0.5\n\tv_mul_f64 v[18:19], v[14:15], v[14:15]\n\tv_fma_f64 v[18:19], v[16:17], v[18:19], 0.5\n\t"
                   "v_fmac_f64_e32 v[14:15], v[14:15], v[18:19]\n\tv_mul_f64 v[18:19], v[14:15], v[14:15]\n\tv_fma_f64 v[16:17], v[16:17], v[18:19], 0.5\n\t"
                   "v_fmac_f64_e32 v[14:15], v[14:15], v[16:17]\n\tv_mul_f64 v[22:23], v[20:21], v[14:15]\n\t"
                   """v_fma_f64 v[10:11], -v[22:23], v[22:23], v[24:25]\n\t") ::: "v10", "v11", "v14", "v15", "v16", "v17", "v18", "v19", "v22", "v23", "s20", "s21", "memory");
  t[n++] = tick();                                                                             // 25: ... without the LDS write
  asm volatile(R16("v_readlane_b32 s21, v11, 5\n\tv_readlane_b32 s20, v10, 5\n\ts_nop 1\n\t"
                   "v_rsq_f64_e32 v[14:15], s[20:21]\n\t"
                   "v_mul_f64 v[16:17], s[20:21], -0.5\n\tv_mul_f64 v[18:19], v[14:15], v[14:15]\n\tv_fma_f64 v[18:19], v[16:17], v[18:19], 0.5\n\t"
                   "v_fmac_f64_e32 v[14:15], v[14:15], v[18:19]\n\tv_mul_f64 v[18:19], v[14:15], v[14:15]\n\tv_fma_f64 v[16:17], v[16:17], v[18:19], 0.5\n\t"
                   "v_fmac_f64_e32 v[14:15], v[14:15], v[16:17]\n\tv_mul_f64 v[22:23], v[20:21], v[14:15]\n\t"
                   "ds_write_b64 v28, v[22:23]\n\t""v_fma_f64 v[10:11], -v[22:23], v[22:23], v[24:25]\n\t") ::: "v10", "v11", "v14", "v15", "v16", "v17", "v18", "v19", "v22", "v23", "s20", "s21", "memory");
  t[n++] = tick();                                                                             // 26: ... with one ds_write_b64
  asm volatile(R16("v_readlane_b32 s21, v11, 5\n\tv_readlane_b32 s20, v10, 5\n\ts_nop 1\n\t"
                   "v_mul_f64 v[14:15], s[20:21], v[20:21]\n\t"
                   "v_mul_f64 v[16:17], s[20:21], -0.5\n\tv_mul_f64 v[18:19], v[14:15], v[14:15]\n\tv_fma_f64 v[18:19], v[16:17], v[18:19], 0.5\n\t"
                   "v_fmac_f64_e32 v[14:15], v[14:15], v[18:19]\n\tv_mul_f64 v[18:19], v[14:15], v[14:15]\n\tv_fma_f64 v[16:17], v[16:17], v[18:19], 0.5\n\t"
                   "v_fmac_f64_e32 v[14:15], v[14:15], v[16:17]\n\tv_mul_f64 v[22:23], v[20:21], v[14:15]\n\t"
                   """v_fma_f64 v[10:11], -v[22:23], v[22:23], v[24:25]\n\t") ::: "v10", "v11", "v14", "v15", "v16", "v17", "v18", "v19", "v22", "v23", "s20", "s21", "memory");
  t[n++] = tick();                                                                             // 27: ... v_rsq_f64 replaced by v_mul_f64, no LDS write
  if (lane == 0)
    for (int k = 0; k + 1 < n; ++k) out[k] = t[k + 1] - t[k];
  sink[lane] = a + d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7 + c;
}

int run_latency_probe(gpbo_ctx* ctx, long long* out_host, int n) {
  long long* out = nullptr;
  double* sink = nullptr;
  GPBO_HIP(ctx, hipMalloc((void**)&out, 32 * sizeof(long long)));
  GPBO_HIP(ctx, hipMalloc((void**)&sink, 64 * sizeof(double)));
  GPBO_HIP(ctx, hipMemset(out, 0, 32 * sizeof(long long)));
  for (int rep = 0; rep < 3; ++rep) {   // the last run is warm (instruction cache)
    latency_probe_kernel<<<dim3(1), dim3(64), 0, ctx->stream>>>(out, sink, 1.25);
    GPBO_HIP(ctx, hipGetLastError());
  }
  GPBO_HIP(ctx, hipStreamSynchronize(ctx->stream));
  long long h[32];
  GPBO_HIP(ctx, hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost));
  for (int k = 0; k < n && k < 32; ++k) out_host[k] = h[k];
  (void)hipFree(out);
  (void)hipFree(sink);
  return GPBO_OK;
}

}  // namespace gpbo
