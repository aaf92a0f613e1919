// Device micro-benchmarks: calibrate the two ceilings the roofline numbers are quoted against.
//   mfma_f64_peak : sustained v_mfma_f64_16x16x4_f64 rate (TFLOP/s) — the fp64 matrix peak is not in
//                   MI355X_MICROARCH.md, so it is measured next to AMD's 78.6 TFLOP/s datasheet figure.
//   hbm_copy_peak : streaming copy bandwidth (GB/s, read + write), cf. 6.29 TB/s measured in the guide.
#include "gpbo_internal.h"

namespace gpbo {

typedef double d4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void mfma_peak_kernel(double* out, int iters) {
  d4 acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = d4{0.0, 0.0, 0.0, 0.0};
  const double a = 1.0 + threadIdx.x * 1e-6, b = 0.5 - threadIdx.x * 1e-6;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[j], 0, 0, 0);
  }
  double s = 0.0;
#pragma unroll
  for (int j = 0; j < 8; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
  out[(int64_t)blockIdx.x * 256 + threadIdx.x] = s;
}

// Same MFMA stream with in-kernel clocks: s_memtime (shader cycles) and s_memrealtime (100 MHz) bracket
// the loop, so cycles per MFMA per SIMD and the sustained shader clock can be told apart.
template <int MODE>  // 0: builtin (compiler picks VGPR accumulators); 1: inline asm with AGPR accumulators;
                     // 2: the posterior GEMM's register pattern — a 2 x 4 tile grid, every MFMA another (A, B) register pair
__global__ __launch_bounds__(1024) void mfma_probe_kernel(double* out, unsigned long long* clk, int iters) {
  d4 acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = d4{0.0, 0.0, 0.0, 0.0};
  const double a = 1.0 + threadIdx.x * 1e-6, b = 0.5 - threadIdx.x * 1e-6;
  double av[2] = {a, a * 1.0000003}, bv[4] = {b, b * 0.9999991, b * 1.0000007, b * 0.9999987};
  const unsigned long long c0 = __builtin_amdgcn_s_memtime();
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; ++i) {
    if constexpr (MODE == 2) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[0], bv[u], acc[u], 0, 0, 0);
        acc[4 + u] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[1], bv[u], acc[4 + u], 0, 0, 0);
      }
      asm volatile("" : "+v"(av[0]), "+v"(av[1]), "+v"(bv[0]), "+v"(bv[1]), "+v"(bv[2]), "+v"(bv[3]));   // keep the six operand registers apart
      continue;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if constexpr (MODE == 1) {
        asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+a"(acc[j]) : "v"(a), "v"(b));
      } else {
        acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[j], 0, 0, 0);
      }
    }
  }
  if constexpr (MODE == 1) asm volatile("s_nop 15\n s_nop 15" ::: "memory");
  double s = 0.0;
#pragma unroll
  for (int j = 0; j < 8; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
  out[(int64_t)blockIdx.x * blockDim.x + threadIdx.x] = s;   // forces completion of the MFMAs before the clocks
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long c1 = __builtin_amdgcn_s_memtime();
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  if ((threadIdx.x & 63) == 0) {
    const int w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    clk[4 * w] = c1 - c0;
    clk[4 * w + 1] = r1 - r0;
    clk[4 * w + 2] = r0;          // absolute 100 MHz stamps: the span first start .. last end is the kernel's own time,
    clk[4 * w + 3] = r1;          // free of the launch and of the event bracket
  }
}

int run_mfma_probe(gpbo_ctx* ctx, int iters, int waves_per_simd, int mode, double* out4) {
  // ONE workgroup per compute unit (256 of them), 4 * waves_per_simd waves each: the dispatcher has nothing to balance, every
  // SIMD gets exactly waves_per_simd waves (256-thread workgroups — the first form of this probe — were packed 8 to a CU on
  // half of the CUs: in-kernel clocks said 62 cycles per MFMA while the event time gave 48 TFLOP/s)
  const int wps = waves_per_simd < 1 ? 1 : (waves_per_simd > 4 ? 4 : waves_per_simd);
  const int grid = 256, threads = 256 * wps;
  double* out = nullptr;
  unsigned long long* clk = nullptr;
  const int nw = grid * 4 * wps;
  GPBO_HIP(ctx, hipMalloc((void**)&out, (size_t)grid * threads * sizeof(double)));
  GPBO_HIP(ctx, hipMalloc((void**)&clk, (size_t)nw * 4 * sizeof(unsigned long long)));
  hipEvent_t e0, e1;
  GPBO_HIP(ctx, hipEventCreate(&e0));
  GPBO_HIP(ctx, hipEventCreate(&e1));
  auto kern = mode == 2 ? mfma_probe_kernel<2> : mode == 1 ? mfma_probe_kernel<1> : mfma_probe_kernel<0>;
  kern<<<dim3(grid), dim3(threads), 0, ctx->stream>>>(out, clk, 16);
  GPBO_HIP(ctx, hipEventRecord(e0, ctx->stream));
  kern<<<dim3(grid), dim3(threads), 0, ctx->stream>>>(out, clk, iters);
  GPBO_HIP(ctx, hipEventRecord(e1, ctx->stream));
  GPBO_HIP(ctx, hipEventSynchronize(e1));
  float ms = 0.f;
  GPBO_HIP(ctx, hipEventElapsedTime(&ms, e0, e1));
  std::string h((size_t)nw * 32, '\0');
  GPBO_HIP(ctx, hipMemcpy(&h[0], clk, h.size(), hipMemcpyDeviceToHost));
  const unsigned long long* c = (const unsigned long long*)h.data();
  double cyc = 0.0, rt = 0.0;
  unsigned long long first = ~0ull, last = 0;
  for (int w = 0; w < nw; ++w) {
    cyc += (double)c[4 * w]; rt += (double)c[4 * w + 1];
    if (c[4 * w + 2] < first) first = c[4 * w + 2];
    if (c[4 * w + 3] > last) last = c[4 * w + 3];
  }
  cyc /= nw; rt /= nw;
  const double mfma_per_simd = (double)iters * 8.0 * wps;
  const double flops = (double)nw * iters * 8.0 * 2048.0;
  const double span_s = (double)(last - first) * 1e-8;              // s_memrealtime: 100 MHz
  out4[0] = span_s > 0.0 ? flops / span_s / 1e12 : 0.0;            // TFLOP/s over the kernel's own span
  out4[1] = cyc / mfma_per_simd;                                    // shader cycles per MFMA per SIMD
  out4[2] = cyc / (rt / 100.0);                                     // shader MHz
  out4[3] = ms;                                                     // event-bracketed milliseconds (launch included)
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  GPBO_HIP(ctx, hipFree(out));
  GPBO_HIP(ctx, hipFree(clk));
  return GPBO_OK;
}

#ifdef GPBO_DEBUG
// Hybrid probe: per loop iteration a wave issues NM independent MFMAs and NV v_fma_f64 whose multiplier is a
// wave-uniform double fetched with scalar loads (the shape of a VALU GEMM row update: acc_r += W[r][k] * k*[k][lane]).
// Answers: how much fp64 VALU FMA throughput is available NEXT TO a saturated fp64 matrix pipe?
template <int NM, int NV>
__global__ __launch_bounds__(256) void hybrid_probe_kernel(const double* __restrict__ wsc, double* out, int iters) {
  d4 acc[NM > 0 ? NM : 1];
#pragma unroll
  for (int j = 0; j < (NM > 0 ? NM : 1); ++j) acc[j] = d4{0.0, 0.0, 0.0, 0.0};
  double va[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) va[r] = 0.0;
  const double a = 1.0 + threadIdx.x * 1e-6, b = 0.5 - threadIdx.x * 1e-6;
  double kv = 1.0 + threadIdx.x * 1e-3;
  for (int i = 0; i < iters; ++i) {
    const double* wrow = wsc + (size_t)(i & 63) * 256;   // wave-uniform -> s_load
#pragma unroll
    for (int j = 0; j < NM; ++j) acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[j], 0, 0, 0);
#pragma unroll
    for (int v = 0; v < NV; ++v) va[v & 15] = fma(wrow[v], kv, va[v & 15]);
    kv += 1e-9;
  }
  double s = 0.0;
#pragma unroll
  for (int j = 0; j < (NM > 0 ? NM : 1); ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
#pragma unroll
  for (int r = 0; r < 16; ++r) s += va[r];
  out[(int64_t)blockIdx.x * 256 + threadIdx.x] = s;
}

// out3 = { ms, MFMA TFLOP/s, VALU TFLOP/s } for config cfg: 0 = 16 MFMA only, 1 = 256 VALU only,
// 2 = 16 MFMA + 256 VALU, 3 = 16 MFMA + 128 VALU, 4 = 8 MFMA + 256 VALU.  4 waves per SIMD.
int run_hybrid_probe(gpbo_ctx* ctx, int iters, int cfg, double* out3) {
  const int grid = 256 * 4;
  double *out = nullptr, *wsc = nullptr;
  GPBO_HIP(ctx, hipMalloc((void**)&out, (size_t)grid * 256 * sizeof(double)));
  GPBO_HIP(ctx, hipMalloc((void**)&wsc, (size_t)64 * 256 * sizeof(double)));
  std::string h((size_t)64 * 256 * 8, '\0');
  double* hw = (double*)h.data();
  for (int i = 0; i < 64 * 256; ++i) hw[i] = 1e-3 * ((i * 37) % 101 - 50);
  GPBO_HIP(ctx, hipMemcpy(wsc, hw, h.size(), hipMemcpyHostToDevice));
  hipEvent_t e0, e1;
  GPBO_HIP(ctx, hipEventCreate(&e0));
  GPBO_HIP(ctx, hipEventCreate(&e1));
  int nm = 0, nv = 0;
  for (int pass = 0; pass < 2; ++pass) {
    const int it = pass == 0 ? 8 : iters;
    if (pass == 1) GPBO_HIP(ctx, hipEventRecord(e0, ctx->stream));
    switch (cfg) {
      case 0: nm = 16; nv = 0; hybrid_probe_kernel<16, 0><<<dim3(grid), dim3(256), 0, ctx->stream>>>(wsc, out, it); break;
      case 1: nm = 0; nv = 256; hybrid_probe_kernel<0, 256><<<dim3(grid), dim3(256), 0, ctx->stream>>>(wsc, out, it); break;
      case 2: nm = 16; nv = 256; hybrid_probe_kernel<16, 256><<<dim3(grid), dim3(256), 0, ctx->stream>>>(wsc, out, it); break;
      case 3: nm = 16; nv = 128; hybrid_probe_kernel<16, 128><<<dim3(grid), dim3(256), 0, ctx->stream>>>(wsc, out, it); break;
      default: nm = 8; nv = 256; hybrid_probe_kernel<8, 256><<<dim3(grid), dim3(256), 0, ctx->stream>>>(wsc, out, it); break;
    }
  }
  GPBO_HIP(ctx, hipEventRecord(e1, ctx->stream));
  GPBO_HIP(ctx, hipEventSynchronize(e1));
  float ms = 0.f;
  GPBO_HIP(ctx, hipEventElapsedTime(&ms, e0, e1));
  const double waves = (double)grid * 4.0;
  out3[0] = ms;
  out3[1] = waves * iters * nm * 2048.0 / (ms * 1e-3) / 1e12;
  out3[2] = waves * iters * nv * 128.0 / (ms * 1e-3) / 1e12;
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  GPBO_HIP(ctx, hipFree(out));
  GPBO_HIP(ctx, hipFree(wsc));
  return GPBO_OK;
}
#endif  // GPBO_DEBUG

int run_mfma_peak(gpbo_ctx* ctx, int iters, double* tflops) {
  const int grid = 256 * 2;  // 2 workgroups of 4 waves per CU -> 2 waves per SIMD
  double* out = nullptr;
  GPBO_HIP(ctx, hipMalloc((void**)&out, (size_t)grid * 256 * sizeof(double)));
  hipEvent_t e0, e1;
  GPBO_HIP(ctx, hipEventCreate(&e0));
  GPBO_HIP(ctx, hipEventCreate(&e1));
  mfma_peak_kernel<<<dim3(grid), dim3(256), 0, ctx->stream>>>(out, 16);  // warm-up
  GPBO_HIP(ctx, hipEventRecord(e0, ctx->stream));
  mfma_peak_kernel<<<dim3(grid), dim3(256), 0, ctx->stream>>>(out, iters);
  GPBO_HIP(ctx, hipEventRecord(e1, ctx->stream));
  GPBO_HIP(ctx, hipEventSynchronize(e1));
  float ms = 0.f;
  GPBO_HIP(ctx, hipEventElapsedTime(&ms, e0, e1));
  const double flops = (double)grid * 4.0 * (double)iters * 8.0 * 2048.0;
  *tflops = flops / (ms * 1e-3) / 1e12;
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  GPBO_HIP(ctx, hipFree(out));
  return GPBO_OK;
}

__global__ __launch_bounds__(256) void copy_kernel(const double2* __restrict__ in, double2* __restrict__ out, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (; i < n; i += stride) out[i] = in[i];
}

int run_copy_peak(gpbo_ctx* ctx, int64_t bytes, double* gbps) {
  const int64_t n = bytes / 16;
  double2 *a = nullptr, *b = nullptr;
  GPBO_HIP(ctx, hipMalloc((void**)&a, (size_t)n * 16));
  GPBO_HIP(ctx, hipMalloc((void**)&b, (size_t)n * 16));
  GPBO_HIP(ctx, hipMemsetAsync(a, 1, (size_t)n * 16, ctx->stream));
  hipEvent_t e0, e1;
  GPBO_HIP(ctx, hipEventCreate(&e0));
  GPBO_HIP(ctx, hipEventCreate(&e1));
  copy_kernel<<<dim3(2048), dim3(256), 0, ctx->stream>>>(a, b, n);
  GPBO_HIP(ctx, hipEventRecord(e0, ctx->stream));
  const int reps = 5;
  for (int r = 0; r < reps; ++r) copy_kernel<<<dim3(2048), dim3(256), 0, ctx->stream>>>(a, b, n);
  GPBO_HIP(ctx, hipEventRecord(e1, ctx->stream));
  GPBO_HIP(ctx, hipEventSynchronize(e1));
  float ms = 0.f;
  GPBO_HIP(ctx, hipEventElapsedTime(&ms, e0, e1));
  *gbps = 2.0 * (double)n * 16.0 * reps / (ms * 1e-3) / 1e9;
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  GPBO_HIP(ctx, hipFree(a));
  GPBO_HIP(ctx, hipFree(b));
  return GPBO_OK;
}

}  // namespace gpbo
