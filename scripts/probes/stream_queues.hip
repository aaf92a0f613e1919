// How many HIP streams of one process run kernels side by side on an MI355X, and which pairs share a hardware queue?
// One-workgroup spin kernels (~200 us each, no resource contention): k streams created back to back, one kernel on each, the
// span from the first launch to the last completion; then every pair of the first eight streams.  Also with the streams created
// at other priorities.   hipcc --offload-arch=gfx950 scripts/probes/stream_queues.hip -o /tmp/sq && /tmp/sq
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <vector>

__global__ void spin(long long ticks, int* sink) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) {}
  if (sink && threadIdx.x == 1000) *sink = 1;
}

static double span_us(const std::vector<hipStream_t>& ss, long long ticks) {
  for (auto s : ss) hipStreamSynchronize(s);
  auto t0 = std::chrono::steady_clock::now();
  for (auto s : ss) spin<<<1, 64, 0, s>>>(ticks, nullptr);
  for (auto s : ss) hipStreamSynchronize(s);
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
}

int main() {
  const long long ticks = 20000;       // 100 MHz wall clock: 200 us
  int least = 0, greatest = 0;
  hipDeviceGetStreamPriorityRange(&least, &greatest);
  printf("priority range: least %d greatest %d\n", least, greatest);
  std::vector<hipStream_t> all(12);
  for (auto& s : all) hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  for (int w = 0; w < 3; ++w) span_us({all[0]}, ticks);
  for (int k = 1; k <= 12; ++k) {
    std::vector<hipStream_t> ss(all.begin(), all.begin() + k);
    double best = 1e30;
    for (int r = 0; r < 5; ++r) best = std::min(best, span_us(ss, ticks));
    printf("%2d streams (normal priority, created back to back): %7.1f us = %.2f kernel times\n", k, best, best / 200.0);
  }
  printf("pairs of the first 8 streams (1 = side by side, 2 = one behind the other):\n");
  for (int i = 0; i < 8; ++i) {
    printf("  %d:", i);
    for (int j = 0; j < 8; ++j) {
      if (j <= i) { printf("   ."); continue; }
      double best = 1e30;
      for (int r = 0; r < 3; ++r) best = std::min(best, span_us({all[i], all[j]}, ticks));
      printf(" %3.1f", best / 200.0);
    }
    printf("\n");
  }
  std::vector<hipStream_t> hi(4), lo(4);
  for (auto& s : hi) hipStreamCreateWithPriority(&s, hipStreamNonBlocking, greatest);
  for (auto& s : lo) hipStreamCreateWithPriority(&s, hipStreamNonBlocking, least);
  for (int k = 1; k <= 4; ++k) {
    std::vector<hipStream_t> a(hi.begin(), hi.begin() + k), b(lo.begin(), lo.begin() + k);
    double ba = 1e30, bb = 1e30;
    for (int r = 0; r < 5; ++r) { ba = std::min(ba, span_us(a, ticks)); bb = std::min(bb, span_us(b, ticks)); }
    printf("%d streams of the greatest priority: %.2f kernel times; of the least: %.2f\n", k, ba / 200.0, bb / 200.0);
  }
  {
    std::vector<hipStream_t> mix = {all[0], all[1], all[2], all[3], hi[0], lo[0]};
    double best = 1e30;
    for (int r = 0; r < 5; ++r) best = std::min(best, span_us(mix, ticks));
    printf("4 normal + 1 greatest + 1 least: %.2f kernel times\n", best / 200.0);
  }
  return 0;
}
