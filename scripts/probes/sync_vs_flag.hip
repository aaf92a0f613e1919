// What does hipStreamSynchronize cost behind a short kernel, against spinning on a word the kernel writes into pinned host memory?
// A one-workgroup kernel (~40 us of spinning, like a small fit) ends with [system fence; flag = sequence number].  Per iteration:
//   a) launch -> hipStreamSynchronize returns;  b) launch -> the host sees the flag (spin) -> hipStreamSynchronize returns.
//   hipcc --offload-arch=gfx950 scripts/probes/sync_vs_flag.hip -o /tmp/svf && /tmp/svf
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>

__global__ void work(long long ticks, volatile int* flag, int seq, double* out) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) {}
  out[threadIdx.x] = (double)seq;
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) *flag = seq;
}

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static double med(std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }

int main() {
  hipStream_t s;
  hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  int* flag = nullptr;
  double* out = nullptr;
  hipHostMalloc((void**)&flag, 64, hipHostMallocDefault);
  hipHostMalloc((void**)&out, 256 * 8, hipHostMallocDefault);
  *flag = 0;
  for (long long ticks : {400LL, 4000LL}) {     // 4 us and 40 us kernels
    std::vector<double> a, b1, b2;
    int seq = 0;
    for (int it = 0; it < 300; ++it) {
      ++seq;
      double t0 = now_us();
      work<<<1, 256, 0, s>>>(ticks, flag, seq, out);
      hipStreamSynchronize(s);
      a.push_back(now_us() - t0);
      ++seq;
      t0 = now_us();
      work<<<1, 256, 0, s>>>(ticks, flag, seq, out);
      while (*(volatile int*)flag != seq) {}
      const double t1 = now_us();
      if (out[255] != (double)seq) printf("ORDER VIOLATION at %d\n", it);
      hipStreamSynchronize(s);
      b1.push_back(t1 - t0);
      b2.push_back(now_us() - t1);
    }
    printf("kernel of %lld us: launch -> synchronize %.1f us | launch -> flag seen %.1f us, then synchronize %.1f us more\n", ticks / 100,
           med(a), med(b1), med(b2));
  }
  return 0;
}
