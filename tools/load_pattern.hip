// Read-only bandwidth against the number of concurrent linear streams (the mirror of store_pattern3.hip): short-lived
// workgroups, workgroup b reads burst (b / K) of stream (b % K); K = 1 is a plain linear read.
// build: hipcc --offload-arch=gfx950 -O3 -w tools/load_pattern.hip -o tools/_build/load_pattern
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int BURST_B>
__global__ __launch_bounds__(256) void streams(const char* p, size_t n_bursts, int K, float* sink) {
  const size_t b = blockIdx.x;
  const size_t per = n_bursts / K;
  const size_t s = b % K, j = b / K;
  if (j >= per) return;
  const char* src = p + (s * per + j) * BURST_B;
  float acc = 0.f;
#pragma unroll
  for (int u = 0; u < BURST_B / 4096; ++u) {
    const float4 v = reinterpret_cast<const float4*>(src)[u * 256 + threadIdx.x];
    acc += v.x + v.y + v.z + v.w;
  }
  if (acc == 123.456f) sink[0] = acc;
}
int main() {
  const size_t bytes = (size_t)6 << 30;
  char* p; float* sink; (void)hipMalloc(&p, bytes); (void)hipMalloc(&sink, 4); (void)hipMemset(p, 0, bytes);
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  auto time = [&](const char* name, int K, auto launch) {
    float sum = 0;
    for (int i = 0; i < 8; ++i) {
      (void)hipEventRecord(a); launch(); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
      float ms; (void)hipEventElapsedTime(&ms, a, b);
      if (i >= 2) sum += ms;
    }
    printf("%-14s K=%5d  mean %.4f ms  %.0f GB/s\n", name, K, sum / 6, bytes / (sum / 6) / 1e6);
  };
  for (int K : {1, 8, 64, 150, 512, 4096}) {
    time("burst 16 KB", K, [&] { hipLaunchKernelGGL(streams<16384>, dim3((unsigned)(bytes / 16384)), dim3(256), 0, 0, p, bytes / 16384, K, sink); });
    time("burst 4 KB", K, [&] { hipLaunchKernelGGL(streams<4096>, dim3((unsigned)(bytes / 4096)), dim3(256), 0, 0, p, bytes / 4096, K, sink); });
  }
  return 0;
}
