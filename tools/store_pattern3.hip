// Store-only bandwidth against the NUMBER of concurrent linear streams and the burst size per workgroup:
// short-lived workgroups, workgroup b writes burst (b / K) of stream (b % K); K = 1 is a plain linear fill.
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int BURST_B>   // bytes per workgroup (256 threads): 256 .. 16384
__global__ __launch_bounds__(256) void streams(char* p, size_t n_bursts, int K) {
  const size_t b = blockIdx.x;
  const size_t per = n_bursts / K;
  const size_t s = b % K, j = b / K;
  if (j >= per) return;
  char* dst = p + (s * per + j) * BURST_B;
  if (BURST_B >= 4096) {
#pragma unroll
    for (int u = 0; u < BURST_B / 4096; ++u) reinterpret_cast<float4*>(dst)[u * 256 + threadIdx.x] = make_float4(1.f, 2.f, 3.f, 4.f);
  } else if (BURST_B == 1024) {
    reinterpret_cast<float*>(dst)[threadIdx.x] = 1.f;
  } else {   // 256 B: one wave writes, the others idle
    if (threadIdx.x < 64) reinterpret_cast<float*>(dst)[threadIdx.x] = 1.f;
  }
}
int main() {
  const size_t bytes = (size_t)3 << 30;
  char* p; (void)hipMalloc(&p, bytes);
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
  for (int K : {1, 2, 8, 64, 512, 4096}) {
    time("burst 16 KB", K, [&] { hipLaunchKernelGGL(streams<16384>, dim3((unsigned)(bytes / 16384)), dim3(256), 0, 0, p, bytes / 16384, K); });
    time("burst 4 KB", K, [&] { hipLaunchKernelGGL(streams<4096>, dim3((unsigned)(bytes / 4096)), dim3(256), 0, 0, p, bytes / 4096, K); });
    time("burst 1 KB", K, [&] { hipLaunchKernelGGL(streams<1024>, dim3((unsigned)(bytes / 1024)), dim3(256), 0, 0, p, bytes / 1024, K); });
  }
  return 0;
}
