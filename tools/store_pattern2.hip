// More store-only patterns (see store_pattern.hip): which launch shape reaches the memset blit's rate?
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int T>
__global__ __launch_bounds__(T) void gs_fill(float4* p, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * T + threadIdx.x; i < n4; i += (size_t)gridDim.x * T) p[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}
template <int CHUNK_KB, bool NT>
__global__ __launch_bounds__(256) void oneshot(float4* p, size_t n4) {
  constexpr int PER = CHUNK_KB * 1024 / 16 / 256;
  const size_t i0 = (size_t)blockIdx.x * (PER * 256);
#pragma unroll
  for (int u = 0; u < PER; ++u) {
    const size_t i = i0 + u * 256 + threadIdx.x;
    if (i < n4) {
      if (NT) { float* q = (float*)(p + i); __builtin_nontemporal_store(1.f, q); __builtin_nontemporal_store(2.f, q + 1); __builtin_nontemporal_store(3.f, q + 2); __builtin_nontemporal_store(4.f, q + 3); }
      else p[i] = make_float4(1.f, 2.f, 3.f, 4.f);
    }
  }
}
// persistent workgroups taking CHUNK_KB chunks in linear order from a queue
template <int CHUNK_KB>
__global__ __launch_bounds__(256) void queue_chunks(float4* p, size_t n4, unsigned* queue) {
  constexpr int PER = CHUNK_KB * 1024 / 16 / 256;
  __shared__ unsigned c_s;
  const size_t n_chunks = (n4 + PER * 256 - 1) / (PER * 256);
  for (;;) {
    __syncthreads();
    if (threadIdx.x == 0) c_s = atomicAdd(queue, 1u);
    __syncthreads();
    const size_t c = c_s;
    if (c >= n_chunks) break;
    const size_t i0 = c * (PER * 256);
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const size_t i = i0 + u * 256 + threadIdx.x;
      if (i < n4) p[i] = make_float4(1.f, 2.f, 3.f, 4.f);
    }
  }
}
int main() {
  const size_t n = ((size_t)3 * 50 * 49 * 64 + 4096) * 1563;
  float* p; unsigned* q;
  (void)hipMalloc(&p, n * 4); (void)hipMalloc(&q, 4);
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  auto time = [&](const char* name, auto launch) {
    float best = 1e9, sum = 0;
    for (int i = 0; i < 12; ++i) {
      (void)hipMemset(q, 0, 4);
      (void)hipEventRecord(a); launch(); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
      float ms; (void)hipEventElapsedTime(&ms, a, b);
      if (i >= 2) { sum += ms; if (ms < best) best = ms; }
    }
    printf("%-44s mean %.4f ms  best %.4f ms  %.0f GB/s\n", name, sum / 10, best, n * 4 / (sum / 10) / 1e6);
  };
  const size_t n4 = n / 4;
  time("hipMemsetD32Async", [&] { (void)hipMemsetD32Async((hipDeviceptr_t)p, 0x3f800000, n, 0); });
  time("grid-stride 1024 thr x 256", [&] { hipLaunchKernelGGL(gs_fill<1024>, dim3(256), dim3(1024), 0, 0, (float4*)p, n4); });
  time("grid-stride 1024 thr x 512", [&] { hipLaunchKernelGGL(gs_fill<1024>, dim3(512), dim3(1024), 0, 0, (float4*)p, n4); });
  time("grid-stride 1024 thr x 2048", [&] { hipLaunchKernelGGL(gs_fill<1024>, dim3(2048), dim3(1024), 0, 0, (float4*)p, n4); });
  time("grid-stride 256 thr x 65536", [&] { hipLaunchKernelGGL(gs_fill<256>, dim3(65536), dim3(256), 0, 0, (float4*)p, n4); });
  time("one-shot 4 KB", [&] { hipLaunchKernelGGL((oneshot<4, false>), dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, 0, (float4*)p, n4); });
  time("one-shot 16 KB", [&] { hipLaunchKernelGGL((oneshot<16, false>), dim3((unsigned)((n4 + 1023) / 1024)), dim3(256), 0, 0, (float4*)p, n4); });
  time("one-shot 16 KB nontemporal", [&] { hipLaunchKernelGGL((oneshot<16, true>), dim3((unsigned)((n4 + 1023) / 1024)), dim3(256), 0, 0, (float4*)p, n4); });
  time("one-shot 64 KB", [&] { hipLaunchKernelGGL((oneshot<64, false>), dim3((unsigned)((n4 + 4095) / 4096)), dim3(256), 0, 0, (float4*)p, n4); });
  time("one-shot 256 KB", [&] { hipLaunchKernelGGL((oneshot<256, false>), dim3((unsigned)((n4 + 16383) / 16384)), dim3(256), 0, 0, (float4*)p, n4); });
  time("queue 16 KB chunks, 512 wg", [&] { hipLaunchKernelGGL(queue_chunks<16>, dim3(512), dim3(256), 0, 0, (float4*)p, n4, q); });
  time("queue 16 KB chunks, 2048 wg", [&] { hipLaunchKernelGGL(queue_chunks<16>, dim3(2048), dim3(256), 0, 0, (float4*)p, n4, q); });
  time("queue 64 KB chunks, 512 wg", [&] { hipLaunchKernelGGL(queue_chunks<64>, dim3(512), dim3(256), 0, 0, (float4*)p, n4, q); });
  time("queue 64 KB chunks, 2048 wg", [&] { hipLaunchKernelGGL(queue_chunks<64>, dim3(2048), dim3(256), 0, 0, (float4*)p, n4, q); });
  return 0;
}
