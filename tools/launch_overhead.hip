// What an isolated launch costs OUTSIDE its workgroups (profiles/r06_metric_timeline.txt: 35-40 us for the metric launch).
// For a few kernel shapes: HIP-event time around ONE launch, the period of launches issued back to back, and the span
// first-workgroup-start .. last-workgroup-end by the 100 MHz clock stamped inside the kernel.
//   hipcc --offload-arch=gfx950 -O3 -w tools/launch_overhead.hip -o /tmp/launch_overhead && /tmp/launch_overhead
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

__device__ __forceinline__ unsigned long long wclk() { return __builtin_readcyclecounter() * 0 + __builtin_amdgcn_s_memrealtime(); }

// mode 0: nothing; 1: nt stores of `bytes_per_wg` per workgroup (contiguous 256 B per wave instruction); 2: plain stores
__global__ __launch_bounds__(512) void probe_kernel(float* out, size_t floats_per_wg, int mode, unsigned long long* stamps) {
  extern __shared__ float lds[];
  const unsigned long long t0 = wclk();
  if (threadIdx.x == 0 && lds) lds[0] = 0.f;
  float* p = out + (size_t)blockIdx.x * floats_per_wg + threadIdx.x;
  const float v = (float)threadIdx.x;
  if (mode == 1) {
    for (size_t i = 0; i < floats_per_wg; i += 512) asm volatile("global_store_dword %0, %1, off nt" : : "v"(p + i), "v"(v) : "memory");
  } else if (mode == 2) {
    for (size_t i = 0; i < floats_per_wg; i += 512) p[i] = v;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    stamps[2 * blockIdx.x] = t0;
    stamps[2 * blockIdx.x + 1] = wclk();
  }
}

// the same probe with 768 bytes of kernel arguments (the row kernels take their parameter structs by value)
struct BigArgs { double v[88]; };
__global__ __launch_bounds__(512) void probe_bigargs_kernel(float* out, size_t floats_per_wg, int mode, unsigned long long* stamps,
                                                           const BigArgs a) {
  extern __shared__ float lds[];
  const unsigned long long t0 = wclk();
  if (threadIdx.x == 0 && lds) lds[0] = (float)a.v[blockIdx.x % 88];
  __syncthreads();
  if (threadIdx.x == 0) {
    stamps[2 * blockIdx.x] = t0;
    stamps[2 * blockIdx.x + 1] = wclk();
  }
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

int main() {
  const int grid = 256;
  const size_t total_bytes = 3ull << 30;
  float* out;
  CK(hipMalloc(&out, total_bytes));
  unsigned long long* stamps;
  CK(hipMalloc(&stamps, sizeof(unsigned long long) * 2 * 4096));
  std::vector<unsigned long long> h(2 * 4096);
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(probe_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(probe_bigargs_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  struct Case { const char* name; int g; int lds; int mode; size_t mb; };
  const Case cases[] = {{"empty, 256 wg, no LDS", 256, 0, 0, 0},        {"empty, 256 wg, 64.6 KB LDS", 256, 64600, 0, 0},
                        {"empty, 2048 wg, no LDS", 2048, 0, 0, 0},      {"nt stores 64 MB, 256 wg", 256, 0, 1, 64},
                        {"nt stores 512 MB, 256 wg", 256, 0, 1, 512},   {"nt stores 3 GB, 256 wg", 256, 0, 1, 3072},
                        {"nt stores 3 GB, 256 wg, 64.6 KB LDS", 256, 64600, 1, 3072},
                        {"plain stores 512 MB, 256 wg", 256, 0, 2, 512}, {"plain stores 3 GB, 256 wg", 256, 0, 2, 3072},
                        {"nt stores 800 MB, 512 wg", 512, 0, 1, 800},
                        {"empty, 256 wg, 64.6 KB LDS, 768 B of arguments", 256, 64600, 9, 0}};
  for (const Case& c : cases) {
    const size_t fpw = c.mb ? ((size_t)c.mb << 20) / 4 / c.g / 512 * 512 : 0;
    BigArgs big = {};
    auto launch = [&]() {
      if (c.mode == 9) hipLaunchKernelGGL(probe_bigargs_kernel, dim3(c.g), dim3(512), c.lds, 0, out, fpw, 0, stamps, big);
      else hipLaunchKernelGGL(probe_kernel, dim3(c.g), dim3(512), c.lds, 0, out, fpw, c.mode, stamps);
    };
    for (int i = 0; i < 3; ++i) launch();
    CK(hipDeviceSynchronize());
    double ev = 0, span = 0;
    const int R = 10;
    for (int i = 0; i < R; ++i) {
      CK(hipEventRecord(a, 0));
      launch();
      CK(hipEventRecord(b, 0));
      CK(hipDeviceSynchronize());
      float ms;
      CK(hipEventElapsedTime(&ms, a, b));
      ev += ms * 1e3;
      CK(hipMemcpy(h.data(), stamps, sizeof(unsigned long long) * 2 * c.g, hipMemcpyDeviceToHost));
      unsigned long long lo = ~0ull, hi = 0;
      for (int k = 0; k < c.g; ++k) { lo = std::min(lo, h[2 * k]); hi = std::max(hi, h[2 * k + 1]); }
      span += (hi - lo) * 1e-2;
    }
    // back to back: N launches, period from the stamps of the first and the last
    const int N = 20;
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a, 0));
    for (int i = 0; i < N; ++i) launch();
    CK(hipEventRecord(b, 0));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    printf("%-40s events %8.1f us   span %8.1f us   outside %6.1f us   back-to-back period %8.1f us (%.1f outside)\n", c.name, ev / R,
           span / R, (ev - span) / R, ms * 1e3 / N, ms * 1e3 / N - span / R);
  }
  return 0;
}
