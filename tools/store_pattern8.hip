// Which property of a row kernel's write stream sets its ceiling?  Generalised store-only replay: tiles of 64 samples,
// rows t = 1 .. S-1 of t pair-steps, F floats per pair-step and lane (driving: 2, drone products: 6), tile stride either
// packed or a multiple of 2 MiB; 8 waves x 512 workgroups, global tile queue, LDS row queue ascending, rows descending.
//   hipcc --offload-arch=gfx950 -O3 -w tools/store_pattern8.hip -o /tmp/sp8 && /tmp/sp8
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
template <int F>
__global__ __launch_bounds__(512) void tiles(float* p, size_t stride_floats, int S, int n_tiles, unsigned* queue, int ascending) {
  __shared__ int tile_s, head;
  const int lane = threadIdx.x & 63;
  for (int first = 1;; first = 0) {
    __syncthreads();
    if (threadIdx.x == 0) {
      tile_s = first ? (int)blockIdx.x : (int)gridDim.x + (int)atomicAdd(queue, 1u);
      head = 1;
    }
    __syncthreads();
    const int tile = tile_s;
    if (tile >= n_tiles) break;
    float* base = p + (size_t)tile * stride_floats;
    for (;;) {
      int t = 0;
      if (lane == 0) t = atomicAdd(&head, 1);
      t = __builtin_amdgcn_readfirstlane(t);
      if (t >= S) break;
      float* row = base + (size_t)(t * (t - 1) / 2) * F * 64;
      if (ascending) {
        for (int k = 1; k <= t; ++k) {
          float* o = row + (size_t)(k - 1) * F * 64;
#pragma unroll
          for (int j = 0; j < F; ++j) o[j * 64 + lane] = (float)k;
        }
      } else {
        for (int k = t; k >= 1; --k) {
          float* o = row + (size_t)(k - 1) * F * 64;
#pragma unroll
          for (int j = 0; j < F; ++j) o[j * 64 + lane] = (float)k;
        }
      }
    }
  }
}
int main() {
  float* p; unsigned* q;
  const size_t cap = (size_t)3400 << 20;
  (void)hipMalloc(&p, cap); (void)hipMalloc(&q, 4);
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  auto run = [&](const char* name, int F, int S, long M, int align2m, int ascending) {
    const int n_tiles = (int)((M + 63) / 64);
    const size_t payload = (size_t)S * (S - 1) / 2 * F * 64;
    size_t stride = payload;
    if (align2m) stride = ((payload * 4 + (2u << 20) - 1) / (2u << 20)) * (2u << 20) / 4;
    if (stride * 4 * n_tiles > cap) { printf("%s: too large\n", name); return; }
    float sum = 0;
    for (int i = 0; i < 10; ++i) {
      (void)hipMemsetAsync(q, 0, 4, 0);
      (void)hipEventRecord(a);
      if (F == 2) hipLaunchKernelGGL((tiles<2>), dim3(512), dim3(512), 0, 0, p, stride, S, n_tiles, q, ascending);
      else hipLaunchKernelGGL((tiles<6>), dim3(512), dim3(512), 0, 0, p, stride, S, n_tiles, q, ascending);
      (void)hipEventRecord(b); (void)hipEventSynchronize(b);
      float ms; (void)hipEventElapsedTime(&ms, a, b);
      if (i >= 2) sum += ms;
    }
    const double bytes = (double)n_tiles * payload * 4;
    printf("%-64s %.4f ms  %.0f GB/s (%.3f)\n", name, sum / 8, bytes / (sum / 8) / 1e6, bytes / (sum / 8) / 1e6 / 8000);
  };
  for (int rep = 0; rep < 2; ++rep) {
    run("F=6 S=50 M=1e5 stride 2 MiB  (the drone products kernel)", 6, 50, 100000, 1, 0);
    run("F=6 S=50 M=1e5 packed", 6, 50, 100000, 0, 0);
    run("F=6 S=50 M=1e5 stride 2 MiB, rows ascending", 6, 50, 100000, 1, 1);
    run("F=6 S=40 M=1e5 packed (1.2 MB tiles)", 6, 40, 100000, 0, 0);
    run("F=6 S=40 M=1e5 stride 2 MiB", 6, 40, 100000, 1, 0);
    run("F=2 S=50 M=3e5 packed (627 KB tiles, same bytes)", 2, 50, 300000, 0, 0);
    run("F=2 S=40 M=4.7e5 packed (399 KB tiles, same bytes: driving)", 2, 40, 470000, 0, 0);
    run("F=2 S=40 M=125000 packed (the C5 shard)", 2, 40, 125000, 0, 0);
    run("F=6 S=50 M=25000 stride 2 MiB (a quarter of the metric batch)", 6, 50, 25000, 1, 0);
  }
  return 0;
}
