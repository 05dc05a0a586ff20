// Store-only replay of car_linearize_rows_kernel's write stream at the C5 shard (M = 125,000, S = 40): 1954 tiles of
// 1560 rows x 256 B, back to back, tiles from a global queue, row tasks t = 1 .. S-1 (2t entries each) from an LDS queue
// in ascending order, each row swept descending two 256 B stores per step -- against a linear fill of the same bytes.
//   hipcc --offload-arch=gfx950 -O3 -w tools/store_pattern_car.hip -o /tmp/spc && /tmp/spc [M]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
constexpr int S = 40;
constexpr size_t TILE_FLOATS = (size_t)S * (S - 1) / 2 * 2 * 64;   // 1560 x 64
template <int NW>
__global__ __launch_bounds__(NW * 64) void tiles(float* p, int n_tiles, unsigned* queue) {
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
    float* base = p + (size_t)tile * TILE_FLOATS;
    for (;;) {
      int t = 0;
      if (lane == 0) t = atomicAdd(&head, 1);
      t = __builtin_amdgcn_readfirstlane(t);
      if (t >= S) break;
      float* row = base + (size_t)(t * (t - 1) / 2) * 2 * 64;
      for (int k = t; k >= 1; --k) {
        float* o = row + (size_t)(k - 1) * 2 * 64;
        o[lane] = (float)k;
        o[64 + lane] = (float)t;
      }
    }
  }
}
__global__ __launch_bounds__(256) void fill(float4* p, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) p[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}
int main(int argc, char** argv) {
  const long M = argc > 1 ? atol(argv[1]) : 125000;
  const int n_tiles = (int)((M + 63) / 64);
  float* p; unsigned* q;
  (void)hipMalloc(&p, (size_t)n_tiles * TILE_FLOATS * 4); (void)hipMalloc(&q, 4);
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  const double bytes = (double)n_tiles * TILE_FLOATS * 4;
  auto time = [&](const char* name, auto launch) {
    float sum = 0;
    for (int i = 0; i < 12; ++i) {
      (void)hipMemsetAsync(q, 0, 4, 0);
      (void)hipEventRecord(a); launch(); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
      float ms; (void)hipEventElapsedTime(&ms, a, b);
      if (i >= 2) sum += ms;
    }
    printf("M=%ld %-52s %.4f ms  %.0f GB/s  (%.3f of 8 TB/s)\n", M, name, sum / 10, bytes / (sum / 10) / 1e6, bytes / (sum / 10) / 1e6 / 8000);
  };
  for (int rep = 0; rep < 2; ++rep) {
    time("kernel's own order, 8 waves x 512 workgroups", [&] { hipLaunchKernelGGL((tiles<8>), dim3(512), dim3(512), 0, 0, p, n_tiles, q); });
    time("kernel's own order, 8 waves x 256 workgroups", [&] { hipLaunchKernelGGL((tiles<8>), dim3(256), dim3(512), 0, 0, p, n_tiles, q); });
    time("kernel's own order, 4 waves x 1024 workgroups", [&] { hipLaunchKernelGGL((tiles<4>), dim3(1024), dim3(256), 0, 0, p, n_tiles, q); });
    time("linear fill, 2048 x 256 threads, 16 B per lane", [&] { hipLaunchKernelGGL(fill, dim3(2048), dim3(256), 0, 0, (float4*)p, (size_t)n_tiles * TILE_FLOATS / 4); });
    time("hipMemsetAsync", [&] { (void)hipMemsetAsync(p, 0, (size_t)n_tiles * TILE_FLOATS * 4, 0); });
  }
  return 0;
}
