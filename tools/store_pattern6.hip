// Do the SCATTERED per-tile accesses of the row kernels (150 noise rows read, 150 g_up rows written, 256 B each, rows
// ld*4 = 400,000 B apart) cost more than their bytes?  Tile store pattern (512 workgroups, 1.88 MB tiles on 2 MiB
// boundaries) alone, + scattered row writes, + scattered row reads, and the same with those rows tile-blocked.
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int MODE>   // 0: G only | 1: + scattered g_up writes | 2: + scattered reads too | 3: both, tile-blocked
__global__ __launch_bounds__(256) void tiles(float* G, float* gup, const float* dW, size_t tile_floats, size_t stride_floats,
                                             int n_tiles, long ld, unsigned* queue, float* sink) {
  __shared__ int tile_s;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float acc = 0.f;
  for (;;) {
    __syncthreads();
    if (threadIdx.x == 0) tile_s = (int)atomicAdd(queue, 1u);
    __syncthreads();
    const int tile = tile_s;
    if (tile >= n_tiles) break;
    if (MODE >= 2) {
      for (int r = wave; r < 150; r += 4)
        acc += (MODE == 3) ? dW[((size_t)tile * 150 + r) * 64 + lane] : dW[(size_t)r * ld + (size_t)tile * 64 + lane];
    }
    float* base = G + (size_t)tile * stride_floats;
    const size_t rows = tile_floats / 64;
    for (size_t r = wave; r < rows; r += 4) base[r * 64 + lane] = (float)r + acc;
    if (MODE >= 1) {
      for (int r = wave; r < 150; r += 4) {
        if (MODE == 3) gup[((size_t)tile * 150 + r) * 64 + lane] = acc; else gup[(size_t)r * ld + (size_t)tile * 64 + lane] = acc;
      }
    }
  }
  if (acc == 123.f) sink[0] = acc;
}
int main() {
  const int n_tiles = 1563;
  const size_t tile_floats = (size_t)6 * 1225 * 64, stride = (2u << 20) / 4;
  const long ld = 100032;
  unsigned* q; float *G, *gup, *dW, *sink;
  (void)hipMalloc(&q, 4); (void)hipMalloc(&sink, 4);
  (void)hipMalloc(&G, n_tiles * stride * 4); (void)hipMalloc(&gup, (size_t)150 * ld * 4); (void)hipMalloc(&dW, (size_t)150 * ld * 4);
  (void)hipMemset(dW, 0, (size_t)150 * ld * 4);
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  auto time = [&](const char* name, auto launch) {
    float sum = 0;
    for (int i = 0; i < 10; ++i) {
      (void)hipMemset(q, 0, 4);
      (void)hipEventRecord(a); launch(); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
      float ms; (void)hipEventElapsedTime(&ms, a, b);
      if (i >= 2) sum += ms;
    }
    printf("%-46s %.4f ms\n", name, sum / 8);
  };
  for (int rep = 0; rep < 2; ++rep) {
    time("G only", [&] { hipLaunchKernelGGL(tiles<0>, dim3(512), dim3(256), 0, 0, G, gup, dW, tile_floats, stride, n_tiles, ld, q, sink); });
    time("G + 150 scattered row writes per tile", [&] { hipLaunchKernelGGL(tiles<1>, dim3(512), dim3(256), 0, 0, G, gup, dW, tile_floats, stride, n_tiles, ld, q, sink); });
    time("G + scattered writes + 150 scattered reads", [&] { hipLaunchKernelGGL(tiles<2>, dim3(512), dim3(256), 0, 0, G, gup, dW, tile_floats, stride, n_tiles, ld, q, sink); });
    time("G + the same rows tile-blocked", [&] { hipLaunchKernelGGL(tiles<3>, dim3(512), dim3(256), 0, 0, G, gup, dW, tile_floats, stride, n_tiles, ld, q, sink); });
    time("scattered writes + reads, 768 workgroups", [&] { hipLaunchKernelGGL(tiles<2>, dim3(768), dim3(256), 0, 0, G, gup, dW, tile_floats, stride, n_tiles, ld, q, sink); });
    time("scattered writes + reads, 1024 workgroups", [&] { hipLaunchKernelGGL(tiles<2>, dim3(1024), dim3(256), 0, 0, G, gup, dW, tile_floats, stride, n_tiles, ld, q, sink); });
    time("G only, 768 workgroups", [&] { hipLaunchKernelGGL(tiles<0>, dim3(768), dim3(256), 0, 0, G, gup, dW, tile_floats, stride, n_tiles, ld, q, sink); });
  }
  return 0;
}
