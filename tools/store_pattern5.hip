// Placement study: the tile-stream store pattern (512 resident workgroups, one 1.9 MB tile each from a queue, 256 B per
// wave store) timed on buffers that are freed and re-allocated between rounds, and with different tile strides.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
__global__ __launch_bounds__(256) void tile_fill(float* p, size_t tile_floats, size_t stride_floats, int n_tiles, unsigned* queue) {
  __shared__ int tile_s;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (;;) {
    __syncthreads();
    if (threadIdx.x == 0) tile_s = (int)atomicAdd(queue, 1u);
    __syncthreads();
    const int tile = tile_s;
    if (tile >= n_tiles) break;
    float* base = p + (size_t)tile * stride_floats;
    const size_t rows = tile_floats / 64;
    for (size_t r = wave; r < rows; r += 4) base[r * 64 + lane] = (float)r;
  }
}
int main() {
  unsigned* q; (void)hipMalloc(&q, 4);
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  struct Case { const char* name; size_t tile_bytes; int n_tiles; int wgs; std::vector<size_t> strides; };
  const size_t K = 1024, Mi = 1 << 20;
  std::vector<Case> cases = {
      {"drone products S=50", 1881600, 1563, 512, {1881600, 1881600 + 4096, 30 * 64 * K, 2 * Mi, 2 * Mi + 256, 2 * Mi + 64 * K, 4 * Mi}},
      {"drone factored S=50", 627200, 1563, 512, {627200, 10 * 64 * K, 1 * Mi, 2 * Mi}},
      {"driving S=40", 399360, 1954, 768, {399360, 7 * 64 * K, 512 * K, 1 * Mi, 2 * Mi}},
      {"drone products S=20", 291840, 1563, 512, {291840, 5 * 64 * K, 512 * K, 2 * Mi}},
  };
  for (int round = 0; round < 3; ++round) {
    void* pad = nullptr; (void)hipMalloc(&pad, (size_t)(1 + 37 * round) << 20);
    for (auto& c : cases) {
      float* p; (void)hipMalloc(&p, (size_t)c.n_tiles * (4 * Mi));
      printf("round %d %-22s:", round, c.name);
      for (size_t st : c.strides) {
        float sum = 0;
        for (int i = 0; i < 8; ++i) {
          (void)hipMemset(q, 0, 4);
          (void)hipEventRecord(a);
          hipLaunchKernelGGL(tile_fill, dim3(c.wgs), dim3(256), 0, 0, p, c.tile_bytes / 4, st / 4, c.n_tiles, q);
          (void)hipEventRecord(b); (void)hipEventSynchronize(b);
          float ms; (void)hipEventElapsedTime(&ms, a, b);
          if (i >= 2) sum += ms;
        }
        printf("  %zu: %.0f GB/s", st, c.tile_bytes * (double)c.n_tiles / (sum / 6) / 1e6);
      }
      printf("\n");
      (void)hipFree(p);
    }
    (void)hipFree(pad);
  }
  return 0;
}
