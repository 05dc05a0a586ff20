// Which shape of the per-tile store stream does the memory system like?  1563 tiles of 1.88 MB on 2 MiB boundaries from
// a queue; variants of how a workgroup's waves cover the tile.
#include <hip/hip_runtime.h>
#include <stdio.h>
// MODE 0: waves interleave 256 B rows (wave w: rows w, w+NW, ...)   -- one advancing front per workgroup
// MODE 1: each wave owns a contiguous 1/NW of the tile               -- NW fronts per workgroup
// MODE 2: like the real kernel: wave takes "row t" tasks of growing length 6 t x 256 B from an LDS queue, sweeps it
//         DESCENDING in 1.5 KB steps
template <int NW, int MODE>
__global__ __launch_bounds__(NW * 64) void tiles(float* p, size_t stride_floats, int n_tiles, unsigned* queue) {
  __shared__ int tile_s, head;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int S = 50;
  const size_t rows = (size_t)6 * 1225;
  for (;;) {
    __syncthreads();
    if (threadIdx.x == 0) { tile_s = (int)atomicAdd(queue, 1u); head = 1; }
    __syncthreads();
    const int tile = tile_s;
    if (tile >= n_tiles) break;
    float* base = p + (size_t)tile * stride_floats;
    if (MODE == 0) {
      for (size_t r = wave; r < rows; r += NW) base[r * 64 + lane] = (float)r;
    } else if (MODE == 1) {
      const size_t per = (rows + NW - 1) / NW;
      for (size_t r = wave * per; r < rows && r < (wave + 1) * per; ++r) base[r * 64 + lane] = (float)r;
    } else {
      for (;;) {
        int t = 0;
        if (lane == 0) t = atomicAdd(&head, 1);
        t = __builtin_amdgcn_readfirstlane(t);
        if (t >= S) break;
        float* row = base + (size_t)(t * (t - 1) / 2) * 6 * 64;
        for (int k = t; k >= 1; --k) {
          float* o = row + (size_t)(k - 1) * 6 * 64;
#pragma unroll
          for (int j = 0; j < 6; ++j) o[j * 64 + lane] = (float)k;
        }
      }
    }
  }
}
int main() {
  const int n_tiles = 1563;
  const size_t stride = (2u << 20) / 4;
  float* p; unsigned* q; (void)hipMalloc(&p, n_tiles * stride * 4); (void)hipMalloc(&q, 4);
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  const double bytes = 1563.0 * 6 * 1225 * 256;
  auto time = [&](const char* name, auto launch) {
    float sum = 0;
    for (int i = 0; i < 10; ++i) {
      (void)hipMemset(q, 0, 4);
      (void)hipEventRecord(a); launch(); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
      float ms; (void)hipEventElapsedTime(&ms, a, b);
      if (i >= 2) sum += ms;
    }
    printf("%-58s %.4f ms  %.0f GB/s\n", name, sum / 8, bytes / (sum / 8) / 1e6);
  };
#define RUN(NW, MODE, WG, NAME) time(NAME, [&] { hipLaunchKernelGGL((tiles<NW, MODE>), dim3(WG), dim3(NW * 64), 0, 0, p, stride, n_tiles, q); })
  for (int rep = 0; rep < 2; ++rep) {
    RUN(4, 0, 512, "4 waves interleaved rows, 512 wg");
    RUN(8, 0, 512, "8 waves interleaved rows, 512 wg");
    RUN(8, 0, 256, "8 waves interleaved rows, 256 wg");
    RUN(16, 0, 256, "16 waves interleaved rows, 256 wg");
    RUN(8, 1, 512, "8 waves, contiguous eighths, 512 wg");
    RUN(8, 2, 512, "8 waves, row tasks descending (the kernel's order), 512 wg");
    RUN(8, 2, 768, "8 waves, row tasks descending, 768 wg");
    RUN(4, 2, 1024, "4 waves, row tasks descending, 1024 wg");
  }
  return 0;
}
