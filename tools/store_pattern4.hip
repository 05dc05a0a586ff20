// Would interleaving the rows of GROUP adjacent tiles help the store stream?  512 resident workgroups, each writes
// its tile row by row (ROW_B bytes per row, 256 B per wave instruction, 4 waves); layout A: tile-major (the packed
// Jacobian today), layout B: [group][row][tile in group] (GROUP tiles' rows adjacent).  Workgroups run in lockstep
// here (equal work), which is the best case for B.
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int GROUP>
__global__ __launch_bounds__(256) void tiles(float* p, int rows, int row_floats, int n_tiles, unsigned* queue, int skew) {
  __shared__ int tile_s;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (;;) {
    __syncthreads();
    if (threadIdx.x == 0) tile_s = (int)atomicAdd(queue, 1u);
    __syncthreads();
    const int tile = tile_s;
    if (tile >= n_tiles) break;
    const int g = tile / GROUP, j = tile % GROUP;
    if (skew && (tile & 1)) __builtin_amdgcn_s_sleep(127);   // odd tiles start late
    for (int r = 0; r < rows; ++r) {
      float* base = p + ((size_t)(g * rows + r) * GROUP + j) * row_floats;
      for (int q = wave * 64; q < row_floats; q += 256) base[q + lane] = (float)r;
    }
  }
}
int main() {
  const int n_tiles = 1536, rows = 49, row_floats = 9728;   // 38 KB rows, 1.9 MB tiles
  const size_t n = (size_t)n_tiles * rows * row_floats;
  float* p; unsigned* q; (void)hipMalloc(&p, n * 4); (void)hipMalloc(&q, 4);
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  auto time = [&](const char* name, auto launch) {
    float sum = 0;
    for (int i = 0; i < 10; ++i) {
      (void)hipMemset(q, 0, 4);
      (void)hipEventRecord(a); launch(); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
      float ms; (void)hipEventElapsedTime(&ms, a, b);
      if (i >= 2) sum += ms;
    }
    printf("%-40s mean %.4f ms  %.0f GB/s\n", name, sum / 8, n * 4 / (sum / 8) / 1e6);
  };
  time("tile-major (GROUP 1)", [&] { hipLaunchKernelGGL(tiles<1>, dim3(512), dim3(256), 0, 0, p, rows, row_floats, n_tiles, q, 0); });
  time("rows of 8 tiles interleaved", [&] { hipLaunchKernelGGL(tiles<8>, dim3(512), dim3(256), 0, 0, p, rows, row_floats, n_tiles, q, 0); });
  time("rows of 64 tiles interleaved", [&] { hipLaunchKernelGGL(tiles<64>, dim3(512), dim3(256), 0, 0, p, rows, row_floats, n_tiles, q, 0); });
  time("rows of 512 tiles interleaved", [&] { hipLaunchKernelGGL(tiles<512>, dim3(512), dim3(256), 0, 0, p, rows, row_floats, n_tiles, q, 0); });
  time("8 interleaved, odd tiles late", [&] { hipLaunchKernelGGL(tiles<8>, dim3(512), dim3(256), 0, 0, p, rows, row_floats, n_tiles, q, 1); });
  time("512 interleaved, odd tiles late", [&] { hipLaunchKernelGGL(tiles<512>, dim3(512), dim3(256), 0, 0, p, rows, row_floats, n_tiles, q, 1); });
  return 0;
}
