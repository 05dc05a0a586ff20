// Ceiling for the EXACT write pattern + dispatch structure of drone_linearize_rows_kernel (no compute):
// grid = n_tiles workgroups of 8 waves, 66 KB dynamic LDS (2 workgroups per CU), LDS work queue of rows
// t = S-1..1, each row swept k = t..1 writing 6 x 256 B at the packed-pair offset.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)
__global__ __launch_bounds__(512) void pattern(float* G, int S) {
  extern __shared__ int lds[];
  const int lane = threadIdx.x & 63;
  int* head = lds;
  if (threadIdx.x == 0) *head = 0;
  __syncthreads();
  const size_t tile_floats = (size_t)(S * (S - 1) / 2) * 6 * 64;
  float* Gt = G + (size_t)blockIdx.x * tile_floats + lane;
  for (;;) {
    int task = 0;
    if (lane == 0) task = atomicAdd(head, 1);
    task = __builtin_amdgcn_readfirstlane(task);
    if (task >= S - 1) break;
    const int t = S - 1 - task;
    float* Grow = Gt + (size_t)(t * (t - 1) / 2) * (6 * 64);
    float v = (float)t;
    for (int k = t; k >= 1; --k) {
      float* o = Grow + (k - 1) * (6 * 64);
      v += 1.0f;
#pragma unroll
      for (int r = 0; r < 6; ++r) o[r * 64] = v + r;
    }
  }
}
int main() {
  const int S = 50;
  for (int M : {100000, 1000000}) {
    const int n_tiles = (M + 63) / 64;
    const size_t bytes = (size_t)n_tiles * (S * (S - 1) / 2) * 6 * 64 * 4;
    float* G; CK(hipMalloc(&G, bytes));
    CK(hipFuncSetAttribute((const void*)pattern, hipFuncAttributeMaxDynamicSharedMemorySize, 66 * 1024));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) pattern<<<n_tiles, 512, 66 * 1024>>>(G, S);
    CK(hipEventRecord(a));
    for (int i = 0; i < 10; ++i) pattern<<<n_tiles, 512, 66 * 1024>>>(G, S);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 10;
    printf("M=%d: %.3f GB in %.4f ms -> %.0f GB/s (pattern + dispatch ceiling)\n", M, bytes / 1e9, ms, bytes / ms / 1e6);
    CK(hipFree(G));
  }
  return 0;
}
