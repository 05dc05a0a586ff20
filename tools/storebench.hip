// Calibration: achievable HBM write bandwidth for "many concurrent streams" patterns.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)

// P0: linear fill, float4 per lane, grid-stride
__global__ void fill4(float4* p, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x)
    p[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}
// P1: each WAVE owns a contiguous region of region_floats; per iteration it writes ROWS rows of 64 floats
// (ROWS dword stores of 256 B each => ROWS*256 B contiguous), ascending or descending.
template <int ROWS, bool DESC>
__global__ void streams_dword(float* p, size_t region_floats, int iters) {
  const int lane = threadIdx.x & 63;
  const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  float* base = p + wave * region_floats + lane;
  for (int it = 0; it < iters; ++it) {
    const int k = DESC ? (iters - 1 - it) : it;
    float* o = base + (size_t)k * ROWS * 64;
    const float v = (float)it;
#pragma unroll
    for (int r = 0; r < ROWS; ++r) o[r * 64] = v + r;
  }
}
// P2: same but float4 per lane: each store = 1 KB; ROWS stores per iteration
template <int ROWS>
__global__ void streams_x4(float4* p, size_t region_f4, int iters) {
  const int lane = threadIdx.x & 63;
  const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  float4* base = p + wave * region_f4 + lane;
  for (int it = 0; it < iters; ++it) {
    float4* o = base + (size_t)it * ROWS * 64;
    const float v = (float)it;
#pragma unroll
    for (int r = 0; r < ROWS; ++r) o[r * 64] = make_float4(v, v + r, v, v);
  }
}
// P3: like the column kernel: a block of 256 threads writes, per iteration, NCOL groups of 6 rows x 1 KB,
// groups strided by `gstride` rows (different columns), rows advancing with the iteration.
template <typename F>
float timeit(F f, int n = 10) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 2; ++i) f();
  CK(hipEventRecord(a));
  for (int i = 0; i < n; ++i) f();
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / n;
}
int main() {
  const size_t bytes = (size_t)3 << 30;  // 3 GiB
  float* p; CK(hipMalloc(&p, bytes));
  CK(hipMemset(p, 0, bytes));
  {
    float ms = timeit([&] { fill4<<<256 * 8, 256>>>((float4*)p, bytes / 16); });
    printf("P0 linear fill x4           : %.3f ms  %.0f GB/s\n", ms, bytes / ms / 1e6);
  }
  for (int nblk : {512, 1024, 2048}) {
    const int waves = nblk * 8;
    const size_t region_floats = bytes / 4 / waves;
    {
      const int iters = region_floats / (6 * 64);
      float ms = timeit([&] { streams_dword<6, true><<<nblk, 512>>>(p, region_floats, iters); });
      printf("P1 %4d blk x8 waves, 6x256B desc : %.3f ms  %.0f GB/s\n", nblk, ms, (double)waves * iters * 6 * 256 / ms / 1e6);
      ms = timeit([&] { streams_dword<6, false><<<nblk, 512>>>(p, region_floats, iters); });
      printf("P1 %4d blk x8 waves, 6x256B asc  : %.3f ms  %.0f GB/s\n", nblk, ms, (double)waves * iters * 6 * 256 / ms / 1e6);
    }
    {
      const int iters = region_floats / (24 * 64);
      float ms = timeit([&] { streams_dword<24, false><<<nblk, 512>>>(p, region_floats, iters); });
      printf("P1 %4d blk x8 waves, 24x256B asc : %.3f ms  %.0f GB/s\n", nblk, ms, (double)waves * iters * 24 * 256 / ms / 1e6);
    }
    {
      const size_t region_f4 = bytes / 16 / waves;
      const int iters = region_f4 / (6 * 64);
      float ms = timeit([&] { streams_x4<6><<<nblk, 512>>>((float4*)p, region_f4, iters); });
      printf("P2 %4d blk x8 waves, 6x1KB asc   : %.3f ms  %.0f GB/s\n", nblk, ms, (double)waves * iters * 6 * 1024 / ms / 1e6);
    }
  }
  for (int nblk : {256, 512, 1024}) {  // 4 waves per block
    const int waves = nblk * 4;
    const size_t region_f4 = bytes / 16 / waves;
    const int iters = region_f4 / (6 * 64);
    float ms = timeit([&] { streams_x4<6><<<nblk, 256>>>((float4*)p, region_f4, iters); });
    printf("P2 %4d blk x4 waves, 6x1KB asc   : %.3f ms  %.0f GB/s\n", nblk, ms, (double)waves * iters * 6 * 1024 / ms / 1e6);
  }
  return 0;
}
