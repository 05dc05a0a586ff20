// Store-bandwidth ceilings for the access pattern of the packed Jacobian (tools/README.md):
//   linear : grid-stride fill, consecutive workgroups write consecutive addresses
//   tiles  : W resident workgroups, each writes whole tiles (TILE_BYTES contiguous) row by row, 256 B (dword per lane)
//            or 1 KB (dwordx4 per lane) per wave instruction, tiles taken from an atomic queue
// build: hipcc --offload-arch=gfx950 -O3 tools/store_pattern.hip -o tools/_build/store_pattern
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
__global__ void linear_fill(float4* p, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x)
    p[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}
__global__ void linear_fill_nt(float4* p, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    float* q = reinterpret_cast<float*>(p + i);
    __builtin_nontemporal_store(1.f, q); __builtin_nontemporal_store(2.f, q + 1);
    __builtin_nontemporal_store(3.f, q + 2); __builtin_nontemporal_store(4.f, q + 3);
  }
}
__global__ void oneshot_fill(float4* p, size_t n4) {   // every workgroup writes one contiguous 16 KB chunk and exits
  const size_t i0 = (size_t)blockIdx.x * 1024;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const size_t i = i0 + u * 256 + threadIdx.x;
    if (i < n4) p[i] = make_float4(1.f, 2.f, 3.f, 4.f);
  }
}
__global__ void oneshot_fill_dword(float* p, size_t n) {   // one contiguous 16 KB chunk, dword per lane
  const size_t i0 = (size_t)blockIdx.x * 4096;
#pragma unroll
  for (int u = 0; u < 16; ++u) {
    const size_t i = i0 + u * 256 + threadIdx.x;
    if (i < n) p[i] = 1.f;
  }
}
template <int VEC, bool NT = false>
__global__ __launch_bounds__(256) void tile_fill(float* p, size_t tile_floats, int n_tiles, unsigned* queue) {
  __shared__ int tile_s;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (;;) {
    __syncthreads();
    if (threadIdx.x == 0) tile_s = (int)atomicAdd(queue, 1u);
    __syncthreads();
    const int tile = tile_s;
    if (tile >= n_tiles) break;
    float* base = p + (size_t)tile * tile_floats;
    const size_t rows = tile_floats / (64 * VEC);
    for (size_t r = wave; r < rows; r += 4) {
      if (VEC == 1 && NT) __builtin_nontemporal_store((float)r, base + r * 64 + lane);
      else if (VEC == 1) base[r * 64 + lane] = (float)r;
      else reinterpret_cast<float4*>(base)[r * 64 + lane] = make_float4((float)r, 1.f, 2.f, 3.f);
    }
  }
}
int main(int argc, char** argv) {
  const int n_tiles = 1563;
  const size_t tile_floats = (size_t)3 * 50 * 49 * 64 + 64 * 64;   // ~ 1.9 MB per tile
  const size_t n = tile_floats * n_tiles;
  float* p; unsigned* q;
  hipMalloc(&p, n * 4); hipMalloc(&q, 4);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  auto time = [&](const char* name, auto launch) {
    float best = 1e9, sum = 0;
    for (int i = 0; i < 12; ++i) {
      hipMemset(q, 0, 4);
      hipEventRecord(a); launch(); hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b);
      if (i >= 2) { sum += ms; if (ms < best) best = ms; }
    }
    printf("%-44s mean %.4f ms  best %.4f ms  %.0f GB/s\n", name, sum / 10, best, n * 4 / (sum / 10) / 1e6);
  };
  time("linear float4, 2048 x 256", [&] { hipLaunchKernelGGL(linear_fill, dim3(2048), dim3(256), 0, 0, (float4*)p, n / 4); });
  time("linear float4, 16384 x 256", [&] { hipLaunchKernelGGL(linear_fill, dim3(16384), dim3(256), 0, 0, (float4*)p, n / 4); });
  time("linear float4 nontemporal, 16384 x 256", [&] { hipLaunchKernelGGL(linear_fill_nt, dim3(16384), dim3(256), 0, 0, (float4*)p, n / 4); });
  time("one-shot 16 KB per workgroup, float4", [&] { hipLaunchKernelGGL(oneshot_fill, dim3((unsigned)((n / 4 + 1023) / 1024)), dim3(256), 0, 0, (float4*)p, n / 4); });
  time("one-shot 16 KB per workgroup, dword", [&] { hipLaunchKernelGGL(oneshot_fill_dword, dim3((unsigned)((n + 4095) / 4096)), dim3(256), 0, 0, p, n); });
  time("hipMemsetD32Async", [&] { hipMemsetD32Async((hipDeviceptr_t)p, 0x3f800000, n, 0); });
  for (int W : {512}) {
    char nm[96];
    snprintf(nm, sizeof nm, "tiles, %d workgroups, 256 B nontemporal", W);
    time(nm, [&] { hipLaunchKernelGGL((tile_fill<1, true>), dim3(W), dim3(256), 0, 0, p, tile_floats, n_tiles, q); });
  }
  for (int W : {256, 512, 1024}) {
    char nm[96];
    snprintf(nm, sizeof nm, "tiles, %d workgroups, 256 B per wave store", W);
    time(nm, [&] { hipLaunchKernelGGL(tile_fill<1>, dim3(W), dim3(256), 0, 0, p, tile_floats, n_tiles, q); });
    snprintf(nm, sizeof nm, "tiles, %d workgroups, 1 KB per wave store", W);
    time(nm, [&] { hipLaunchKernelGGL(tile_fill<4>, dim3(W), dim3(256), 0, 0, p, tile_floats, n_tiles, q); });
  }
  return 0;
}
