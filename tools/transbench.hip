// Issue rate of the hardware transcendental unit (v_cos_f32 / v_sin_f32) on gfx950: the roofline of the
// hopper friction kernel.  Build: hipcc --offload-arch=gfx950 -O3 -o transbench tools/transbench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
  float x[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) x[i] = seed + 0.001f * (threadIdx.x + i);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (MODE == 0) x[i] = __builtin_amdgcn_cosf(x[i]);                       // 1 trans
      if (MODE == 1) x[i] = __builtin_amdgcn_cosf(x[i]) * 0.5f + 0.25f;        // 1 trans + 1 fma
      if (MODE == 2) x[i] = x[i] * 0.999f + 0.001f;                            // 1 fma
    }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += x[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
  float* out;
  const int blocks = 256 * 8, iters = 20000;   // 8 waves per SIMD
  hipMalloc(&out, blocks * 256 * sizeof(float));
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int mode = 0; mode < 3; ++mode) {
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(a);
      if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, out, iters, 0.3f);
      if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, out, iters, 0.3f);
      if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, out, iters, 0.3f);
      hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b);
      const double ops = (double)blocks * 256 * 8.0 * iters;
      if (rep) printf("mode %d (%s): %.3f ms  %.3e lane-ops/s  -> %.2f cycles per wave64 instruction per SIMD at 2.4 GHz\n",
                      mode, mode == 0 ? "cos" : mode == 1 ? "cos+fma" : "fma", ms, ops / (ms * 1e-3),
                      2.4e9 * 1024 * 64 / (ops / (ms * 1e-3)));
    }
  }
  return 0;
}
