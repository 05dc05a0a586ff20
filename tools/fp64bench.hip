// Issue rate of the fp64 vector instructions the cut-oracle kernels are made of (v_fma_f64, v_mul_f64, v_add_f64, a 64-bit
// select = 2 v_cndmask, v_cmp_gt_f64, v_cvt_f64_f32) on gfx950, against the waves per SIMD: the roofline behind
// `scp.kernels` of the bench line.  Build: hipcc --offload-arch=gfx950 -O3 -o /tmp/fp64bench tools/fp64bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ __launch_bounds__(256) void k(double* out, int iters, double seed) {
  double x[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) x[i] = seed + 0.001 * (threadIdx.x + i);
  const double a = 0.999, b = 0.001;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (MODE == 0) x[i] = __builtin_fma(x[i], a, b);                                   // v_fma_f64, 8 independent chains
      if (MODE == 1) x[i] = x[i] * a;                                                     // v_mul_f64
      if (MODE == 2) x[i] = x[i] + b;                                                     // v_add_f64
      if (MODE == 3) x[i] = (x[i] > 0.5) ? x[i] * a : x[i] + b;                           // cmp + mul + add + 2 cndmask
      if (MODE == 4) x[i] = __builtin_fma((double)(float)x[i], a, b);                     // cvt_f32_f64 + cvt_f64_f32 + fma
    }
    if (MODE == 5) {                                                                       // ONE dependent chain of fmas
#pragma unroll
      for (int i = 0; i < 8; ++i) x[0] = __builtin_fma(x[0], a, b);
    }
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += x[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE>
void run(double* out, int waves_per_simd, const char* what, double instr_per_elem) {
  const int blocks = 256 * waves_per_simd, iters = 4000;
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  float ms = 0;
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(a);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, 0.3);
    hipEventRecord(b); hipEventSynchronize(b);
    hipEventElapsedTime(&ms, a, b);
  }
  const double wave_instr = (double)blocks * 4 * 8.0 * iters * instr_per_elem;     // per launch, all SIMDs
  printf("%-44s %d wave(s) per SIMD: %.3f ms -> %.2f cycles per wave64 instruction per SIMD at 2.4 GHz\n", what, waves_per_simd, ms,
         ms * 1e-3 * 2.4e9 * 1024 / wave_instr);
}
int main() {
  double* out;
  hipMalloc(&out, 256 * 8 * 256 * sizeof(double));
  for (int w : {1, 2, 4, 8}) {
    run<0>(out, w, "v_fma_f64 (8 independent chains)", 1);
    run<1>(out, w, "v_mul_f64", 1);
    run<2>(out, w, "v_add_f64", 1);
    run<3>(out, w, "cmp + mul + add + 64-bit select (5 instr)", 5);
    run<4>(out, w, "cvt f64->f32->f64 + fma (3 instr)", 3);
    run<5>(out, w, "v_fma_f64, ONE dependent chain", 1);
  }
  return 0;
}
