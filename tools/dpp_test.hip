#include <hip/hip_runtime.h>
#include <cstdio>
#include "rato_common.h"
__global__ void k(const float* in, float* out) {
  float v = in[threadIdx.x];
  out[threadIdx.x] = rato::wave_sum_dpp(v);
  out[64 + threadIdx.x] = rato::wave_sum(v);
}
int main() {
  float h[64], *d, *o, r[128];
  double ref = 0;
  for (int i = 0; i < 64; ++i) { h[i] = (float)(i * 1.37 - 20.1); ref += h[i]; }
  hipMalloc(&d, 256); hipMalloc(&o, 512);
  hipMemcpy(d, h, 256, hipMemcpyHostToDevice);
  k<<<1, 64>>>(d, o);
  hipMemcpy(r, o, 512, hipMemcpyDeviceToHost);
  printf("ref %.6f dpp lane0 %.6f lane17 %.6f lane63 %.6f shfl %.6f\n", ref, r[0], r[17], r[63], r[64]);
  return 0;
}
