// Standalone timing of rato_risk_stats through the C ABI (no Python): eager launches, HIP events; run under
// `rocprofv3 --kernel-trace --stats` for per-kernel durations.  build: see tools/README.md
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "rato_saa.h"
__global__ void empty_kernel(double* out) { if (threadIdx.x == 0 && out == nullptr) printf("x"); }
int main(int argc, char** argv) {
  const long M = argc > 1 ? atol(argv[1]) : 10000;
  const int clustered = argc > 2 ? atoi(argv[2]) : 1;
  std::vector<float> h(M);
  srand(1);
  for (long i = 0; i < M; ++i) {
    const float u = rand() / (float)RAND_MAX, v = rand() / (float)RAND_MAX;
    h[i] = clustered ? 0.9f + 0.05f * (u + v - 1.0f) : (u - 0.5f) * expf(8.0f * v);
  }
  float* Z; double* out; void* ws;
  const size_t wsb = rato_risk_stats_workspace_bytes(M);
  hipMalloc(&Z, M * 4); hipMalloc(&out, 16 * 8); hipMalloc(&ws, wsb);
  hipMemcpy(Z, h.data(), M * 4, hipMemcpyHostToDevice);
  hipStream_t st; hipStreamCreate(&st);
  rato_risk_stats_init(ws, wsb, st);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 10; ++i) rato_risk_stats(Z, M, 0.1, 1e-6f, ws, wsb, out, st);
  hipStreamSynchronize(st);
  const int N = 200;
  hipEventRecord(a, st);
  for (int i = 0; i < N; ++i) rato_risk_stats(Z, M, 0.1, 1e-6f, ws, wsb, out, st);
  hipEventRecord(b, st); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  double o[11]; hipMemcpy(o, out, sizeof(o), hipMemcpyDeviceToHost);
  printf("M=%ld %s: %.2f us/call (eager, back to back)  VaR=%.7g CVaR=%.7g\n", M, clustered ? "clustered" : "spread", ms * 1000 / N, o[0], o[1]);
  hipEventRecord(a, st);
  for (int i = 0; i < N; ++i) hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(1024), 0, st, out);
  hipEventRecord(b, st); hipEventSynchronize(b);
  hipEventElapsedTime(&ms, a, b);
  printf("empty 1024-thread kernel: %.2f us/launch\n", ms * 1000 / N);
  return 0;
}
