// EXPERIMENT: the "address-ordered expansion" form of the drone products Jacobian.
// Kernel 1 of that design leaves the per-sample step-Jacobian table in HBM (here: rato_drone_linearize_generators'
// A22 / W, which the library already writes); THIS kernel is kernel 2: short-lived workgroups, each wave writing one
// contiguous chunk of CH pair-steps (CH * 1536 B) of the packed products Jacobian, chunks handed out in ADDRESS ORDER
// (blockIdx -> consecutive addresses), so that at any moment the chip writes one compact window of the buffer instead
// of 512 tiles 2 MiB apart.  A chunk that starts inside row t re-runs the adjoint from t down to its first column
// without storing (the price of not keeping the tables in LDS).
// Built and timed by tools/expand_proto.py; not part of librato_saa.so.
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {
constexpr int NOBS = 3;
__host__ __device__ inline int row_off(int t) { return (t * (t - 1)) >> 1; }

template <int NW>
__global__ __launch_bounds__(NW * 64) void expand_kernel(const float* __restrict__ A22, int a22_axes,
                                                        const float* __restrict__ W, const float* __restrict__ mass,
                                                        float* __restrict__ G, long M, long ld, int S, float dt, float kp,
                                                        int CH, int wg_per_tile, size_t tile_stride) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int tile = blockIdx.x / wg_per_tile;
  const int c = (blockIdx.x - tile * wg_per_tile) * NW + wave;
  const int npairs = row_off(S);
  const int p0 = c * CH;
  if (p0 >= npairs) return;
  const int p1 = (p0 + CH < npairs) ? p0 + CH : npairs;
  const long m_raw = (long)tile * 64 + lane;
  const bool valid = m_raw < M;
  const long m = valid ? m_raw : M - 1;
  const float inv_m = 1.0f / mass[m];
  const float a21 = -kp * dt * inv_m, dtm = dt * inv_m;
  float* __restrict__ Gt = G + (size_t)tile * tile_stride + lane;
  int t = (int)((1.0f + sqrtf(1.0f + 8.0f * (float)p0)) * 0.5f);
  while (row_off(t) > p0) --t;
  while (row_off(t + 1) <= p0) ++t;
  for (; t < S && row_off(t) < p1; ++t) {
    const int off = row_off(t);
    const int s_lo = (p0 > off) ? p0 - off : 0;
    const int s_hi = ((p1 - off < t) ? p1 - off : t) - 1;
    float wx[NOBS], wy[NOBS];
#pragma unroll
    for (int j = 0; j < NOBS; ++j) {
      wx[j] = W[(((size_t)j * S + t) * 2 + 0) * ld + m] * dtm;
      wy[j] = W[(((size_t)j * S + t) * 2 + 1) * ld + m] * dtm;
    }
    float m0x = 1.0f, m0y = 1.0f, m1x = 0.0f, m1y = 0.0f;
    float* __restrict__ Grow = Gt + (size_t)off * (6 * 64);
    constexpr int KB = 8;   // table loads of 8 steps in flight before their dependent recursion (as in cvar.hip)
    for (int kb = t; kb >= s_lo + 1; kb -= KB) {
      float ax[KB], ay[KB];
#pragma unroll
      for (int i = 0; i < KB; ++i) {
        const int k = (kb - i >= s_lo + 1) ? kb - i : s_lo + 1;
        ax[i] = A22[((size_t)k * a22_axes + 0) * ld + m];
        ay[i] = A22[((size_t)k * a22_axes + 1) * ld + m];
      }
#pragma unroll
      for (int i = 0; i < KB; ++i) {
        const int k = kb - i;
        if (k >= s_lo + 1) {
          const float n0x = m0x + m1x * a21, n0y = m0y + m1y * a21;
          const float n1x = m0x * dt + m1x * ax[i], n1y = m0y * dt + m1y * ay[i];
          m0x = n0x; m0y = n0y; m1x = n1x; m1y = n1y;
          if (k - 1 <= s_hi && valid) {
            float* __restrict__ o = Grow + (size_t)(k - 1) * (6 * 64);
#pragma unroll
            for (int j = 0; j < NOBS; ++j) {
              o[j * 64] = wx[j] * m1x;
              o[(NOBS + j) * 64] = wy[j] * m1y;
            }
          }
        }
      }
    }
  }
}
}  // namespace

extern "C" int expand_proto(const float* A22, int a22_axes, const float* W, const float* mass, float* G, long M, long ld,
                            int S, float dt, float kp, int CH, int NW, size_t tile_stride, void* stream) {
  const int npairs = row_off(S);
  const int chunks = (npairs + CH - 1) / CH;
  const int wg_per_tile = (chunks + NW - 1) / NW;
  const long ntiles = (M + 63) / 64;
  dim3 grid((unsigned)(ntiles * wg_per_tile));
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  switch (NW) {
    case 1: hipLaunchKernelGGL(expand_kernel<1>, grid, dim3(64), 0, st, A22, a22_axes, W, mass, G, M, ld, S, dt, kp, CH, wg_per_tile, tile_stride); break;
    case 2: hipLaunchKernelGGL(expand_kernel<2>, grid, dim3(128), 0, st, A22, a22_axes, W, mass, G, M, ld, S, dt, kp, CH, wg_per_tile, tile_stride); break;
    case 4: hipLaunchKernelGGL(expand_kernel<4>, grid, dim3(256), 0, st, A22, a22_axes, W, mass, G, M, ld, S, dt, kp, CH, wg_per_tile, tile_stride); break;
    case 8: hipLaunchKernelGGL(expand_kernel<8>, grid, dim3(512), 0, st, A22, a22_axes, W, mass, G, M, ld, S, dt, kp, CH, wg_per_tile, tile_stride); break;
    default: return -1;
  }
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
