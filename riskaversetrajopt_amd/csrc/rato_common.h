// Shared device/host helpers for librato_saa.so (gfx950 only: wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rato_saa.h"

#define RATO_BLOCK 256           // 4 waves per workgroup (== RATO_TILE of rato_saa.h)
#define RATO_WAVE 64

// hipGetLastError() is sticky per host thread: clear anything left behind by an
// earlier, unrelated runtime call before launching, then check our own launches.
#define RATO_CLEAR_ERROR() (void)hipGetLastError()

#define RATO_LAUNCH_CHECK()                              \
  do {                                                   \
    hipError_t e__ = hipGetLastError();                  \
    if (e__ != hipSuccess) return RATO_EHIP - (int)e__;  \
  } while (0)

static_assert(RATO_BLOCK == RATO_TILE, "one workgroup writes one Jacobian tile");

namespace rato {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, RATO_WAVE);
  return v;
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, RATO_WAVE);
  return v;
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, RATO_WAVE));
  return v;
}

// pair(t, s) = t(t-1)/2 + s for 0 <= s < t  (row-major causal packing)
__host__ __device__ __forceinline__ int pair_row_offset(int t) { return (t * (t - 1)) >> 1; }

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

static inline int nblocks_for(int32_t M) { return (M + RATO_BLOCK - 1) / RATO_BLOCK; }

}  // namespace rato
