// Device side of the sparse QP assembly: emits the obstacle/separation block of the CSC value array
// ("Ax" of osqp.update(Ax=...), drone_risk.py:451) straight from the packed causal Jacobian.
//
// In the reference's matrix (drone_risk.py:282-423) the u-column (s, g) holds, after its
// final-constraint entries, one entry per (sample i, row-group r, t > s), sorted by row, i.e. in the order
// [i][r][t]; consecutive samples are adjacent, so a 64-sample chunk of one column is ONE contiguous run of
// 64 * R * (S-1-s) values.  The packed Jacobian stores the same numbers as [tile][pair(t,s)][g][r][lane].
// This kernel is the transposition between the two, staged through LDS so that both the reads (256 B per
// wave instruction) and the writes (linear) are coalesced.  Bound: HBM, one read + one write of the G data.
#include <atomic>

#include "rato_common.h"

namespace {

__global__ __launch_bounds__(RATO_BLOCK) void emit_csc_kernel(const float* __restrict__ G,
                                                              const float* __restrict__ W, long ld, int tileW, int n_g,
                                                              int R, int S, long M, float scale,
                                                              float* __restrict__ out) {
  extern __shared__ float lds[];
  const int s = blockIdx.y;                 // control step of this column pair
  const int nt = S - 1 - s;                 // rows t = s+1 .. S-1
  const int L = R * nt;                     // values per sample in this column
  const long i0 = (long)blockIdx.x * 64;    // first sample of the chunk
  const int nvalid = (int)((M - i0) < 64 ? (M - i0) : 64);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t n_pairs = (size_t)S * (S - 1) / 2;
  const bool fact = (W != nullptr);         // factored Jacobian: entry = W[r,t,g] * Phi[t,s,g]
  const int RR = fact ? 1 : R;
  const size_t tile_floats = rato::packed_tile_stride(n_pairs * n_g * RR * tileW);
  const float* __restrict__ Gt = G + (size_t)(i0 / tileW) * tile_floats + (i0 % tileW) + lane;
  const long before = (long)s * (S - 1) - (long)s * (s - 1) / 2;      // sum_{s'<s} (S-1-s')
  for (int g = 0; g < n_g; ++g) {
    for (int row = wave; row < L; row += RATO_BLOCK / 64) {
      const int r = row / nt, dtt = row - r * nt;                      // output index inside a sample: r*nt + (t-s-1)
      const int t = s + 1 + dtt;
      const size_t pair = (size_t)rato::pair_row_offset(t) + s;
      float v = 0.0f;
      if (lane < nvalid) {
        v = Gt[((pair * n_g + g) * RR + (fact ? 0 : r)) * tileW];
        if (fact) v *= W[(((size_t)r * S + t) * n_g + g) * ld + i0 + lane];
      }
      lds[lane * (L + 1) + row] = v * scale;
    }
    __syncthreads();
    float* __restrict__ o = out + ((long)M * R) * ((long)n_g * before + (long)g * nt) + i0 * L;
    const int total = nvalid * L;
    for (int q = threadIdx.x; q < total; q += RATO_BLOCK) {
      const int i = q / L, idx = q - i * L;
      o[q] = lds[i * (L + 1) + idx];
    }
    __syncthreads();
  }
}

}  // namespace

extern "C" int rato_emit_csc_values(const float* G, const float* W, int64_t ld, int32_t tile, int32_t n_g, int32_t R,
                                    int32_t S, int64_t M, float scale, float* out, void* stream) {
  RATO_CLEAR_ERROR();
  if (!G || !out || M <= 0 || S < 2 || n_g <= 0 || R <= 0 || (tile != 64 && tile != 256)) return RATO_EINVAL;
  const size_t lds = (size_t)64 * ((size_t)R * (S - 1) + 1) * sizeof(float);
  if (lds > 160 * 1024) return RATO_EINVAL;  // S <= 213 (drone, R = 3) / 639 (driving, R = 1): beyond, assemble on the host
  static rato::DynamicLdsLimit lds_limit;   // per device
  {
    const hipError_t e = lds_limit.ensure(lds, [](size_t) {
      return hipFuncSetAttribute(reinterpret_cast<const void*>(emit_csc_kernel),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    });
    if (e != hipSuccess) return RATO_EHIP - (int)e;
  }
  dim3 grid((unsigned)((M + 63) / 64), (unsigned)(S - 1)), block(RATO_BLOCK);
  hipLaunchKernelGGL(emit_csc_kernel, grid, block, lds, rato::as_stream(stream), G, W, (long)ld, tile, n_g, R, S,
                     (long)M, scale, out);
  RATO_LAUNCH_CHECK();
  return RATO_OK;
}
