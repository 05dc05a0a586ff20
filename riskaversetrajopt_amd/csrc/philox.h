// Philox4x32-10 counter-based generator (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3",
// SC'11; Random123's philox4x32 with 10 rounds) and the uniform / Box–Muller transforms of the device sampler
// (SURVEY 8f-4).  The reference draws with NumPy's MT19937 on the host (drone_utils.py:61-93, driving.py:84-120,
// hopper.py:70-74); for batches that never leave HBM a counter-based generator gives every (sample, step) its own
// independent stream with no state: the value at (seed, stream, step t, sample m) is a pure function of those four
// numbers, so it can be materialised once (rato_*_sample) or REGENERATED inside the rollout kernels
// (rato_*_eval_philox) with bit-identical results.
//
//   counter = (m & 0xffffffff, m >> 32, t, stream)      key = (seed & 0xffffffff, seed >> 32)
//   4 output words r0..r3 -> up to 4 values:
//     uniform   u_k = ((r_k >> 8) + 0.5) * 2^-24                       in (0, 1), 24 bits
//     normal    (n0, n1) = BoxMuller(r0, r1), (n2, n3) = BoxMuller(r2, r3):
//               rho = sqrt(-2 ln(((r_a >> 8) + 1) * 2^-24)),  phi = (r_b >> 8) * 2^-24 revolutions,
//               (rho cos 2 pi phi, rho sin 2 pi phi)      [v_log_f32, v_sqrt_f32, v_cos_f32 / v_sin_f32: the
//               hardware trig takes its argument in revolutions, so there is no range reduction]
// oracle/philox.py restates both in NumPy (integers bit for bit; the transforms in fp64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rato {

struct u32x4 {
  uint32_t x, y, z, w;
};

__host__ __device__ __forceinline__ u32x4 philox4x32_10(u32x4 c, uint32_t k0, uint32_t k1) {
  constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)M0 * c.x, p1 = (uint64_t)M1 * c.z;
    u32x4 n;
    n.x = (uint32_t)(p1 >> 32) ^ c.y ^ k0;
    n.y = (uint32_t)p1;
    n.z = (uint32_t)(p0 >> 32) ^ c.w ^ k1;
    n.w = (uint32_t)p0;
    c = n;
    k0 += W0;
    k1 += W1;
  }
  return c;
}

__device__ __forceinline__ u32x4 philox_at(uint64_t seed, uint32_t stream, uint32_t t, uint64_t m) {
  u32x4 c;
  c.x = (uint32_t)m;
  c.y = (uint32_t)(m >> 32);
  c.z = t;
  c.w = stream;
  return philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
}

__device__ __forceinline__ float u01(uint32_t r) { return ((float)(r >> 8) + 0.5f) * 5.9604644775390625e-8f; }

// one Box–Muller pair from two words
__device__ __forceinline__ void box_muller(uint32_t ra, uint32_t rb, float& n0, float& n1) {
  const float u = ((float)(ra >> 8) + 1.0f) * 5.9604644775390625e-8f;   // (0, 1]
  const float phi = (float)(rb >> 8) * 5.9604644775390625e-8f;          // [0, 1) revolutions
  // ln u = log2(u) * ln 2
  const float rho = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u));
  n0 = rho * __builtin_amdgcn_cosf(phi);
  n1 = rho * __builtin_amdgcn_sinf(phi);
}

// stream ids (counter word 3): one per sampled array, so that arrays never share counters
enum : uint32_t {
  PHILOX_STREAM_DW = 1,      // Brownian increments (drone: 3 velocity components; car: 2 pedestrian components)
  PHILOX_STREAM_MASS = 2,    // drone masses
  PHILOX_STREAM_RADII = 3,   // drone obstacle semi-axes (t = obstacle index, 3 components)
  PHILOX_STREAM_OMEGA = 4,   // car: (omega_speed, omega_repulsive)
  PHILOX_STREAM_X0 = 5,      // car: pedestrian initial state (4 normals)
  PHILOX_STREAM_FIELD = 6,   // hopper: t = feature index, components (intensity, theta, tau)
  PHILOX_STREAM_USER = 16    // rato_philox_* test / utility entry points add the caller's stream to this
};

}  // namespace rato
