// Device-side sampler: the reference's distributions drawn in HBM, in the kernels' SoA layouts, with
// Philox4x32-10 (philox.h).  Replaces — for synthetic / Monte-Carlo batches that never leave the device — the host
// loops of drone_utils.py:61-93 (masses ~ U(29,35); obstacle semi-axes r_j + U(-.025,.025) per dimension;
// DWs = sqrt(dt) N(0,1)), driving.py:84-120 (omega_speed ~ U(.025,.175), omega_repulsive ~ U(.005,.095); pedestrian
// initial state + diag(1e-1,1e-1,1e-4,1e-4) N(0,I); DWs) and hopper.py:70-74 (a ~ sqrt(2/30) 0.025 U(0,1),
// theta ~ U(0,pi), tau ~ U(0,2pi)).  These are NOT the reference's MT19937 draws (identical draws: the host samplers
// of drone_utils.py / driving.py / hopper.py here replay its stream); they are the same distributions.
// One lane = one sample; every store is 256 contiguous bytes per wave.
#include "philox.h"
#include "rato_common.h"

namespace {

using rato::u32x4;

// generic [T][C][ld] fills (tests, utilities): C <= 4 components per (t, m), counter = (m, t, USER + stream)
__global__ __launch_bounds__(RATO_BLOCK) void philox_u32_kernel(uint32_t* __restrict__ out, int T, long M, long ld,
                                                               uint64_t seed, uint32_t stream) {
  const long m = (long)blockIdx.x * RATO_BLOCK + threadIdx.x;
  const int t = blockIdx.y;
  if (m >= M || t >= T) return;
  const u32x4 r = rato::philox_at(seed, stream, (uint32_t)t, (uint64_t)m);
  uint32_t* o = out + (size_t)t * 4 * ld + m;
  o[0] = r.x; o[ld] = r.y; o[2 * ld] = r.z; o[3 * ld] = r.w;
}

struct Affine4 {
  float a[4], b[4];   // value_k = a[k] * draw_k + b[k]
};

template <bool NORMAL>
__global__ __launch_bounds__(RATO_BLOCK) void philox_fill_kernel(float* __restrict__ out, int T, int C, long M, long ld,
                                                                uint64_t seed, uint32_t stream, Affine4 f) {
  const long m = (long)blockIdx.x * RATO_BLOCK + threadIdx.x;
  const int t = blockIdx.y;
  if (m >= M || t >= T) return;
  const u32x4 r = rato::philox_at(seed, stream, (uint32_t)t, (uint64_t)m);
  float v[4];
  if (NORMAL) {
    rato::box_muller(r.x, r.y, v[0], v[1]);
    rato::box_muller(r.z, r.w, v[2], v[3]);
  } else {
    v[0] = rato::u01(r.x); v[1] = rato::u01(r.y); v[2] = rato::u01(r.z); v[3] = rato::u01(r.w);
  }
  float* o = out + (size_t)t * C * ld + m;
#pragma unroll
  for (int k = 0; k < 4; ++k)
    if (k < C) o[(size_t)k * ld] = f.a[k] * v[k] + f.b[k];
}

// drone: mass [ld], Qsym [3 obs][3][ld] = (1/rx^2, 0, 1/ry^2) (obs_Qs is diagonal: drone_utils.py:69-76)
__global__ __launch_bounds__(RATO_BLOCK) void drone_params_sample_kernel(long M, long ld, uint64_t seed,
                                                                        float mass_nom, float mass_delta,
                                                                        float r0, float r1, float r2, float r_delta,
                                                                        float* __restrict__ mass,
                                                                        float* __restrict__ Qsym) {
  const long m = (long)blockIdx.x * RATO_BLOCK + threadIdx.x;
  if (m >= M) return;
  const u32x4 rm = rato::philox_at(seed, rato::PHILOX_STREAM_MASS, 0, (uint64_t)m);
  mass[m] = mass_nom + mass_delta * (2.0f * rato::u01(rm.x) - 1.0f);
  const float radii[3] = {r0, r1, r2};
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const u32x4 r = rato::philox_at(seed, rato::PHILOX_STREAM_RADII, (uint32_t)j, (uint64_t)m);
    const float lx = radii[j] + r_delta * (2.0f * rato::u01(r.x) - 1.0f);
    const float ly = radii[j] + r_delta * (2.0f * rato::u01(r.y) - 1.0f);   // (the z semi-axis, r.z, never enters [:2,:2])
    Qsym[(size_t)(j * 3 + 0) * ld + m] = 1.0f / (lx * lx);
    Qsym[(size_t)(j * 3 + 1) * ld + m] = 0.0f;
    Qsym[(size_t)(j * 3 + 2) * ld + m] = 1.0f / (ly * ly);
  }
}

// car: w_speed, w_rep [M]; x0_ped [4][M]
__global__ __launch_bounds__(RATO_BLOCK) void car_params_sample_kernel(long M, uint64_t seed, float ws_nom, float ws_del,
                                                                      float wr_nom, float wr_del, Affine4 x0,
                                                                      float* __restrict__ w_speed,
                                                                      float* __restrict__ w_rep,
                                                                      float* __restrict__ x0_ped) {
  const long m = (long)blockIdx.x * RATO_BLOCK + threadIdx.x;
  if (m >= M) return;
  const u32x4 ro = rato::philox_at(seed, rato::PHILOX_STREAM_OMEGA, 0, (uint64_t)m);
  w_speed[m] = ws_nom + ws_del * (2.0f * rato::u01(ro.x) - 1.0f);
  w_rep[m] = wr_nom + wr_del * (2.0f * rato::u01(ro.y) - 1.0f);
  const u32x4 rx = rato::philox_at(seed, rato::PHILOX_STREAM_X0, 0, (uint64_t)m);
  float n[4];
  rato::box_muller(rx.x, rx.y, n[0], n[1]);
  rato::box_muller(rx.z, rx.w, n[2], n[3]);
#pragma unroll
  for (int k = 0; k < 4; ++k) x0_ped[(size_t)k * M + m] = x0.b[k] + x0.a[k] * n[k];
}

// hopper: a, theta, tau [30][M]
__global__ __launch_bounds__(RATO_BLOCK) void hopper_fields_sample_kernel(long M, uint64_t seed, float a_scale,
                                                                         float* __restrict__ a,
                                                                         float* __restrict__ theta,
                                                                         float* __restrict__ tau) {
  const long m = (long)blockIdx.x * RATO_BLOCK + threadIdx.x;
  const int k = blockIdx.y;
  if (m >= M) return;
  const u32x4 r = rato::philox_at(seed, rato::PHILOX_STREAM_FIELD, (uint32_t)k, (uint64_t)m);
  a[(size_t)k * M + m] = a_scale * rato::u01(r.x);
  theta[(size_t)k * M + m] = 3.14159265358979323846f * rato::u01(r.y);
  tau[(size_t)k * M + m] = 6.28318530717958647692f * rato::u01(r.z);
}

inline unsigned nblk(long M) { return (unsigned)((M + RATO_BLOCK - 1) / RATO_BLOCK); }

}  // namespace

extern "C" int rato_philox_u32(uint32_t* out, int32_t T, int64_t M, int64_t ld, uint64_t seed, uint32_t stream_id,
                               void* stream) {
  RATO_CLEAR_ERROR();
  if (!out || T <= 0 || T > 65535 || M <= 0 || ld < M) return RATO_EINVAL;
  hipLaunchKernelGGL(philox_u32_kernel, dim3(nblk(M), T), dim3(RATO_BLOCK), 0, rato::as_stream(stream), out, T, (long)M,
                     (long)ld, seed, rato::PHILOX_STREAM_USER + stream_id);
  RATO_LAUNCH_CHECK();
  return RATO_OK;
}

static int fill(bool normal, float* out, int32_t T, int32_t C, int64_t M, int64_t ld, uint64_t seed,
                uint32_t stream_id, const float* a, const float* b, void* stream) {
  RATO_CLEAR_ERROR();
  if (!out || T <= 0 || T > 65535 || C < 1 || C > 4 || M <= 0 || ld < M) return RATO_EINVAL;
  Affine4 f;
  for (int k = 0; k < 4; ++k) {
    f.a[k] = a ? a[k < C ? k : 0] : 1.0f;
    f.b[k] = b ? b[k < C ? k : 0] : 0.0f;
  }
  if (normal)
    hipLaunchKernelGGL(philox_fill_kernel<true>, dim3(nblk(M), T), dim3(RATO_BLOCK), 0, rato::as_stream(stream), out, T,
                       C, (long)M, (long)ld, seed, stream_id, f);
  else
    hipLaunchKernelGGL(philox_fill_kernel<false>, dim3(nblk(M), T), dim3(RATO_BLOCK), 0, rato::as_stream(stream), out,
                       T, C, (long)M, (long)ld, seed, stream_id, f);
  RATO_LAUNCH_CHECK();
  return RATO_OK;
}

extern "C" int rato_philox_normal(float* out, int32_t T, int32_t C, int64_t M, int64_t ld, uint64_t seed,
                                  uint32_t stream_id, const float* scale, const float* mean, void* stream) {
  return fill(true, out, T, C, M, ld, seed, rato::PHILOX_STREAM_USER + stream_id, scale, mean, stream);
}

extern "C" int rato_philox_uniform(float* out, int32_t T, int32_t C, int64_t M, int64_t ld, uint64_t seed,
                                   uint32_t stream_id, const float* width, const float* low, void* stream) {
  return fill(false, out, T, C, M, ld, seed, rato::PHILOX_STREAM_USER + stream_id, width, low, stream);
}

extern "C" int rato_drone_sample(int64_t M, int64_t ld, int32_t S, float sampler_dt, uint64_t seed, float mass_nom,
                                 float mass_delta, const float* obs_radii, float obs_radii_delta, float* dW,
                                 float* mass, float* Qsym, void* stream) {
  RATO_CLEAR_ERROR();
  if (M <= 0 || ld < M || S <= 0 || S > 65535 || !(sampler_dt >= 0.0f) || !obs_radii) return RATO_EINVAL;
  hipStream_t st = rato::as_stream(stream);
  if (dW) {
    const float s = sqrtf(sampler_dt);
    const float a[4] = {s, s, s, s};
    const int rc = fill(true, dW, S, 3, M, ld, seed, rato::PHILOX_STREAM_DW, a, nullptr, stream);
    if (rc != RATO_OK) return rc;
  }
  if (mass && Qsym) {
    hipLaunchKernelGGL(drone_params_sample_kernel, dim3(nblk(M)), dim3(RATO_BLOCK), 0, st, (long)M, (long)ld, seed,
                       mass_nom, mass_delta, obs_radii[0], obs_radii[1], obs_radii[2], obs_radii_delta, mass, Qsym);
    RATO_LAUNCH_CHECK();
  }
  return RATO_OK;
}

extern "C" int rato_car_sample(int64_t M, int32_t S, float sampler_dt, uint64_t seed, float w_speed_nom,
                               float w_speed_del, float w_rep_nom, float w_rep_del, const float* x0_mean,
                               const float* x0_std, float* dW, float* x0_ped, float* w_speed, float* w_rep,
                               void* stream) {
  RATO_CLEAR_ERROR();
  if (M <= 0 || S <= 0 || S > 65535 || !(sampler_dt >= 0.0f)) return RATO_EINVAL;
  hipStream_t st = rato::as_stream(stream);
  if (dW) {
    const float s = sqrtf(sampler_dt);
    const float a[4] = {s, s, s, s};
    const int rc = fill(true, dW, S, 2, M, M, seed, rato::PHILOX_STREAM_DW, a, nullptr, stream);
    if (rc != RATO_OK) return rc;
  }
  if (x0_ped && w_speed && w_rep) {
    if (!x0_mean || !x0_std) return RATO_EINVAL;
    Affine4 f;
    for (int k = 0; k < 4; ++k) {
      f.a[k] = x0_std[k];
      f.b[k] = x0_mean[k];
    }
    hipLaunchKernelGGL(car_params_sample_kernel, dim3(nblk(M)), dim3(RATO_BLOCK), 0, st, (long)M, seed, w_speed_nom,
                       w_speed_del, w_rep_nom, w_rep_del, f, w_speed, w_rep, x0_ped);
    RATO_LAUNCH_CHECK();
  }
  return RATO_OK;
}

extern "C" int rato_hopper_sample(int64_t M, uint64_t seed, float* a, float* theta, float* tau, void* stream) {
  RATO_CLEAR_ERROR();
  if (M <= 0 || !a || !theta || !tau) return RATO_EINVAL;
  const float a_scale = 0.025f * sqrtf(2.0f / RATO_HOPPER_NFEAT);     // hopper.py:70-72
  hipLaunchKernelGGL(hopper_fields_sample_kernel, dim3(nblk(M), RATO_HOPPER_NFEAT), dim3(RATO_BLOCK), 0,
                     rato::as_stream(stream), (long)M, seed, a_scale, a, theta, tau);
  RATO_LAUNCH_CHECK();
  return RATO_OK;
}
