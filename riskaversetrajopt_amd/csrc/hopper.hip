// Hopper uncertain-friction slip kernels (gfx950).  Replaces the sample-dependent
// part of hopper.py:75-81 (friction_at_px), :300-367 (slip_risk_constraints), the
// slices of jacrev(g) / hessian(lambda.g) that touch the samples (:569,:577-580)
// and the Monte-Carlo check (:901-925).
//
// mu_i(p) = mu_nom + sum_k a_ik cos(theta_ik p + tau_ik), 30 features per sample.
// One lane = one sample (features live in 90 VGPRs, loaded once, coalesced);
// the second grid dimension splits the contact steps so that small batches still
// fill the chip.  This path is transcendental-issue bound, not HBM bound.
#include "rato_common.h"

namespace {

constexpr int NF = RATO_HOPPER_NFEAT;
constexpr float MU_NOM = 0.10f;  // hopper.py:68

// Trig path.  The arguments theta*p + tau lie in [0, pi*p + 2 pi] (theta <= pi, tau <= 2 pi,
// hopper.py:72-74; p is the end-effector x position, |p| < 3 by the NLP bounds), far inside the
// range where the hardware v_sin_f32 / v_cos_f32 (inputs in revolutions) are accurate: measured
// on MI355X against the fp64 oracle over 5e4 samples x 40 contacts, p in [-3, 3]: max |error| 1.5e-6 on
// h = fx - mu fz, 4e-8 on mu, 3.3e-6 on dh/dpx (tolerances 2e-5 / 2e-6 / 2e-5 in the tests), at about
// half the time of the OCML sincosf path (0.20 -> 0.10 ms for the derivative kernel at M = 5e4).
__device__ __forceinline__ void fast_sincos(float x, float& sn, float& cs) {
  const float r = x * 0.15915494309189535f;  // 1/(2 pi)
  sn = __builtin_amdgcn_sinf(r);
  cs = __builtin_amdgcn_cosf(r);
}

template <bool DERIV>
__global__ __launch_bounds__(RATO_BLOCK) void hopper_slip_kernel(
    int M_, int C, int cpg, const float* __restrict__ px, const float* __restrict__ fx,
    const float* __restrict__ fz, const float* __restrict__ a, const float* __restrict__ theta,
    const float* __restrict__ tau, const float* __restrict__ lam, float* __restrict__ Z, int z_atomic,
    float* __restrict__ h, float* __restrict__ dh_dfz, float* __restrict__ dh_dpx,
    float* __restrict__ part_hess) {
  const size_t M = (size_t)M_;
  const size_t m_raw = (size_t)blockIdx.x * RATO_BLOCK + threadIdx.x;
  const bool valid = m_raw < M;
  const size_t m = valid ? m_raw : M - 1;
  float fa[NF], fth[NF], ftau[NF];
#pragma unroll
  for (int k = 0; k < NF; ++k) {
    fa[k] = a[(size_t)k * M + m];
    fth[k] = theta[(size_t)k * M + m];
    ftau[k] = tau[(size_t)k * M + m];
  }
  const int c0 = blockIdx.y * cpg;
  const int c1 = min(C, c0 + cpg);
  float zmax = -INFINITY;
  __shared__ float red[RATO_BLOCK / RATO_WAVE][2];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int c = c0; c < c1; ++c) {
    const float p = px[c], f_x = fx[c], f_z = fz[c];
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f;
#pragma unroll
    for (int k = 0; k < NF; ++k) {
      const float arg = fth[k] * p + ftau[k];
      if (DERIV) {
        float sn, cs;
        fast_sincos(arg, sn, cs);
        s0 += fa[k] * cs;
        const float ath = fa[k] * fth[k];
        s1 += ath * sn;
        s2 += ath * fth[k] * cs;
      } else {
        s0 += fa[k] * __builtin_amdgcn_cosf(arg * 0.15915494309189535f);
      }
    }
    const float mu = MU_NOM + s0;
    const float hv = f_x - mu * f_z;  // hopper.py:322
    zmax = fmaxf(zmax, hv);
    if (valid) {
      if (h) h[(size_t)c * M + m] = hv;
      if (DERIV) {
        if (dh_dfz) dh_dfz[(size_t)c * M + m] = -mu;
        if (dh_dpx) dh_dpx[(size_t)c * M + m] = s1 * f_z;  // -mu'(p) fz
      }
    }
    if (DERIV && part_hess) {  // wave-uniform
      const float l = valid ? lam[(size_t)c * M + m] : 0.0f;
      const float d1 = rato::wave_sum(l * s1);        // lam * d2h/(dpx dfz) = -lam mu'
      const float d2 = rato::wave_sum(l * s2 * f_z);  // lam * d2h/dpx^2   = -lam mu'' fz
      if (lane == 0) {
        red[wave][0] = d1;
        red[wave][1] = d2;
      }
      __syncthreads();
      if (threadIdx.x < 2) {
        float acc = 0.0f;
#pragma unroll
        for (int w = 0; w < RATO_BLOCK / RATO_WAVE; ++w) acc += red[w][threadIdx.x];
        part_hess[((size_t)blockIdx.x * C + c) * 2 + threadIdx.x] = acc;
      }
      __syncthreads();
    }
  }
  if (Z && valid) {
    if (!z_atomic) {
      Z[m] = zmax;
    } else if (zmax >= 0.0f) {  // order-independent float max through integer atomics
      atomicMax(reinterpret_cast<int*>(Z) + m, __float_as_int(zmax));
    } else {
      atomicMin(reinterpret_cast<unsigned*>(Z) + m, __float_as_uint(zmax));
    }
  }
}

}  // namespace

extern "C" int rato_hopper_nblocks(int32_t M) { return M > 0 ? rato::nblocks_for(M) : RATO_EINVAL; }

extern "C" int rato_hopper_slip(int32_t M, int32_t C, const float* px, const float* fx, const float* fz,
                                const float* a, const float* theta, const float* tau, const float* lam,
                                float* Z, float* h, float* dh_dfz, float* dh_dpx, float* part_hess,
                                void* stream) {
  RATO_CLEAR_ERROR();
  if (M <= 0 || C <= 0 || !px || !fx || !fz || !a || !theta || !tau) return RATO_EINVAL;
  if (part_hess && !lam) return RATO_EINVAL;
  hipStream_t st = rato::as_stream(stream);
  const int nblk = rato::nblocks_for(M);
  // split the contacts until there are ~4 waves per SIMD
  long waves = (long)nblk * (RATO_BLOCK / RATO_WAVE);
  int groups = (int)((4096 + waves - 1) / waves);
  if (groups < 1) groups = 1;
  if (groups > C) groups = C;
  const int cpg = (C + groups - 1) / groups;
  groups = (C + cpg - 1) / cpg;
  const int z_atomic = (Z && groups > 1) ? 1 : 0;
  if (z_atomic) {
    hipError_t e = hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(Z), (int)0xff800000u, (size_t)M, st);  // -inf
    if (e != hipSuccess) return RATO_EHIP - (int)e;
  }
  dim3 grid(nblk, groups), block(RATO_BLOCK);
  const bool deriv = dh_dfz || dh_dpx || part_hess;
  if (deriv)
    hipLaunchKernelGGL(hopper_slip_kernel<true>, grid, block, 0, st, M, C, cpg, px, fx, fz, a, theta, tau, lam, Z,
                       z_atomic, h, dh_dfz, dh_dpx, part_hess);
  else
    hipLaunchKernelGGL(hopper_slip_kernel<false>, grid, block, 0, st, M, C, cpg, px, fx, fz, a, theta, tau, lam, Z,
                       z_atomic, h, dh_dfz, dh_dpx, part_hess);
  RATO_LAUNCH_CHECK();
  return RATO_OK;
}
