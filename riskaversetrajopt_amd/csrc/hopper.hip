// Hopper uncertain-friction slip kernels (gfx950).  Replaces the sample-dependent
// part of hopper.py:75-81 (friction_at_px), :300-367 (slip_risk_constraints), the
// slices of jacrev(g) / hessian(lambda.g) that touch the samples (:569,:577-580)
// and the Monte-Carlo check (:901-925).
//
// mu_i(p) = mu_nom + sum_k a_ik cos(theta_ik p + tau_ik), 30 features per sample.
// One lane = one sample (features live in 90 VGPRs, loaded once, coalesced);
// the second grid dimension splits the contact steps so that small batches still
// fill the chip.  This path is transcendental-issue bound, not HBM bound.
#include <string.h>

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
// (__builtin_amdgcn_sinf / cosf on phases pre-scaled by 1/(2 pi))

typedef float hfloat2 __attribute__((ext_vector_type(2)));

// VALU-issue bound (measured: ~280 wave-instructions per contact before this form, 93 % of the issue rate), so the
// instruction count is the roofline.  Features are processed in PAIRS with packed fp32 math (v_pk_fma_f32 /
// v_pk_mul_f32: two features per instruction): per pair one packed FMA for the two phases (theta and tau are
// pre-scaled to revolutions at load), 2 + 2 hardware sin / cos, and one packed FMA per accumulated quantity with
// the loop-invariant products a*theta, a*theta^2 kept in registers.  The lambda-weighted Hessian sums are
// reduced per wave with DPP into an LDS table and combined once per workgroup at the end (no barrier per
// contact); lambda for a contact is fetched before that contact's trig block so that its latency is covered.
// The per-contact inputs (foot position and contact force of every contact step) change with every NLP iterate and
// come from the host.  BYVAL: they travel in the kernel's argument block (scalar loads from the kernarg segment) --
// no staging buffer, no upload node in front of the kernel; otherwise they are read through device pointers.
struct HopperContacts {
  float px[RATO_HOPPER_MAX_HOST_CONTACTS], fx[RATO_HOPPER_MAX_HOST_CONTACTS], fz[RATO_HOPPER_MAX_HOST_CONTACTS];
};

template <bool DERIV, bool BYVAL>
__global__ __launch_bounds__(RATO_BLOCK, DERIV ? 3 : 4) void hopper_slip_kernel(
    int M_, int C, int cpg, const float* __restrict__ px, const float* __restrict__ fx,
    const float* __restrict__ fz, const HopperContacts hc, const float* __restrict__ a, const float* __restrict__ theta,
    const float* __restrict__ tau, const float* __restrict__ lam, float* __restrict__ Z, int z_atomic,
    float* __restrict__ h, float* __restrict__ dh_dfz, float* __restrict__ dh_dpx,
    float* __restrict__ part_hess) {
  static_assert(NF % 2 == 0, "features are processed in pairs");
  constexpr int NP = NF / 2;
  constexpr float INV_2PI = 0.15915494309189535f;
  const size_t M = (size_t)M_;
  const size_t m_raw = (size_t)blockIdx.x * RATO_BLOCK + threadIdx.x;
  const bool valid = m_raw < M;
  const size_t m = valid ? m_raw : M - 1;
  hfloat2 fa[NP], rth[NP], rtau[NP];   // amplitude | theta / 2 pi | tau / 2 pi

#pragma unroll
  for (int k = 0; k < NP; ++k) {
    hfloat2 th, ta;
    fa[k].x = a[(size_t)(2 * k) * M + m];
    fa[k].y = a[(size_t)(2 * k + 1) * M + m];
    th.x = theta[(size_t)(2 * k) * M + m];
    th.y = theta[(size_t)(2 * k + 1) * M + m];
    ta.x = tau[(size_t)(2 * k) * M + m];
    ta.y = tau[(size_t)(2 * k + 1) * M + m];
    rth[k] = th * INV_2PI;
    rtau[k] = ta * INV_2PI;
  }
  const int c0 = blockIdx.y * cpg;
  const int c1 = min(C, c0 + cpg);
  float zmax = -INFINITY;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  extern __shared__ float hess_lds[];   // [waves][cpg][2]
  const bool want_hess = DERIV && part_hess;
  for (int c = c0; c < c1; ++c) {
    const float p = BYVAL ? hc.px[c] : px[c], f_x = BYVAL ? hc.fx[c] : fx[c], f_z = BYVAL ? hc.fz[c] : fz[c];
    float l = 0.0f;
    if (want_hess && valid) l = lam[(size_t)c * M + m];   // consumed after the trig block
    hfloat2 p2;
    p2.x = p;
    p2.y = p;
    hfloat2 s0 = {0.0f, 0.0f}, s1 = {0.0f, 0.0f}, s2 = {0.0f, 0.0f};
#pragma unroll
    for (int k = 0; k < NP; ++k) {
      const hfloat2 r = rth[k] * p2 + rtau[k];   // phase in revolutions (hardware trig unit)
      hfloat2 cs;
      cs.x = __builtin_amdgcn_cosf(r.x);
      cs.y = __builtin_amdgcn_cosf(r.y);
      if (!DERIV) {
        s0 += fa[k] * cs;
      } else {
        // a theta and a theta^2 are NOT kept in registers (60 VGPRs: 194-202 -> 2 waves per SIMD).  Every product
        // starts from a trig value, so nothing is loop invariant and nothing can be hoisted back into registers:
        //   u = a cos,  s0 += u,  s2 += (u r) r;   v = a sin,  s1 += v r      (r = theta / 2 pi; the factors 2 pi and
        //   (2 pi)^2 are applied once per contact to the sums): 7 packed ops per feature pair instead of 4.
        hfloat2 sn;
        sn.x = __builtin_amdgcn_sinf(r.x);
        sn.y = __builtin_amdgcn_sinf(r.y);
        const hfloat2 u = fa[k] * cs;
        s0 += u;
        s2 += (u * rth[k]) * rth[k];
        s1 += (fa[k] * sn) * rth[k];
      }
    }
    if (DERIV) {
      s1 *= 6.28318530717958647692f;
      s2 *= 39.4784176043574344753f;
    }
    const float mu = MU_NOM + (s0.x + s0.y);
    const float hv = f_x - mu * f_z;  // hopper.py:322
    zmax = fmaxf(zmax, hv);
    if (valid) {
      if (h) h[(size_t)c * M + m] = hv;
      if (DERIV) {
        if (dh_dfz) dh_dfz[(size_t)c * M + m] = -mu;
        if (dh_dpx) dh_dpx[(size_t)c * M + m] = (s1.x + s1.y) * f_z;  // -mu'(p) fz
      }
    }
    if (want_hess) {  // wave-uniform
      const float d1 = rato::wave_sum_dpp(l * (s1.x + s1.y));        // lam * d2h/(dpx dfz) = -lam mu'
      const float d2 = rato::wave_sum_dpp(l * (s2.x + s2.y) * f_z);  // lam * d2h/dpx^2   = -lam mu'' fz
      if (lane == 0) {
        hess_lds[(wave * cpg + (c - c0)) * 2 + 0] = d1;
        hess_lds[(wave * cpg + (c - c0)) * 2 + 1] = d2;
      }
    }
  }
  if (want_hess) {
    __syncthreads();
    for (int i = threadIdx.x; i < (c1 - c0) * 2; i += RATO_BLOCK) {
      float acc = 0.0f;
#pragma unroll
      for (int w = 0; w < RATO_BLOCK / RATO_WAVE; ++w) acc += hess_lds[w * cpg * 2 + i];   // fixed order
      part_hess[((size_t)blockIdx.x * C + c0) * 2 + i] = acc;
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

namespace {
// Z = -inf before the contact groups fold their maxima into it.  A kernel, not hipMemsetD32Async: inside a captured
// hipGraph a memset node costs ~10 us of cross-queue hand-off on ROCm 7.2, a kernel node stays on the launch queue.
__global__ __launch_bounds__(RATO_BLOCK) void fill_neg_inf_kernel(float* __restrict__ Z, int M) {
  const int i = blockIdx.x * RATO_BLOCK + threadIdx.x;
  if (i < M) Z[i] = -INFINITY;
}

int hopper_slip_impl(int32_t M, int32_t C, const float* px, const float* fx, const float* fz, bool host_inputs,
                     const float* a, const float* theta, const float* tau, const float* lam, float* Z, float* h,
                     float* dh_dfz, float* dh_dpx, float* part_hess, void* stream) {
  RATO_CLEAR_ERROR();
  if (M <= 0 || C <= 0 || !px || !fx || !fz || !a || !theta || !tau) return RATO_EINVAL;
  if (part_hess && !lam) return RATO_EINVAL;
  if (host_inputs && C > RATO_HOPPER_MAX_HOST_CONTACTS) return RATO_EINVAL;
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
  if (z_atomic)
    hipLaunchKernelGGL(fill_neg_inf_kernel, dim3((M + RATO_BLOCK - 1) / RATO_BLOCK), dim3(RATO_BLOCK), 0, st, Z, (int)M);
  dim3 grid(nblk, groups), block(RATO_BLOCK);
  const bool deriv = dh_dfz || dh_dpx || part_hess;
  const size_t lds = deriv ? (size_t)(RATO_BLOCK / RATO_WAVE) * cpg * 2 * sizeof(float) : 0;
  HopperContacts hc;
  if (host_inputs) {
    ::memcpy(hc.px, px, sizeof(float) * C);
    ::memcpy(hc.fx, fx, sizeof(float) * C);
    ::memcpy(hc.fz, fz, sizeof(float) * C);
    px = fx = fz = nullptr;
  }
#define RATO_HOPPER_LAUNCH(D, B)                                                                                  \
  hipLaunchKernelGGL((hopper_slip_kernel<D, B>), grid, block, lds, st, M, C, cpg, px, fx, fz, hc, a, theta, tau, lam, \
                     Z, z_atomic, h, dh_dfz, dh_dpx, part_hess)
  if (deriv) {
    if (host_inputs) RATO_HOPPER_LAUNCH(true, true); else RATO_HOPPER_LAUNCH(true, false);
  } else {
    if (host_inputs) RATO_HOPPER_LAUNCH(false, true); else RATO_HOPPER_LAUNCH(false, false);
  }
#undef RATO_HOPPER_LAUNCH
  RATO_LAUNCH_CHECK();
  return RATO_OK;
}
}  // namespace

extern "C" int rato_hopper_slip(int32_t M, int32_t C, const float* px, const float* fx, const float* fz,
                                const float* a, const float* theta, const float* tau, const float* lam,
                                float* Z, float* h, float* dh_dfz, float* dh_dpx, float* part_hess,
                                void* stream) {
  return hopper_slip_impl(M, C, px, fx, fz, false, a, theta, tau, lam, Z, h, dh_dfz, dh_dpx, part_hess, stream);
}

extern "C" int rato_hopper_slip_host_inputs(int32_t M, int32_t C, const float* px_host, const float* fx_host,
                                            const float* fz_host, const float* a, const float* theta,
                                            const float* tau, const float* lam, float* Z, float* h, float* dh_dfz,
                                            float* dh_dpx, float* part_hess, void* stream) {
  return hopper_slip_impl(M, C, px_host, fx_host, fz_host, true, a, theta, tau, lam, Z, h, dh_dfz, dh_dpx, part_hess,
                          stream);
}
