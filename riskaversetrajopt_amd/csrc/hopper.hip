// Hopper uncertain-friction slip kernels (gfx950).  Replaces the sample-dependent
// part of hopper.py:75-81 (friction_at_px), :300-367 (slip_risk_constraints), the
// slices of jacrev(g) / hessian(lambda.g) that touch the samples (:569,:577-580)
// and the Monte-Carlo check (:901-925).
//
// mu_i(p) = mu_nom + sum_k a_ik cos(theta_ik p + tau_ik), 30 features per sample.
// One lane = one sample (features live in 90 VGPRs, loaded once, coalesced);
// the second grid dimension splits the contact steps so that small batches still
// fill the chip.  This path is transcendental-issue bound, not HBM bound.
#include <stdlib.h>
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

// Launch shape.  A workgroup is 4 waves = SW sample-waves x NW contact-waves (NW = 2 or 4; 1 only as a diagnostic):
// the SW sample-waves take 64 samples each, the NW contact-waves of a sample-wave share its samples and take the
// contacts c = cw, cw + NW, ... ; max_c is folded through LDS.  Small batches split the contacts 4 ways so that the
// chip still gets ~3 waves per SIMD (M = 5e4: 782 workgroups of 64 samples, 3128 waves; measured per step at C4:
// NW = 4 53.8-55.2 us, 2 56.0-58.4, 1 64.8-65.9, 8 waves of 5 contacts in 512-thread workgroups 60.4-62.4); large
// ones 2 ways (M = 1e6: 0.394 ms against 0.404 unsplit and 0.403 at 4).
// Everything a sample needs meets inside ONE workgroup: no atomics on Z, nothing to initialise before the launch
// (round 2 until here: contact groups on blockIdx.y folded with atomic max into a Z pre-filled with -inf -- one more
// node in every step).
// HC = 2: part_hess holds (D1, D2) per contact; HC = 3 (rato_hopper_slip_hessian): also D0 = sum lam dh/dpx, the factor
// of the end-effector map's own curvature in the reference's Lagrangian Hessian (hopper.py:577-580).
template <bool DERIV, bool BYVAL, int HC = 2>
__global__ __launch_bounds__(RATO_BLOCK, DERIV ? 3 : 4) void hopper_slip_kernel(
    int M_, int C, int nw_log2, const float* __restrict__ px, const float* __restrict__ fx,
    const float* __restrict__ fz, const HopperContacts hc, const float* __restrict__ a,
    const float* __restrict__ theta, const float* __restrict__ tau, const float* __restrict__ lam,
    float* __restrict__ Z, float* __restrict__ h, float* __restrict__ dh_dfz, float* __restrict__ dh_dpx,
    float* __restrict__ part_hess) {
  static_assert(NF % 2 == 0, "features are processed in pairs");
  constexpr int NP = NF / 2;
  constexpr int WAVES = RATO_BLOCK / RATO_WAVE;
  constexpr float INV_2PI = 0.15915494309189535f;
  const size_t M = (size_t)M_;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform (SGPR)
  const int NW = 1 << nw_log2, SW = WAVES >> nw_log2;
  const int sw = wave >> nw_log2, cw = wave & (NW - 1);
  const unsigned m_raw = (blockIdx.x * SW + sw) * RATO_WAVE + lane;      // M < 2^31
  const bool valid = m_raw < (unsigned)M_;
  const unsigned m = valid ? m_raw : (unsigned)M_ - 1u;
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
  float zmax = -INFINITY;
  extern __shared__ float lds[];        // [WAVES][64] per-wave maxima | [SW][C][HC] per-wave Hessian sums
  float* zred = lds;
  float* hess_lds = lds + WAVES * RATO_WAVE;
  const bool want_hess = DERIV && part_hess;
  for (int c = cw; c < C; c += NW) {
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
    if (want_hess) {  // wave-uniform; contact c belongs to exactly one contact-wave of each sample-wave
      const float d1 = rato::wave_sum_dpp(l * (s1.x + s1.y));        // lam * d2h/(dpx dfz) = -lam mu'
      const float d2 = rato::wave_sum_dpp(l * (s2.x + s2.y) * f_z);  // lam * d2h/dpx^2   = -lam mu'' fz
      float d0 = 0.0f;
      if (HC == 3) d0 = rato::wave_sum_dpp(l * (s1.x + s1.y) * f_z);   // lam * dh/dpx         = -lam mu' fz
      if (lane == 0) {
        hess_lds[(sw * C + c) * HC + 0] = d1;
        hess_lds[(sw * C + c) * HC + 1] = d2;
        if (HC == 3) hess_lds[(sw * C + c) * HC + 2] = d0;
      }
    }
  }
  const bool fold_z = Z && NW > 1;
  if (fold_z) zred[wave * RATO_WAVE + lane] = zmax;
  if (want_hess || fold_z) __syncthreads();
  if (want_hess) {
    for (int i = threadIdx.x; i < C * HC; i += RATO_BLOCK) {
      float acc = 0.0f;
      for (int w = 0; w < SW; ++w) acc += hess_lds[w * C * HC + i];   // fixed order
      part_hess[(size_t)blockIdx.x * C * HC + i] = acc;
    }
  }
  if (Z && valid && cw == 0) {
    for (int j = 1; j < NW; ++j) zmax = fmaxf(zmax, zred[(wave + j) * RATO_WAVE + lane]);
    Z[m] = zmax;
  }
}

constexpr size_t RATO_HOPPER_LDS_MAX = 160 * 1024;

// contact-waves per sample-wave for a batch of M samples: ~3000 waves or more whenever the batch allows it
inline int hopper_nw_log2(int32_t M) {
  // RATO_HOPPER_NW_LOG2=0|1|2: diagnostic override (A/B runs of the launch shape)
  static const int forced = [] { const char* e = getenv("RATO_HOPPER_NW_LOG2"); return (e && e[0] >= '0' && e[0] <= '2') ? e[0] - '0' : -1; }();
  if (forced >= 0) return forced;
  const long sample_waves = ((long)M + RATO_WAVE - 1) / RATO_WAVE;
  return sample_waves >= 1536 ? 1 : 2;
}
inline int hopper_blocks(int32_t M) {
  const int samples_per_block = RATO_BLOCK >> hopper_nw_log2(M);
  return (int)(((long)M + samples_per_block - 1) / samples_per_block);
}

}  // namespace

extern "C" int rato_hopper_nblocks(int32_t M) { return M > 0 ? hopper_blocks(M) : RATO_EINVAL; }

namespace {
int hopper_slip_impl(int32_t M, int32_t C, const float* px, const float* fx, const float* fz, bool host_inputs,
                     const float* a, const float* theta, const float* tau, const float* lam, float* Z, float* h,
                     float* dh_dfz, float* dh_dpx, float* part_hess, void* stream, int hc = 2) {
  RATO_CLEAR_ERROR();
  if (M <= 0 || C <= 0 || !px || !fx || !fz || !a || !theta || !tau) return RATO_EINVAL;
  if (part_hess && !lam) return RATO_EINVAL;
  if (hc == 3 && !part_hess) return RATO_EINVAL;
  if (host_inputs && C > RATO_HOPPER_MAX_HOST_CONTACTS) return RATO_EINVAL;
  hipStream_t st = rato::as_stream(stream);
  const int nw_log2 = hopper_nw_log2(M);
  dim3 grid(hopper_blocks(M)), block(RATO_BLOCK);
  const bool deriv = dh_dfz || dh_dpx || part_hess;
  const int SW = (RATO_BLOCK / RATO_WAVE) >> nw_log2;
  const size_t lds = (size_t)(RATO_BLOCK + (part_hess ? SW * C * hc : 0)) * sizeof(float);
  // the Hessian sums keep 2 C floats per sample-wave in LDS: C <= 4064 (two sample-waves) / 2032 (four) contacts; the
  // reference's hopper has 2 S / 3 contacts (hopper.py:306-311).  Beyond the 64 KB default the kernels are raised (per
  // device) up to the 160 KB of a CU; past that the call is refused instead of failing inside the launch.
  if (lds > RATO_HOPPER_LDS_MAX) return RATO_EINVAL;
  static rato::DynamicLdsLimit lds_limit;
  {
    const hipError_t e = lds_limit.ensure(lds, [](size_t) {
      hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void*>(hopper_slip_kernel<true, true>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)RATO_HOPPER_LDS_MAX);
      if (err == hipSuccess)
        err = hipFuncSetAttribute(reinterpret_cast<const void*>(hopper_slip_kernel<true, false>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)RATO_HOPPER_LDS_MAX);
      if (err == hipSuccess)
        err = hipFuncSetAttribute(reinterpret_cast<const void*>(hopper_slip_kernel<true, true, 3>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)RATO_HOPPER_LDS_MAX);
      if (err == hipSuccess)
        err = hipFuncSetAttribute(reinterpret_cast<const void*>(hopper_slip_kernel<true, false, 3>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)RATO_HOPPER_LDS_MAX);
      return err;
    });
    if (e != hipSuccess) return RATO_EHIP - (int)e;
  }
  HopperContacts hc_args;
  if (host_inputs) {
    ::memcpy(hc_args.px, px, sizeof(float) * C);
    ::memcpy(hc_args.fx, fx, sizeof(float) * C);
    ::memcpy(hc_args.fz, fz, sizeof(float) * C);
    px = fx = fz = nullptr;
  }
#define RATO_HOPPER_LAUNCH(D, B)                                                                                  \
  hipLaunchKernelGGL((hopper_slip_kernel<D, B>), grid, block, lds, st, M, C, nw_log2, px, fx, fz, hc_args, a, theta, tau, \
                     lam, Z, h, dh_dfz, dh_dpx, part_hess)
  if (hc == 3) {
    if (host_inputs)
      hipLaunchKernelGGL((hopper_slip_kernel<true, true, 3>), grid, block, lds, st, M, C, nw_log2, px, fx, fz, hc_args, a,
                         theta, tau, lam, Z, h, dh_dfz, dh_dpx, part_hess);
    else
      hipLaunchKernelGGL((hopper_slip_kernel<true, false, 3>), grid, block, lds, st, M, C, nw_log2, px, fx, fz, hc_args, a,
                         theta, tau, lam, Z, h, dh_dfz, dh_dpx, part_hess);
  } else if (deriv) {
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

// The same call with part_hess [nblocks][C][3] = per-block sums of (lam d2h/(dpx dfz), lam d2h/dpx^2, lam dh/dpx): everything
// hessian(lambda . g) of the reference (hopper.py:575-580) needs from the samples.  host_inputs != 0: px / fx / fz are host
// arrays of C <= RATO_HOPPER_MAX_HOST_CONTACTS values (kernel arguments); 0: device arrays.
extern "C" int rato_hopper_slip_hessian(int32_t M, int32_t C, const float* px, const float* fx, const float* fz,
                                        int32_t host_inputs, const float* a, const float* theta, const float* tau,
                                        const float* lam, float* Z, float* h, float* dh_dfz, float* dh_dpx,
                                        float* part_hess3, void* stream) {
  return hopper_slip_impl(M, C, px, fx, fz, host_inputs != 0, a, theta, tau, lam, Z, h, dh_dfz, dh_dpx, part_hess3, stream, 3);
}

// ---- jacrev(slip_risk_constraints) in the reference's layout: the CSC value array --------------------------------------
// Rows (hopper.py:351-366, 'saa'): 0 = (M alpha) t + sum y;  1 + i = -y_i;  1 + M + i C + c = h_ic - t - y_i - slack;  last: 0.
// ('baseline', :339-348: rows i C + c = h_ic - slack.)  Columns in Z order (:105-132): xs [t][8], us [t][4], y_i, slack, t.
// With the entries of a column sorted by row, the value array is, in this order:
//   [c][k = x0, x2, x3][i]   dh_ic/dpx * chain[c][k]            3 C M   (chain = d(x0 + x3 sin x2)/d(x0, x2, x3) at step t_c)
//   [c][fx, fz][i]           1, dh_ic/dfz = -mu_i(p_c)            2 C M
//   saa: [i][1, -1, C x -1]  the y_i columns                      M (2 + C)
//   slack: [i][c] -1                                              M C
//   saa: t_risk: M alpha, then [i][c] -1                          1 + M C
// Blocks [0, data_blocks) write the data-dependent part (lane = sample: coalesced reads of dh_dpx / dh_dfz [C][M] and
// coalesced writes); the blocks behind them fill the constant part linearly (skipped with write_constants = 0: the
// constants of a buffer written once do not change).  Exact zeros (sin x2 = 0) are written as zeros: the facade drops them.
namespace {
struct HopperChain {
  float j[RATO_HOPPER_MAX_HOST_CONTACTS][3];
};

template <bool BYVAL>
__global__ __launch_bounds__(RATO_BLOCK) void hopper_emit_jacobian_kernel(int M_, int C, int saa, float M_alpha,
                                                                           const float* __restrict__ dh_dfz,
                                                                           const float* __restrict__ dh_dpx,
                                                                           const HopperChain chain,
                                                                           const float* __restrict__ chain_dev,
                                                                           int data_blocks, float* __restrict__ out) {
  const size_t M = (size_t)M_;
  const int chunks = (M_ + RATO_BLOCK - 1) / RATO_BLOCK;
  if ((int)blockIdx.x < data_blocks) {
    const int c = blockIdx.x / chunks;
    const size_t i = (size_t)(blockIdx.x - c * chunks) * RATO_BLOCK + threadIdx.x;
    if (i >= M) return;
    const float dpx = dh_dpx[(size_t)c * M + i], dfz = dh_dfz[(size_t)c * M + i];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const float j = BYVAL ? chain.j[c][k] : chain_dev[c * 3 + k];
      out[((size_t)c * 3 + k) * M + i] = dpx * j;
    }
    float* u = out + (size_t)3 * C * M + (size_t)c * 2 * M;
    u[i] = 1.0f;
    u[M + i] = dfz;
    return;
  }
  // constant part: [y columns | slack column | t column], one value per thread and step
  const size_t n_y = saa ? M * (size_t)(2 + C) : 0, n_s = M * (size_t)C, n_t = saa ? 1 + M * (size_t)C : 0;
  const size_t total = n_y + n_s + n_t;
  float* o = out + (size_t)5 * C * M;
  const size_t stride = (size_t)(gridDim.x - data_blocks) * RATO_BLOCK;
  for (size_t q = (size_t)(blockIdx.x - data_blocks) * RATO_BLOCK + threadIdx.x; q < total; q += stride) {
    float v = -1.0f;
    if (q < n_y) {
      if (q % (size_t)(2 + C) == 0) v = 1.0f;
    } else if (q == n_y + n_s) {
      v = M_alpha;
    }
    o[q] = v;
  }
}
}  // namespace

extern "C" int64_t rato_hopper_jacobian_nnz(int32_t M, int32_t C, int32_t saa) {
  if (M <= 0 || C <= 0) return RATO_EINVAL;
  return saa ? (int64_t)8 * C * M + (int64_t)2 * M + 1 : (int64_t)6 * C * M;
}

extern "C" int rato_hopper_emit_jacobian_values(int32_t M, int32_t C, int32_t saa, double alpha, const float* dh_dfz,
                                                const float* dh_dpx, const float* chain_host, const float* chain_dev,
                                                int32_t write_constants, float* out, void* stream) {
  RATO_CLEAR_ERROR();
  if (M <= 0 || C <= 0 || !dh_dfz || !dh_dpx || !out || (!chain_host && !chain_dev)) return RATO_EINVAL;
  const bool byval = chain_host && C <= RATO_HOPPER_MAX_HOST_CONTACTS;
  if (!byval && !chain_dev) return RATO_EINVAL;
  HopperChain ch;
  if (byval) ::memcpy(ch.j, chain_host, sizeof(float) * 3 * (size_t)C);
  const int chunks = (M + RATO_BLOCK - 1) / RATO_BLOCK;
  const long data_blocks = (long)C * chunks;
  const int const_blocks = write_constants ? 512 : 0;
  if (data_blocks + const_blocks > 0x7fffffffL) return RATO_EINVAL;
  dim3 grid((unsigned)(data_blocks + const_blocks)), block(RATO_BLOCK);
  hipStream_t st = rato::as_stream(stream);
  const float Ma = (float)((double)M * alpha);
  if (byval)
    hipLaunchKernelGGL((hopper_emit_jacobian_kernel<true>), grid, block, 0, st, M, C, saa ? 1 : 0, Ma, dh_dfz, dh_dpx, ch,
                       chain_dev, (int)data_blocks, out);
  else
    hipLaunchKernelGGL((hopper_emit_jacobian_kernel<false>), grid, block, 0, st, M, C, saa ? 1 : 0, Ma, dh_dfz, dh_dpx, ch,
                       chain_dev, (int)data_blocks, out);
  RATO_LAUNCH_CHECK();
  return RATO_OK;
}
