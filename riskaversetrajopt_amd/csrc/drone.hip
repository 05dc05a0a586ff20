// Drone SAA kernels (gfx950).  Replaces drone_risk.py:122-296,656-662 of the
// reference (per-sample Euler–Maruyama rollout, ellipsoidal-obstacle constraint,
// control-Jacobian, per-block stage of the sample mean).
//
// Mapping: one lane = one sample (SoA, sample index fastest) so every global
// access of a wave is 256 contiguous bytes.  The linearize kernel adds a second
// grid dimension over groups of control columns: the forward sensitivity
// Phi_{t+1,s} = A_t Phi_{t,s}, Phi_{s+1,s} = B_s is independent per column s once
// the trajectory is known, and each thread re-rolls the (cheap) trajectory while
// it propagates CPT columns held in registers.  Axes decouple:
//   A_t = [[1, dt], [-kp dt/m, 1 - dt (kd + 2 c_d |v_t|)/m]],  B = [0, dt/m]^T.
#include <stdlib.h>

#include <atomic>

#include "philox.h"
#include "rato_common.h"
#include <type_traits>
#include "rato_select.h"

namespace {

constexpr int NOBS = RATO_DRONE_NOBS;

// work queues of the dynamic launch forms (row-parallel linearize, eval): {next tile, workgroups gone}; zero at load,
// every launch leaves its queue zeroed again; handed out by rato::TileQueuePool (per stream / per captured launch)
__device__ unsigned g_tile_queues[RATO_QUEUES_TOTAL * 2];

struct SampleConsts {
  float inv_m, a21, cn, dtm;
  float q00[NOBS], qs[NOBS], q11[NOBS];
};

__device__ __forceinline__ SampleConsts load_consts(const rato_drone_params& P, const float* __restrict__ mass,
                                                    const float* __restrict__ Qsym, size_t ld, size_t m) {
  SampleConsts c;
  c.inv_m = 1.0f / mass[m];
  c.a21 = -P.kp * P.dt * c.inv_m;
  c.cn = sqrtf(P.dt) * P.beta * c.inv_m;  // sqrt(dt) * (beta/m): drone_risk.py:136,151
  c.dtm = P.dt * c.inv_m;
#pragma unroll
  for (int j = 0; j < NOBS; ++j) {
    c.q00[j] = Qsym[(size_t)(j * 3 + 0) * ld + m];
    c.qs[j] = Qsym[(size_t)(j * 3 + 1) * ld + m];
    c.q11[j] = Qsym[(size_t)(j * 3 + 2) * ld + m];
  }
  return c;
}

// One Euler-Maruyama step of one axis (drone_risk.py:122-131,148-153) and the obstacle row g_j = 1 - (p - o_j)' Q_j (p - o_j)
// (:169-196), with every rounding spelled out (contraction off: only the written fmas fuse).
// -ffp-contract=fast leaves the contraction pattern to the optimiser, and the same source compiled into ANOTHER kernel came
// out one rounding different (1e-7 relative, round 5).  Every kernel that reports Z / g / trajectories of the plain rollout
// -- drone_eval_kernel (trajectories, Philox) and drone_eval_tiles_kernel -- calls THESE, so that they agree to the bit by
// construction and not by what a compiler version happens to fuse.
__device__ __forceinline__ void step_axis_exact(const rato_drone_params& P, const SampleConsts& c, float u, float xi,
                                                float& p, float& v) {
#pragma clang fp contract(off)
  const float kv = P.kd * v;
  const float pn = __builtin_fmaf(P.dt, v, p);
  const float s = __builtin_fmaf(P.kp, p, kv);
  const float dq = v * (P.drag * fabsf(v));
  const float w = c.inv_m * dq;
  const float acc = __builtin_fmaf(c.inv_m, u - s, -w);
  v = __builtin_fmaf(c.cn, xi, __builtin_fmaf(P.dt, acc, v));
  p = pn;
}
__device__ __forceinline__ float obstacle_value_exact(const rato_drone_params& P, const SampleConsts& c, int j, float px, float py) {
#pragma clang fp contract(off)
  const float dx = px - P.obs_xy[j][0], dy = py - P.obs_xy[j][1];
  float a = dy * (c.qs[j] * dx);
  a = __builtin_fmaf(dx, c.q00[j] * dx, a);
  a = __builtin_fmaf(dy, c.q11[j] * dy, a);
  return 1.0f - a;
}

// PHILOX: the noise of step t is REGENERATED in the kernel (Philox4x32-10 at counter (m, t), philox.h) instead of
// read from dW: bit-identical to rato_drone_sample's dW, no 12 B per sample-step of HBM reads, no 3 S floats per
// sample of HBM capacity.  noise_scale = sqrt(sampler dt) (drone_utils.py:90).
template <bool PHILOX>
__device__ __forceinline__ void drone_eval_block(
    const rato_drone_params& P, size_t blk, const float* __restrict__ us, const float* __restrict__ dW, uint64_t seed,
    float noise_scale, const float* __restrict__ mass, const float* __restrict__ Qsym, float* __restrict__ Z,
    float* __restrict__ xs, float* __restrict__ g) {
  const size_t M = (size_t)P.M, ld = (size_t)P.ld;
  const size_t m = blk * RATO_BLOCK + threadIdx.x;
  if (m >= M) return;
  const int S = P.S;
  const SampleConsts c = load_consts(P, mass, Qsym, ld, m);
  float p[3], v[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    p[a] = P.x_init[a];
    v[a] = P.x_init[3 + a];
  }
  if (xs) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      xs[(size_t)a * ld + m] = p[a];
      xs[(size_t)(3 + a) * ld + m] = v[a];
    }
  }
  float zmax = -INFINITY;
  float xi[3];
  if (!PHILOX) {
#pragma unroll
    for (int a = 0; a < 3; ++a) xi[a] = dW[(size_t)a * ld + m];
  }
  for (int t = 0; t < S; ++t) {
    float nxt[3];
    if (PHILOX) {
      const rato::u32x4 r = rato::philox_at(seed, rato::PHILOX_STREAM_DW, (uint32_t)t, (uint64_t)m);
      float n3;
      rato::box_muller(r.x, r.y, xi[0], xi[1]);
      rato::box_muller(r.z, r.w, xi[2], n3);
#pragma unroll
      for (int a = 0; a < 3; ++a) xi[a] *= noise_scale;
    } else {
      const int tn = (t + 1 < S) ? t + 1 : t;  // prefetch next step's noise
#pragma unroll
      for (int a = 0; a < 3; ++a) nxt[a] = dW[(size_t)(tn * 3 + a) * ld + m];
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) step_axis_exact(P, c, us[t * 3 + a], xi[a], p[a], v[a]);
    if (xs) {
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        xs[((size_t)(t + 1) * 6 + a) * ld + m] = p[a];
        xs[((size_t)(t + 1) * 6 + 3 + a) * ld + m] = v[a];
      }
    }
#pragma unroll
    for (int j = 0; j < NOBS; ++j) {
      const float gj = obstacle_value_exact(P, c, j, p[0], p[1]);
      zmax = fmaxf(zmax, gj);
      if (g) g[((size_t)j * S + t) * ld + m] = gj;
    }
    if (!PHILOX) {
#pragma unroll
      for (int a = 0; a < 3; ++a) xi[a] = nxt[a];
    }
  }
  if (Z) Z[m] = zmax - P.tol;
}

// tile_queue != NULL (large batches): 8 workgroups per CU, each taking 256-sample blocks from a global counter until
// none is left -- the hardware deals a plain grid out to the XCDs up front and every other XCD streams slower (see
// drone_linearize_rows_kernel), so with one block per workgroup half the chip idles at the end of the launch.
template <bool PHILOX>
__global__ __launch_bounds__(RATO_BLOCK) void drone_eval_kernel(
    rato_drone_params P, const float* __restrict__ us, const float* __restrict__ dW, uint64_t seed,
    float noise_scale, const float* __restrict__ mass, const float* __restrict__ Qsym, float* __restrict__ Z,
    float* __restrict__ xs, float* __restrict__ g, unsigned* __restrict__ tile_queue, int nblk) {
  if (!tile_queue) {
    drone_eval_block<PHILOX>(P, blockIdx.x, us, dW, seed, noise_scale, mass, Qsym, Z, xs, g);
    return;
  }
  __shared__ int next_blk;
  for (int blk = blockIdx.x; blk < nblk;) {
    drone_eval_block<PHILOX>(P, (size_t)blk, us, dW, seed, noise_scale, mass, Qsym, Z, xs, g);
    __syncthreads();
    if (threadIdx.x == 0)
      next_blk = (int)gridDim.x + (int)__hip_atomic_fetch_add(tile_queue, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    blk = next_blk;
  }
  if (threadIdx.x == 0) {
    const unsigned gone = __hip_atomic_fetch_add(tile_queue + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (gone == gridDim.x - 1) {
      __hip_atomic_store(tile_queue, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(tile_queue + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// Monte-Carlo validation batches (drone_risk.py:643-725: M = 1e4) are LATENCY bound: drone_eval_kernel at C2 runs 40
// workgroups on 256 CUs and every one of its 50 steps waits for a global load issued one step earlier (20.6 us for
// 6.5 MB: 0.8 us of HBM time).  This form, for calls that want Z (and g) but no trajectories:
//   * one wave = one tile of 64 samples, four tiles per workgroup; the vertical axis is not rolled out at all (no
//     obstacle row reads it: drone_risk.py:174 uses [:2,:2] of Q and the first two coordinates);
//   * the noise of 2 x EV_TB steps is requested before the first step is taken and the next batch is always in flight
//     (two register buffers): the rollout is bound by its own arithmetic chain, not by memory;
//   * the statistics of Z ride in the same launch (params.stats_*: extra workgroups at the end of the grid run the exact
//     selection as soon as the last tile's Z has landed -- rato_select.h), so that a Monte-Carlo step is ONE launch.
// Same arithmetic per sample as drone_eval_kernel, rounding for rounding (step_axis_exact, obstacle_value_exact): Z and g
// agree to the bit (tests/test_gpu_fused_stats.py).
constexpr int EV_TB = 16;
constexpr int EV_NW = RATO_BLOCK / RATO_WAVE;

// blockIdx.y = control sequence k of a batch of K (rato_drone_eval_batch: us [K][S][3], Z [K][ldz]); K = 1: rato_drone_eval.
template <bool WANT_G>
__global__ __launch_bounds__(RATO_BLOCK) void drone_eval_tiles_kernel(
    rato_drone_params P, const float* __restrict__ us_base, const float* __restrict__ dW, const float* __restrict__ mass,
    const float* __restrict__ Qsym, float* __restrict__ Z_base, long ldz, float* __restrict__ g, int n_tiles,
    const rato_sel::StatsTail tail) {
  extern __shared__ __attribute__((aligned(16))) unsigned char ev_lds[];
  if (tail.is_stats((int)blockIdx.x)) {
    rato_sel::stats_tail_run<RATO_BLOCK>(tail, Z_base, (long)P.M, ev_lds);
    return;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int tile = (int)blockIdx.x * EV_NW + wave;
  if (tile >= n_tiles) return;                     // (wave-uniform)
  const size_t M = (size_t)P.M, ld = (size_t)P.ld;
  const size_t m_raw = (size_t)tile * RATO_WAVE + lane;
  const bool valid = m_raw < M;
  const size_t m = valid ? m_raw : M - 1;
  const int S = P.S;
  const float* __restrict__ us = us_base + (size_t)blockIdx.y * S * 3;
  float* __restrict__ Z = Z_base ? Z_base + (size_t)blockIdx.y * ldz : nullptr;
  float xa[EV_TB][2], xb[EV_TB][2];
  auto load = [&](float (&xi)[EV_TB][2], int t0) {
#pragma unroll
    for (int i = 0; i < EV_TB; ++i) {
      const int t = (t0 + i < S) ? t0 + i : S - 1;
      xi[i][0] = dW[((size_t)t * 3 + 0) * ld + m];
      xi[i][1] = dW[((size_t)t * 3 + 1) * ld + m];
    }
  };
  load(xa, 0);                                     // in flight before anything else
  load(xb, EV_TB);
  // the controls of 64 steps live in the lanes of two registers (lane l: step 64 c + l) and reach the step as a
  // v_readlane: a scalar load per step put its memory latency into every step of the one wave a SIMD runs here
  float ux = 0.0f, uy = 0.0f;
  auto load_us = [&](int t0) {
    const int t = (t0 + lane < S) ? t0 + lane : S - 1;
    ux = us[t * 3 + 0];
    uy = us[t * 3 + 1];
  };
  load_us(0);
  const SampleConsts c = load_consts(P, mass, Qsym, ld, m);
  float p[2] = {P.x_init[0], P.x_init[1]}, v[2] = {P.x_init[3], P.x_init[4]};
  float zmax = -INFINITY;
  auto steps = [&](const float (&xi)[EV_TB][2], int t0) {
#pragma unroll
    for (int i = 0; i < EV_TB; ++i) {
      const int t = t0 + i;
      if (t < S) {
        const float u0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ux), t & 63));
        const float u1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(uy), t & 63));
        step_axis_exact(P, c, u0, xi[i][0], p[0], v[0]);
        step_axis_exact(P, c, u1, xi[i][1], p[1], v[1]);
#pragma unroll
        for (int j = 0; j < NOBS; ++j) {
          const float gj = obstacle_value_exact(P, c, j, p[0], p[1]);
          zmax = fmaxf(zmax, gj);
          if (WANT_G && valid) g[((size_t)j * S + t) * ld + m] = gj;
        }
      }
    }
  };
  for (int t0 = 0; t0 < S; t0 += 2 * EV_TB) {
    if (t0 && (t0 & 63) == 0) load_us(t0);         // (2 EV_TB divides 64: a chunk of controls never changes inside a batch)
    steps(xa, t0);
    if (t0 + 2 * EV_TB < S) load(xa, t0 + 2 * EV_TB);
    steps(xb, t0 + EV_TB);
    if (t0 + 3 * EV_TB < S) load(xb, t0 + 3 * EV_TB);
  }
  if (!Z) return;
  if (!tail.ws) {
    if (valid) Z[m] = zmax - P.tol;
    return;
  }
  // statistics in this launch: Z as agent-scope stores, completed, then the tile is counted in; the tile that completes
  // the count raises z_ready (the protocol of the row-parallel linearize kernels)
  if (valid)
    __hip_atomic_store(reinterpret_cast<unsigned*>(Z) + m, __float_as_uint(zmax - P.tol), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (lane == 0) {
    unsigned* z_signal = tail.ws->sig;
    const unsigned cnt = __hip_atomic_fetch_add(z_signal + rato_sel::SIG_Z_COUNT, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (cnt == (unsigned)n_tiles - 1u) {
      __hip_atomic_store(z_signal + rato_sel::SIG_Z_COUNT, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(z_signal + rato_sel::SIG_Z_READY, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// ---------------------------------------------------------------------------
// Generators-only linearization: the whole linearization of a sample as 12 S numbers -- the state-dependent entry of
// the step Jacobian for the three axes (A22), the row gradients W and g_up -- WITHOUT the S(S-1) entries of Phi
// (Phi[t,s,a] = e_0' A_t ... A_{s+1} B is generated from A22 by whoever needs it: rato_drone_rowmax_implicit,
// rato_drone_tail_rows_implicit).  One lane per sample, no LDS tables: a forward pass (rollout + the linear
// response dx(u_bar) for g_up) that is HBM-read bound like the eval kernel, then the backward final-state
// adjoint over the a22 the lane has just written.  60 B per sample-step instead of 245 B (factored) / 613 B (products).
// TABLES: W and g_up are written (the table forms of the cut oracle); WANT_Z: Z is.  A reduced SCP iteration whose oracle
// re-runs the rollout (the benchmarked configuration) needs neither -- only the sample sums of the final rows -- and then
// neither the tangent of the linearized dynamics nor any obstacle row is formed: <false, false> is a third of the work.
#ifndef RATO_GEN_LDS_REDUCE
#define RATO_GEN_LDS_REDUCE 1   // A/B: 0 = six fp64 DPP trees per backward step
#endif
template <bool TABLES, bool WANT_Z>
__global__ __launch_bounds__(RATO_BLOCK) void drone_linearize_generators_kernel(
    rato_drone_params P, const float* __restrict__ us, const float* __restrict__ dW,
    const float* __restrict__ mass, const float* __restrict__ Qsym, float* __restrict__ A22,
    float* __restrict__ W, float* __restrict__ g_up, float* __restrict__ Z, float* __restrict__ part) {
  // Arithmetic in fp64, outputs rounded ONCE to fp32 (round 3).  The kernel is bound by the latency of its loads, one
  // lane per sample: ~100 flops per step cost nothing in double precision, and the tables then carry 6e-8 of rounding
  // instead of the ~1e-6 that 50 fp32 Euler steps accumulate -- which the cutting-plane solve of an SCP subproblem whose
  // step from the linearization point is O(1) (the iteration where the CVaR rows switch on) amplified to 4e-5 in u
  // (profiles/r03_k_parity_sweep.txt).  Inputs are the same fp32 arrays as everywhere else.
  const size_t M = (size_t)P.M, ld = (size_t)P.ld;
  const size_t m_raw = (size_t)blockIdx.x * RATO_BLOCK + threadIdx.x;
  const bool valid = m_raw < M;
  const size_t m = valid ? m_raw : M - 1;
  const int S = P.S;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const double dt = P.dt64, kp = P.kp64, kd = P.kd64, drag = P.drag64;   // the fp64 constants (rato_saa.h)
  const double inv_m = 1.0 / (double)mass[m];
  const double a21 = -kp * dt * inv_m, dtm = dt * inv_m;
  const double cn = sqrt(dt) * P.beta64 * inv_m;   // sqrt(dt) * (beta/m): drone_risk.py:136,151
  const double c1 = dt * kd * inv_m, one_c1 = 1.0 - c1, c2 = dt * drag * inv_m, c22 = 2.0 * c2;
  double q00[NOBS], qs[NOBS], q11[NOBS];
#pragma unroll
  for (int j = 0; j < NOBS; ++j) {
    q00[j] = (double)Qsym[(size_t)(j * 3 + 0) * ld + m];
    qs[j] = (double)Qsym[(size_t)(j * 3 + 1) * ld + m];
    q11[j] = (double)Qsym[(size_t)(j * 3 + 2) * ld + m];
  }
  double p[3], v[3], dp[2] = {0.0, 0.0}, dv[2] = {0.0, 0.0};   // state | response of the linearized x, y axes to u_bar
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    p[a] = P.x_init64[a];
    v[a] = P.x_init64[3 + a];
  }
  double zmax = -INFINITY;
  constexpr int TB = 8;   // the noise of 8 steps (24 loads) is requested as one batch and the NEXT batch is in flight while a
                          // batch is consumed (two register buffers): the kernel is bound by load latency (one lane per
                          // sample, 50 sequential steps), not by bandwidth or flops
  // Addresses: every array is walked through a UNIFORM row pointer (scalar registers) advanced by whole rows + the lane's
  // 32-bit sample offset -- the rows' 64-bit index arithmetic was a quarter of this kernel's instructions, and the kernel is
  // bound by instruction issue (~200 instructions per forward + backward step of the one or two waves of a SIMD).  The
  // pointers stop at the last row, so a batch requested past the horizon re-reads row S - 1 (never consumed).
  // (byte offsets as 32-bit numbers: `uniform pointer + zero-extended 32-bit lane offset` is the form the global
  //  load / store instructions take directly -- scalar base, one offset register -- with no per-access 64-bit add)
  const unsigned boff0 = (unsigned)m * 4u, boff1 = (unsigned)(ld + m) * 4u, boff2 = (unsigned)(2 * ld + m) * 4u;
  auto at = [](const float* base, unsigned boff) -> const float& {
    return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + boff);
  };
  auto at_w = [](float* base, unsigned boff) -> float& {
    return *reinterpret_cast<float*>(reinterpret_cast<char*>(base) + boff);
  };
  const size_t row3 = 3 * ld;
  const float* nz = dW;                      // row 3 t of the noise
  const float* const nz_last = dW + (size_t)(S - 1) * row3;
  float* a22w = A22;                         // row 3 t of the table being written
  auto load_noise = [&](float (&xib)[TB][3]) {   // the next 8 steps' noise
#pragma unroll
    for (int i = 0; i < TB; ++i) {
      xib[i][0] = at(nz, boff0);
      xib[i][1] = at(nz, boff1);
      xib[i][2] = at(nz, boff2);
      nz = (nz != nz_last) ? nz + row3 : nz;
    }
  };
  auto forward_steps = [&](const float (&xib)[TB][3], int t0, auto guarded) {
#pragma unroll
    for (int i = 0; i < TB; ++i) {
    const int t = t0 + i;
    if (!decltype(guarded)::value || t < S) {
    const float* xi = xib[i];
    double a22[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      // A_t[1,1] = 1 - e22 at the state BEFORE the step.  The table stores e22 = dt (k_d + 2 c_d |v|) / m ~ 1e-3, not a22
      // itself: rounded to fp32 it is exact to 1e-10 of a22, where a22 ~ 1 would carry 6e-8 -- and a row of Phi is a
      // product of up to S of them.  The consumers rebuild a22 = 1 - e22 in fp64 from the SAME fp32 number.
      const float e22 = (float)fma(c22, fabs(v[a]), c1);   // dt (kd + 2 drag |v|) / m
      a22[a] = 1.0 - (double)e22;
      if (valid) at_w(a22w, a == 0 ? boff0 : (a == 1 ? boff1 : boff2)) = e22;
    }
    a22w += row3;
    // d x_{t+1} = A_t d x_t + B u_t  (x, y): the forward form of the adjoint row sweep; d p(t+1) = (Phi u_bar)[t]
    if (TABLES) {
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        const double ndp = dp[a] + dt * dv[a];
        const double ndv = a21 * dp[a] + a22[a] * dv[a] + dtm * (double)us[t * 3 + a];
        dp[a] = ndp;
        dv[a] = ndv;
      }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {   // one Euler-Maruyama step of one axis (drone_risk.py:122-131,148-153), multiplied out
      // as in the cut oracle's rollout kernels (cvar.hip): v' = ((1 - c1) - c2 |v|) v + (dt/m) u - (kp dt/m) p + cn xi
      const double av = fabs(v[a]);
      const double tv = fma(a21, p[a], fma(dtm, (double)us[t * 3 + a], cn * (double)xi[a]));
      const double pn = fma(dt, v[a], p[a]);
      v[a] = fma(fma(-c2, av, one_c1), v[a], tv);
      p[a] = pn;
    }
    if (TABLES || WANT_Z) {
#pragma unroll
    for (int j = 0; j < NOBS; ++j) {
      const double dx = p[0] - P.obs_xy64[j][0], dy = p[1] - P.obs_xy64[j][1];
      const double gj = 1.0 - (q00[j] * dx * dx + qs[j] * dx * dy + q11[j] * dy * dy);
      const double wx = -(2.0 * q00[j] * dx + qs[j] * dy), wy = -(qs[j] * dx + 2.0 * q11[j] * dy);
      zmax = fmax(zmax, gj);
      if (TABLES && valid) {   // (!TABLES: a caller whose cut oracle re-runs the rollout needs only the sample sums)
        W[(((size_t)j * S + t) * 2 + 0) * ld + m] = (float)wx;
        W[(((size_t)j * S + t) * 2 + 1) * ld + m] = (float)wy;
        // -g + (grad g) . u   (drone_risk.py:278), or -- rows_out = 1 -- the constraint value g itself
        g_up[((size_t)j * S + t) * ld + m] = (float)(P.rows_out ? gj : (-gj + wx * dp[0] + wy * dp[1]));
      }
    }
    }
    }
    }   // step t of the batch
  };
  {
    float xa[TB][3], xb[TB][3];
    load_noise(xa);
    // (every step behind its guard: whole unguarded batches, as in the cut oracle's rowmax kernel, let the scheduler hoist
    //  two batches' loads and conversions and took this kernel from 139 to 298 registers -- one wave per SIMD, or spills)
    for (int t0 = 0; t0 < S; t0 += 2 * TB) {
      load_noise(xb);
      forward_steps(xa, t0, std::true_type{});
      load_noise(xa);
      forward_steps(xb, t0 + TB, std::true_type{});
    }
  }
  if (WANT_Z && Z && valid) Z[m] = (float)(zmax - P.tol64);
  // final-state Jacobian d x_S / d u_s (rows P, V of each axis), summed over the block's samples, and the rhs
  // (drone_risk.py:271): adjoint from S over the a22 (recomputed from the fp32 table this lane wrote: the consumers of
  // the table see the same numbers)
  extern __shared__ double gen_red[];   // [waves][S + 1][6]: per-wave sums, combined once after the sweep
  constexpr int NWV = RATO_BLOCK / RATO_WAVE;
#if RATO_GEN_LDS_REDUCE
  // The six sums over the wave's samples that every backward step needs used to be six fp64 DPP trees per step (76 DPP
  // moves + 36 adds of the ~160 instructions of a step: the kernel is bound by instruction issue at M ~ 1e5).  Now the
  // lanes park their six numbers of FOUR steps in a tile of LDS ([24 columns][64 lanes], rows padded to 65) and 48 lanes
  // sum half a column each, serially: ~18 instructions per step instead of 112.  Fixed order: lanes 0..31, then 32..63.
  double* const Tw = gen_red + (size_t)NWV * (S + 1) * 6 + (size_t)wave * (24 * 65);
  auto flush_group = [&](int s_first) {   // the columns of steps s_first, s_first - 1, s_first - 2, s_first - 3
    const int col = lane % 24, half = lane / 24;
    const int j = col / 6, e = col - j * 6;
    const int s2c = s_first - j;
    double psum = 0.0;
    if (lane < 48 && s2c >= 0) {
      const double* src = Tw + col * 65 + half * 32;
#pragma unroll 8
      for (int l = 0; l < 32; ++l) psum += src[l];
    }
    const double other = __shfl_down(psum, 24, RATO_WAVE);
    if (lane < 24 && s2c >= 0) gen_red[(wave * (S + 1) + s2c) * 6 + e] = psum + other;
  };
#endif
  double mP0[3], mP1[3], mV0[3], mV1[3], dP[3], dV[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    mP0[a] = 1.0; mP1[a] = 0.0; mV0[a] = 0.0; mV1[a] = 1.0; dP[a] = 0.0; dV[a] = 0.0;
  }
  const float* a22r = A22 + (size_t)(S - 1) * row3;   // row 3 s of the table being read back (stops at row 0)
  // A lane past the batch reads NOTHING back: its clamped address is the last sample's entry, which ANOTHER wave may not
  // have written yet when this wave is entirely past the batch (M = 5: waves 1..3; M = 1e5: the last wave of the last
  // block) -- whatever the buffer held before would then enter the adjoint, and a NaN bit pattern of a fresh allocation
  // survives the multiplication by dtm_v = 0 below (found by tools/soak.py in round 6: NaN sample sums on the second call
  // at S = 64, M = 5; with the persistent buffers of the SCP the stale entries were finite and the product zero).
  auto load_e22 = [&](float (&e22b)[TB][3]) {   // the table entries of the next 8 steps down (same lane wrote them: program order)
#pragma unroll
    for (int i = 0; i < TB; ++i) {
      e22b[i][0] = valid ? at(a22r, boff0) : 0.0f;
      e22b[i][1] = valid ? at(a22r, boff1) : 0.0f;
      e22b[i][2] = valid ? at(a22r, boff2) : 0.0f;
      a22r = (a22r != A22) ? a22r - row3 : a22r;
    }
  };
  const double dtm_v = valid ? dtm : 0.0;   // a lane past the batch contributes zeros to every sum (no select per number)
  auto backward_steps = [&](const float (&e22b)[TB][3], int sb, auto guarded) {
#pragma unroll
    for (int i = 0; i < TB; ++i) {
    const int s2 = sb - i;
    if (!decltype(guarded)::value || s2 >= 0) {
    double eP[3], eV[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      eP[a] = mP1[a] * dtm_v;
      eV[a] = mV1[a] * dtm_v;
      const double ua = (double)us[s2 * 3 + a];
      dP[a] += eP[a] * ua;
      dV[a] += eV[a] * ua;
    }
#if RATO_GEN_LDS_REDUCE
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      Tw[((i & 3) * 6 + a) * 65 + lane] = eP[a];
      Tw[((i & 3) * 6 + 3 + a) * 65 + lane] = eV[a];
    }
#else
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const double sp = rato::wave_sum_dpp(eP[a]);
      const double sv = rato::wave_sum_dpp(eV[a]);
      if (lane == 0) {
        gen_red[(wave * (S + 1) + s2) * 6 + a] = sp;
        gen_red[(wave * (S + 1) + s2) * 6 + 3 + a] = sv;
      }
    }
#endif
    if (s2 > 0) {  // mu_s = mu_{s+1} A_s
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        const double a22 = 1.0 - (double)e22b[i][a];
        const double nP0 = mP0[a] + mP1[a] * a21, nP1 = mP0[a] * dt + mP1[a] * a22;
        const double nV0 = mV0[a] + mV1[a] * a21, nV1 = mV0[a] * dt + mV1[a] * a22;
        mP0[a] = nP0; mP1[a] = nP1; mV0[a] = nV0; mV1[a] = nV1;
      }
    }
    }
#if RATO_GEN_LDS_REDUCE
    if ((i & 3) == 3 && sb - (i - 3) >= 0) flush_group(sb - (i - 3));   // (wave-uniform)
#endif
    }   // step s2 of the batch
  };
  {
    float ea[TB][3], eb[TB][3];
    load_e22(ea);
    for (int sb = S - 1; sb >= 0; sb -= 2 * TB) {
      load_e22(eb);
      backward_steps(ea, sb, std::true_type{});
      load_e22(ea);
      backward_steps(eb, sb - TB, std::true_type{});
    }
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const double rp = rato::wave_sum_dpp(valid ? (-(p[a] - P.x_final64[a]) + dP[a]) : 0.0);
    const double rv = rato::wave_sum_dpp(valid ? (-(v[a] - P.x_final64[3 + a]) + dV[a]) : 0.0);
    if (lane == 0) {
      gen_red[(wave * (S + 1) + S) * 6 + a] = rp;
      gen_red[(wave * (S + 1) + S) * 6 + 3 + a] = rv;
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 6 * S + 6; i += RATO_BLOCK) {   // [s*6 + e] and, last, the 6 rhs entries
    double acc = 0.0;
#pragma unroll
    for (int w = 0; w < NWV; ++w) acc += gen_red[w * (S + 1) * 6 + i];   // fixed order
    part[(size_t)blockIdx.x * (6 * S + 6) + i] = (float)acc;
  }
}

__global__ __launch_bounds__(RATO_BLOCK) void drone_obstacle_kernel(rato_drone_params P,
                                                                    const float* __restrict__ xs,
                                                                    const float* __restrict__ Qsym,
                                                                    float* __restrict__ g) {
  const size_t M = (size_t)P.M, ld = (size_t)P.ld;
  const size_t m = (size_t)blockIdx.x * RATO_BLOCK + threadIdx.x;
  const int t = blockIdx.y;
  if (m >= M) return;
  const float px = xs[((size_t)(t + 1) * 6 + 0) * ld + m];
  const float py = xs[((size_t)(t + 1) * 6 + 1) * ld + m];
#pragma unroll
  for (int j = 0; j < NOBS; ++j) {
    const float q00 = Qsym[(size_t)(j * 3 + 0) * ld + m], qs = Qsym[(size_t)(j * 3 + 1) * ld + m],
                q11 = Qsym[(size_t)(j * 3 + 2) * ld + m];
    const float dx = px - P.obs_xy[j][0], dy = py - P.obs_xy[j][1];
    g[((size_t)j * P.S + t) * ld + m] = 1.0f - (q00 * dx * dx + qs * dx * dy + q11 * dy * dy);
  }
}

// Column owned by slot k of column-group grp (serpentine, so that the
// triangular work S-1-s is balanced over groups).  >= S means "no column".
__device__ __forceinline__ int column_of(int k, int grp, int ngroups) {
  return k * ngroups + ((k & 1) ? (ngroups - 1 - grp) : grp);
}

template <int N>
struct Vec;
template <>
struct Vec<1> {
  typedef float type;
};
template <>
struct Vec<2> {
  typedef float type __attribute__((ext_vector_type(2)));
};
template <>
struct Vec<4> {
  typedef float type __attribute__((ext_vector_type(4)));
};

template <int N>
__device__ __forceinline__ void vload(const float* __restrict__ src, float (&x)[N]) {
  if constexpr (N == 1) {
    x[0] = *src;
  } else {
    const typename Vec<N>::type v = *reinterpret_cast<const typename Vec<N>::type*>(src);
#pragma unroll
    for (int i = 0; i < N; ++i) x[i] = v[i];
  }
}
template <int N>
__device__ __forceinline__ void vstore(float* __restrict__ dst, const float (&x)[N]) {
  if constexpr (N == 1) {
    *dst = x[0];
  } else {
    typename Vec<N>::type v;
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = x[i];
    *reinterpret_cast<typename Vec<N>::type*>(dst) = v;
  }
}

// CPT control columns and SPL consecutive samples per lane.  SPL > 1 turns every
// global access into an 8/16-byte-per-lane vector access and lets the per-sample
// arithmetic pair up into packed fp32 instructions (the SPL=1 form spends ~15
// wave-instructions per stored dword and is issue-bound, not HBM-bound).
template <int CPT, int SPL>
__global__ __launch_bounds__(RATO_BLOCK) void drone_linearize_kernel(
    rato_drone_params P, const float* __restrict__ us, const float* __restrict__ dW,
    const float* __restrict__ mass, const float* __restrict__ Qsym, float* __restrict__ G,
    float* __restrict__ g_up, float* __restrict__ Z, float* __restrict__ part) {
  const size_t M = (size_t)P.M, ld = (size_t)P.ld;
  const int S = P.S;
  const size_t m_raw = ((size_t)blockIdx.x * RATO_BLOCK + threadIdx.x) * SPL;  // first sample of this lane
  const bool any_valid = m_raw < M;
  // clamp loads to the last in-range lane group; stores are predicated on any_valid
  const size_t m0 = any_valid ? m_raw : ((M - 1) / SPL) * SPL;
  bool vld[SPL];
#pragma unroll
  for (int i = 0; i < SPL; ++i) vld[i] = any_valid && (m0 + i < M);
  const int grp = blockIdx.y, ngroups = gridDim.y;
  const bool lead = (grp == 0);  // group 0 also emits g_up, Z and the rhs partial

  // per-sample constants
  float inv_m[SPL], a21[SPL], cn[SPL], dtm[SPL];
  float q00[NOBS][SPL], qs[NOBS][SPL], q11[NOBS][SPL];
  {
    float ms[SPL];
    vload<SPL>(mass + m0, ms);
    const float sdt_beta = sqrtf(P.dt) * P.beta;
#pragma unroll
    for (int i = 0; i < SPL; ++i) {
      inv_m[i] = 1.0f / ms[i];
      a21[i] = -P.kp * P.dt * inv_m[i];
      cn[i] = sdt_beta * inv_m[i];
      dtm[i] = P.dt * inv_m[i];
    }
#pragma unroll
    for (int j = 0; j < NOBS; ++j) {
      vload<SPL>(Qsym + (size_t)(j * 3 + 0) * ld + m0, q00[j]);
      vload<SPL>(Qsym + (size_t)(j * 3 + 1) * ld + m0, qs[j]);
      vload<SPL>(Qsym + (size_t)(j * 3 + 2) * ld + m0, q11[j]);
    }
  }
  // this lane's slot in the tile-blocked Jacobian: tile = m0 / TILE, lane offset m0 % TILE
  const size_t tile_floats = rato::packed_tile_stride((size_t)rato::pair_row_offset(S) * 2 * NOBS * RATO_TILE);
  float* __restrict__ Gt = G + (m0 / RATO_TILE) * tile_floats + (m0 % RATO_TILE);

  int col[CPT];
#pragma unroll
  for (int k = 0; k < CPT; ++k) {
    const int s = column_of(k, grp, ngroups);
    col[k] = (s < S) ? s : 0x7fffffff;
  }
  // Phi[k][a][i] = (P, V) = d(p_a, v_a)_t / d u_{col[k], a} of sample i
  float phiP[CPT][3][SPL], phiV[CPT][3][SPL];
#pragma unroll
  for (int k = 0; k < CPT; ++k)
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int i = 0; i < SPL; ++i) phiP[k][a][i] = phiV[k][a][i] = 0.0f;

  float p[3][SPL], v[3][SPL], dp[3][SPL], dv[3][SPL];  // state, and its tangent along u (g_up / rhs)
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int i = 0; i < SPL; ++i) {
      p[a][i] = P.x_init[a];
      v[a][i] = P.x_init[3 + a];
      dp[a][i] = dv[a][i] = 0.0f;
    }
  float zmax[SPL];
#pragma unroll
  for (int i = 0; i < SPL; ++i) zmax[i] = -INFINITY;
  float xi[3][SPL];
#pragma unroll
  for (int a = 0; a < 3; ++a) vload<SPL>(dW + (size_t)a * ld + m0, xi[a]);

  for (int t = 0; t < S; ++t) {
    float nxt[3][SPL];
    const int tn = (t + 1 < S) ? t + 1 : t;  // prefetch next step's noise
#pragma unroll
    for (int a = 0; a < 3; ++a) vload<SPL>(dW + (size_t)(tn * 3 + a) * ld + m0, nxt[a]);

    float a22[3][SPL];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int i = 0; i < SPL; ++i)
        a22[a][i] = 1.0f - P.dt * (P.kd + 2.0f * P.drag * fabsf(v[a][i])) * inv_m[i];

    if (lead) {  // tangent along u: d x_{t+1} = A_t d x_t + B u_t
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        const float u = us[t * 3 + a];
#pragma unroll
        for (int i = 0; i < SPL; ++i) {
          const float dpn = dp[a][i] + P.dt * dv[a][i];
          const float dvn = a21[i] * dp[a][i] + a22[a][i] * dv[a][i] + dtm[i] * u;
          dp[a][i] = dpn;
          dv[a][i] = dvn;
        }
      }
    }
    // column sensitivities (columns not yet active hold exact zeros and are skipped)
#pragma unroll
    for (int k = 0; k < CPT; ++k) {
      if (col[k] <= t) {  // wave-uniform
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
          for (int i = 0; i < SPL; ++i) {
            const float Pn = phiP[k][a][i] + P.dt * phiV[k][a][i];
            const float Vn = a21[i] * phiP[k][a][i] + a22[a][i] * phiV[k][a][i];
            phiP[k][a][i] = Pn;
            phiV[k][a][i] = (col[k] == t) ? dtm[i] : Vn;  // Phi_{t+1,t} = B_t
          }
      }
    }
    // state (drone_risk.py:122-131,148-153)
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const float u = us[t * 3 + a];
#pragma unroll
      for (int i = 0; i < SPL; ++i) {
        const float acc = (u - (P.kp * p[a][i] + P.kd * v[a][i])) * inv_m[i] -
                          P.drag * fabsf(v[a][i]) * v[a][i] * inv_m[i];
        const float pn = p[a][i] + P.dt * v[a][i];
        const float vn = v[a][i] + P.dt * acc + cn[i] * xi[a][i];
        p[a][i] = pn;
        v[a][i] = vn;
      }
    }
    // constraint row t (uses p_{t+1}) and its gradient w = -(Q+Q^T) d
    float wx[NOBS][SPL], wy[NOBS][SPL];
#pragma unroll
    for (int j = 0; j < NOBS; ++j) {
      float gu[SPL];
#pragma unroll
      for (int i = 0; i < SPL; ++i) {
        const float dx = p[0][i] - P.obs_xy[j][0], dy = p[1][i] - P.obs_xy[j][1];
        const float gj = 1.0f - (q00[j][i] * dx * dx + qs[j][i] * dx * dy + q11[j][i] * dy * dy);
        wx[j][i] = -(2.0f * q00[j][i] * dx + qs[j][i] * dy);
        wy[j][i] = -(qs[j][i] * dx + 2.0f * q11[j][i] * dy);
        zmax[i] = fmaxf(zmax[i], gj);
        gu[i] = P.rows_out ? gj : (-gj + wx[j][i] * dp[0][i] + wy[j][i] * dp[1][i]);
      }
      if (lead && any_valid) vstore<SPL>(g_up + ((size_t)j * S + t) * ld + m0, gu);
    }
    const int row = rato::pair_row_offset(t);
#pragma unroll
    for (int k = 0; k < CPT; ++k) {
      if (col[k] < t) {  // wave-uniform
        // tile-blocked SoA: row = (pair*2 + axis)*NOBS + j; 32-bit offsets from the lane's tile slot
        const int base = ((row + col[k]) * 2 * NOBS) * RATO_TILE;
        if (any_valid) {
#pragma unroll
          for (int j = 0; j < NOBS; ++j) {
            float ox[SPL], oy[SPL];
#pragma unroll
            for (int i = 0; i < SPL; ++i) {
              ox[i] = wx[j][i] * phiP[k][0][i];
              oy[i] = wy[j][i] * phiP[k][1][i];
            }
            vstore<SPL>(Gt + base + j * RATO_TILE, ox);
            vstore<SPL>(Gt + base + (NOBS + j) * RATO_TILE, oy);
          }
        }
      }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int i = 0; i < SPL; ++i) xi[a][i] = nxt[a][i];
  }

  if (lead && any_valid && Z) {
    float zo[SPL];
#pragma unroll
    for (int i = 0; i < SPL; ++i) zo[i] = zmax[i] - P.tol;
    vstore<SPL>(Z + m0, zo);
  }

  // per-block sums for the sample mean of the final-constraint linearization
  __shared__ float red[RATO_BLOCK / RATO_WAVE][CPT * 6 + 6];
  const int lane = threadIdx.x & (RATO_WAVE - 1), wave = threadIdx.x / RATO_WAVE;
#pragma unroll
  for (int k = 0; k < CPT; ++k) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      float lp = 0.0f, lv = 0.0f;
#pragma unroll
      for (int i = 0; i < SPL; ++i) {
        lp += vld[i] ? phiP[k][a][i] : 0.0f;
        lv += vld[i] ? phiV[k][a][i] : 0.0f;
      }
      const float sp = rato::wave_sum(lp);
      const float sv = rato::wave_sum(lv);
      if (lane == 0) {
        red[wave][k * 6 + a] = sp;
        red[wave][k * 6 + 3 + a] = sv;
      }
    }
  }
  if (lead) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      // val_final = -(x_S - x_final) + v_final_du . u   (drone_risk.py:271)
      float lp = 0.0f, lv = 0.0f;
#pragma unroll
      for (int i = 0; i < SPL; ++i) {
        lp += vld[i] ? (-(p[a][i] - P.x_final[a]) + dp[a][i]) : 0.0f;
        lv += vld[i] ? (-(v[a][i] - P.x_final[3 + a]) + dv[a][i]) : 0.0f;
      }
      const float rp = rato::wave_sum(lp);
      const float rv = rato::wave_sum(lv);
      if (lane == 0) {
        red[wave][CPT * 6 + a] = rp;
        red[wave][CPT * 6 + 3 + a] = rv;
      }
    }
  }
  __syncthreads();
  const int tid = threadIdx.x;
  if (tid < CPT * 6) {
    const int k = tid / 6, e = tid % 6;
    const int s = column_of(k, grp, ngroups);
    if (s < S) {
      float acc = 0.0f;
#pragma unroll
      for (int w = 0; w < RATO_BLOCK / RATO_WAVE; ++w) acc += red[w][tid];
      part[(size_t)blockIdx.x * (6 * S + 6) + s * 6 + e] = acc;
    }
  } else if (lead && tid < CPT * 6 + 6) {
    float acc = 0.0f;
#pragma unroll
    for (int w = 0; w < RATO_BLOCK / RATO_WAVE; ++w) acc += red[w][tid];
    part[(size_t)blockIdx.x * (6 * S + 6) + 6 * S + (tid - CPT * 6)] = acc;
  }
}

// ---------------------------------------------------------------------------
// Row-parallel (adjoint) linearization — the default drone linearize kernel.
//
// The forward/column form above needs 6 registers per control column and
// re-rolls the trajectory once per column group; it spends ~15 wave-
// instructions per stored dword.  Here a workgroup owns 64 samples (one per
// lane) and 8 waves:
//   phase 0  all waves stage the block's noise tile dW[S][3][64] into LDS with
//            every load in flight at once (a per-step global load would queue
//            behind the chip-wide store stream for microseconds per step);
//   phase 1  waves 0,1,2 roll out ONE AXIS each (the axes decouple:
//            drone_risk.py:122-131) and leave a22_t (the only state-dependent
//            entry of the step Jacobian) and p_{t+1} in LDS: 20 B per sample-step,
//            publishing their progress in LDS after every step; each then runs the
//            final-state adjoint of its own axis and joins phase 2;
//   phase 2  (overlapped with phase 1: row t only needs steps 0..t, rows are
//            taken shortest first and wait on the published progress)
//            all waves pull tasks from an LDS work queue.  Row task t rebuilds
//            g_j(t), w = -(Q+Q^T) d from p_{t+1} and sweeps the row with the adjoint
//              mu_{t+1} = e_0^T,  mu_k = mu_{k+1} A_k,  d p_{t+1}/d u_s = mu_{s+1}[1] dt/m
//            (A_k = [[1, dt], [a21, a22_k]], B = [0, dt/m]^T): 3 packed FMAs + 6
//            multiplies per 6 stored dwords, ~40 VGPRs, one contiguous descending
//            1.5 KB-per-step store stream per wave.  The sweep also accumulates the
//            row's dot product with u, so g_up = -g + (grad g).u is exactly the
//            reference's expression (drone_risk.py:278).  One extra task runs the
//            adjoint from t = S for the final-state Jacobian (2 rows x 3 axes),
//            reduces it over the block's samples and forms the rhs (:271).
#ifndef RATO_ROWS_NW
#define RATO_ROWS_NW 8
#endif
#ifndef RATO_ROWS_MINW
#define RATO_ROWS_MINW 1
#endif
constexpr int ROWS_NW = RATO_ROWS_NW;  // waves per workgroup
constexpr int ROWS_SAMPLES = 64;  // samples per workgroup (one per lane)

typedef float float2_t __attribute__((ext_vector_type(2)));



#ifndef RATO_DIAG
#define RATO_DIAG 0  // diagnostic builds only (tools/): 1 = no phase 2, 2 = no phase 1, 3 = phase 2 without G stores,
#endif               // 4 = timeline: part[tile][0..7] <- wall_clock64 at block start / after phase 0 / after phase 1 / end

__host__ __device__ inline size_t rows_lds_floats(int S) {
  // A2 (2) + PP (2) + AZ (1) per (t, lane) | US float2[S] + uz[S] | queue head + rollout progress (+pad)
  return (size_t)S * ROWS_SAMPLES * 5 + (size_t)S * 3 + 4 + (RATO_DIAG == 4 ? 4 : 0);
}

// 1-D grid: workgroups [0, n_whole) own one whole tile each; after them every remaining tile is dealt out to `split`
// workgroups that each rebuild the LDS tables (noise staging + rollout: latency, almost no bandwidth) and take the row
// tasks congruent to their part (mod split).  Small batches (fewer tiles than workgroup slots) split every tile to
// fill the chip.  (Large batches CAN split the tiles of the last round into smaller work units to shorten the drain
// at the end of the launch -- RATO_TAIL_SPLIT / RATO_TAIL_PCT -- but that was measured to cost more than it returns,
// see the launch code; off by default.)
// FACT: factored output.  d g[j,t] / d u[s,a] = W[j,t,a] * Phi[t,s,a] with W = -(Q+Q^T)(p_{t+1} - o_j) (the
// gradient of g wrt position) and Phi = d p_{t+1,a} / d u_{s,a} SHARED by the three obstacles, so the same
// information is S(S-1) + 6S numbers per sample instead of 3S(S-1): 2.67x less HBM traffic at S = 50, for
// this kernel and for every consumer that reads the Jacobian (rowmax / tail-rows oracle, CSC emission).
// PHILOX: the tile's noise is REGENERATED while staging (Philox4x32-10 at counter (m, t), the numbers rato_drone_sample
// would have written) instead of read: no noise array, no reads mixed into the store stream.
template <bool FACT, bool PHILOX = false>
__global__ __launch_bounds__(ROWS_NW* RATO_WAVE, RATO_ROWS_MINW) void drone_linearize_rows_kernel(
    rato_drone_params P, int n_whole, int split, int tile_stride, int n_tiles_total, unsigned* __restrict__ tile_queue,
    uint64_t seed, float noise_scale, const float* __restrict__ us, const float* __restrict__ dW,
    const float* __restrict__ mass, const float* __restrict__ Qsym, float* __restrict__ G,
    float* __restrict__ W, float* __restrict__ A22, float* __restrict__ g_up, float* __restrict__ Z,
    float* __restrict__ part, const rato_sel::StatsTail tail, int flags) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
#if RATO_DIAG == 4
  const unsigned long long tl_kernel_in = wall_clock64();   // the workgroup's first instruction
  int tl_last_tile = -1;
#endif
  // Statistics in the same launch (rato_saa.h: params.stats_*): the workgroups behind the producer's own wait until every
  // tile's Z has been counted in, then run the exact selection on it -- beside the workgroups still storing the Jacobian.
  const int n_prod = tail.ws ? tail.n_prod : (int)gridDim.x;
  if (tail.is_stats((int)blockIdx.x)) {
    rato_sel::stats_tail_run<ROWS_NW * RATO_WAVE>(tail, Z, (long)P.M, lds_raw);
    return;
  }
  unsigned* const z_signal = tail.ws ? tail.ws->sig : nullptr;
  const bool noise_tiled = (flags & 1) != 0;   // dW is the re-tiled copy (rato_drone_tile_noise)
  const bool nt_stores = (flags & 2) != 0;     // the Jacobian goes out as streaming (non-temporal) stores
  const size_t M = (size_t)P.M, ld = (size_t)P.ld;
  const int S = P.S;
  const int lane = threadIdx.x & (RATO_WAVE - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / RATO_WAVE);  // scalar: wave-level branches stay scalar
  // Before step t of the rollout, PP[t] / AZ[t] hold that step's noise (xi_x, xi_y) / xi_z; the
  // axis wave that consumes a component overwrites it with its own output (same lane, so program
  // order is enough).  No __restrict__ on these pointers for that reason.
  float2_t* A2 = reinterpret_cast<float2_t*>(lds_raw);              // [S][64] (a22x, a22y)
  float2_t* PP = A2 + (size_t)S * ROWS_SAMPLES;                     // [S][64] p_{t+1} (x, y)
  float* AZ = reinterpret_cast<float*>(PP + (size_t)S * ROWS_SAMPLES);  // [S][64] a22z
  float2_t* US = reinterpret_cast<float2_t*>(AZ + (size_t)S * ROWS_SAMPLES);  // [S] (ux, uy)
  float* UZ = reinterpret_cast<float*>(US + S);                     // [S]
  int* head = reinterpret_cast<int*>(UZ + S);                       // [0] task queue, [1..2] rollout progress x, y

  const int bid = blockIdx.x;
  // A work UNIT is a whole tile (units [0, n_whole)) or one of `split` row-interleaved parts of a tile (the units after
  // them).  One unit per workgroup (plain grid), or several:
  //   tile_queue != NULL  DYNAMIC: the grid fills every workgroup slot once; a workgroup that has finished a unit
  //                       takes the next one from a global counter (one returning atomic per unit).  Workgroups on
  //                       XCDs / CUs that happen to run faster simply take more units, so the whole chip finishes
  //                       together (with one tile per workgroup the hardware deals the grid out to the XCDs up front
  //                       and four of them sat idle for the last ~13 % of the launch: DESIGN.md 4.1).
  //   tile_stride > 0     static: units bid, bid + stride, ... (A/B only)
  const int n_units = n_whole + (n_tiles_total - n_whole) * split;
  int* next_tile = head + 3;   // LDS word: the unit thread 0 has fetched for this workgroup
  for (int unit = bid; unit < n_units;) {
  const bool whole = unit < n_whole;
  const int tile = whole ? unit : n_whole + (unit - n_whole) / split;
  const int part_id = whole ? 0 : (unit - n_whole) % split, row_split = whole ? 1 : split;
  if (unit != bid && !tile_queue) __syncthreads();   // the previous tile's tables are dead (dynamic: synced below)
#if RATO_DIAG == 4
  unsigned long long tl0 = wall_clock64(), tl1 = 0, tl2 = 0;
  // LDS word behind the queue state: the clock at which the first row WITH stores (task 1 of this part) was picked
  unsigned long long* tl_first = reinterpret_cast<unsigned long long*>(head + 4);
#endif
  const size_t m_raw = (size_t)tile * ROWS_SAMPLES + lane;
  const bool valid = m_raw < M;
  const size_t m = valid ? m_raw : M - 1;  // clamp loads; stores are predicated
  const float inv_m = 1.0f / mass[m];
  const float a21 = -P.kp * P.dt * inv_m;
  const float dtm = P.dt * inv_m;
  // every wave needs its samples' obstacle matrices (row tasks rebuild g and w = -(Q+Q^T) d)
  float q00[NOBS], qs[NOBS], q11[NOBS];
#pragma unroll
  for (int j = 0; j < NOBS; ++j) {
    q00[j] = Qsym[(size_t)(j * 3 + 0) * ld + m];
    qs[j] = Qsym[(size_t)(j * 3 + 1) * ld + m];
    q11[j] = Qsym[(size_t)(j * 3 + 2) * ld + m];
  }

  // ---- phase 0: stage dW[t][a][64 samples], the controls and the queue state
  {
    float* PPf = reinterpret_cast<float*>(PP);
    const int nrows = 3 * S;
    constexpr int MAXR = (160 + ROWS_NW - 1) / ROWS_NW;  // rows per wave per batch: S = 50 needs 150 / 8 = 19 -> one batch, all loads in flight
    if (PHILOX) {
      for (int t = wave; t < S; t += ROWS_NW) {
        const rato::u32x4 rr = rato::philox_at(seed, rato::PHILOX_STREAM_DW, (uint32_t)t, (uint64_t)m);
        float x0, x1, x2, x3;
        rato::box_muller(rr.x, rr.y, x0, x1);
        rato::box_muller(rr.z, rr.w, x2, x3);
        PPf[(t * ROWS_SAMPLES + lane) * 2 + 0] = x0 * noise_scale;
        PPf[(t * ROWS_SAMPLES + lane) * 2 + 1] = x1 * noise_scale;
        AZ[t * ROWS_SAMPLES + lane] = x2 * noise_scale;
      }
    }
    if (!PHILOX && noise_tiled) {
      // the tile's block IS the LDS image (PP: [S][64] (xi_x, xi_y) pairs, then AZ: [S][64] xi_z): a straight copy, 16 bytes
      // per lane and request, every request of the thread in flight before the first one is stored
      typedef float rfloat4_t __attribute__((ext_vector_type(4)));
      const rfloat4_t* __restrict__ src4 = reinterpret_cast<const rfloat4_t*>(dW + (size_t)tile * nrows * ROWS_SAMPLES);
      rfloat4_t* dst4 = reinterpret_cast<rfloat4_t*>(PPf);
      const int n4 = nrows * (ROWS_SAMPLES / 4);
      constexpr int MAX4 = 5, NT = ROWS_NW * RATO_WAVE;   // S = 50: 2400 requests over 512 threads
      for (int b0 = threadIdx.x; b0 < n4; b0 += NT * MAX4) {
        rfloat4_t tmp4[MAX4];
#pragma unroll
        for (int i = 0; i < MAX4; ++i) {
          const int idx = b0 + i * NT;
          tmp4[i] = src4[idx < n4 ? idx : 0];
        }
#pragma unroll
        for (int i = 0; i < MAX4; ++i) {
          const int idx = b0 + i * NT;
          if (idx < n4) dst4[idx] = tmp4[i];
        }
      }
    }
    for (int r0 = wave; !PHILOX && !noise_tiled && r0 < nrows; r0 += ROWS_NW * MAXR) {
      float tmp[MAXR];
#pragma unroll
      for (int i = 0; i < MAXR; ++i) {
        const int r = r0 + i * ROWS_NW;
#if RATO_DIAG == 5   // diagnostic: no noise reads at all (a cheap hash instead): what do the reads cost beside the stores?
        tmp[i] = 0.01f * (float)((int)((r * 2654435761u + (unsigned)m * 40503u) >> 20) - 2048) * (1.0f / 2048.0f);
#else
        tmp[i] = (r < nrows) ? dW[(size_t)r * ld + m] : 0.0f;
#endif
      }
#pragma unroll
      for (int i = 0; i < MAXR; ++i) {
        const int r = r0 + i * ROWS_NW;
        if (r < nrows) {
          const int t = r / 3, a = r - 3 * t;
          if (a < 2) PPf[(t * ROWS_SAMPLES + lane) * 2 + a] = tmp[i];
          else AZ[t * ROWS_SAMPLES + lane] = tmp[i];
        }
      }
    }
    for (int i = threadIdx.x; i < S; i += ROWS_NW * RATO_WAVE) {
      float2_t u2;
      u2.x = us[i * 3 + 0];
      u2.y = us[i * 3 + 1];
      US[i] = u2;
      UZ[i] = us[i * 3 + 2];
    }
    if (threadIdx.x == 0) {
      head[0] = 0;
      head[1] = head[2] = (RATO_DIAG == 2) ? S : 0;   // rollout progress of the x and y axes
    }
  }
  __syncthreads();
#if RATO_DIAG == 4
  tl1 = wall_clock64();
#endif

  // ---- phase 1 (waves 0..2) overlapped with phase 2 (the other waves, then everybody).
  // Wave a < 3 rolls out axis a, publishing its progress after every step (x and y: prog[a] = steps done),
  // then runs the final-state adjoint of its own axis and joins the row queue.  Row task t only needs
  // steps 0..t, so the row waves start sweeping (shortest rows first) while the rollout is still running:
  // a workgroup starts storing ~1 step after its noise tile has landed instead of after S sequential steps.
  typedef __attribute__((address_space(3))) volatile int lds_vint;   // keeps the polls / publishes ds_ instructions
  lds_vint* prog = (lds_vint*)(head + 1);
  constexpr int RT = ROWS_SAMPLES;  // tile width: each row sweep below is one contiguous descending stream
  constexpr int RPP = FACT ? 2 : 2 * NOBS;  // tile rows per (t, s) pair
  const size_t tile_floats = rato::packed_tile_stride((size_t)rato::pair_row_offset(S) * RPP * RT);
  float* __restrict__ Gt = G + (size_t)tile * tile_floats + lane;

  auto wait_steps = [&](int need) {  // both horizontal axes rolled out through step need-1
    while (true) {
      const int p0 = prog[0], p1 = prog[1];
      if ((p0 < p1 ? p0 : p1) >= need) break;
      __builtin_amdgcn_s_sleep(4);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  };
  // Z = max_{j,t} g - tol from the p_{t+1} table (drone_risk.py:656-662), computed by the z-axis wave as soon as both
  // horizontal axes have been rolled out -- BEFORE the rows are swept, so that the statistics workgroups of this launch
  // (params.stats_*) can select on Z while the Jacobian is still being stored.  (Until round 4 this was the LAST task of
  // the row queue.)  Row-split parts: the part that owned that task writes.
  auto write_Z = [&]() {
    wait_steps(S);
    float zmax = -INFINITY;
    for (int t = 0; t < S; ++t) {
      const float2_t pp = PP[t * ROWS_SAMPLES + lane];
#pragma unroll
      for (int j = 0; j < NOBS; ++j) {
        const float dx = pp.x - P.obs_xy[j][0], dy = pp.y - P.obs_xy[j][1];
        zmax = fmaxf(zmax, 1.0f - (q00[j] * dx * dx + qs[j] * dx * dy + q11[j] * dy * dy));
      }
    }
    if (!z_signal) {
      if (valid) Z[m] = zmax - P.tol;
    } else {
      // Statistics in this launch: the workgroups behind the producer's are waiting for every tile's Z.  Z goes
      // out as agent-scope atomic stores (written through to the point of coherence of the device -- a release FENCE
      // here would write back this XCD's whole L2, in the middle of the Jacobian's store stream, once per tile), the
      // wave waits for them to complete, then counts its tile in; the tile that completes the count raises z_ready.
      if (valid)
        __hip_atomic_store(reinterpret_cast<unsigned*>(Z) + m, __float_as_uint(zmax - P.tol), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane == 0) {
        const unsigned c = __hip_atomic_fetch_add(z_signal + 0, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (c == (unsigned)n_tiles_total - 1u) {
          __hip_atomic_store(z_signal + 0, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(z_signal + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
    }
  };
  if (wave < 3) {
    const int a = wave;
    // all three axis waves run the same instruction stream: table offsets (in floats from the LDS base) and
    // strides instead of per-axis branches.  x/y: noise in PP[t].{x,y}, a22 -> A2[t].{x,y}, p -> PP[t].{x,y};
    // z: noise in AZ[t], a22 -> AZ[t].
    float* L = reinterpret_cast<float*>(lds_raw);
    const int offA2 = 0, offPP = 2 * S * ROWS_SAMPLES, offAZ = 4 * S * ROWS_SAMPLES, offUS = 5 * S * ROWS_SAMPLES,
              offUZ = offUS + 2 * S;
    const int st = (a < 2) ? 2 : 1;                                   // floats per (t, lane) slot
    const int o_noise = ((a < 2) ? offPP + a : offAZ) + lane * st;
    const int o_a22 = ((a < 2) ? offA2 + a : offAZ) + lane * st;
    const int o_u = (a < 2) ? offUS + a : offUZ;
    float* A2f = reinterpret_cast<float*>(A2);
    float p = P.x_init[a], v = P.x_init[3 + a];
    if (RATO_DIAG != 2) {
      const float cn = sqrtf(P.dt) * P.beta * inv_m;  // sqrt(dt) * (beta/m): drone_risk.py:136,151
      float xi = L[o_noise], u = L[o_u];
      for (int t = 0; t < S; ++t) {
        // next step's inputs are fetched BEFORE this step's table stores (same arrays, other slots)
        const int tn = (t + 1 < S) ? t + 1 : t;
        const float xi_n = L[o_noise + tn * ROWS_SAMPLES * st], u_n = L[o_u + tn * st];
        const float a22 = 1.0f - P.dt * (P.kd + 2.0f * P.drag * fabsf(v)) * inv_m;
        const float acc = (u - (P.kp * p + P.kd * v)) * inv_m - P.drag * fabsf(v) * v * inv_m;
        const float pn = p + P.dt * v;
        const float vn = v + P.dt * acc + cn * xi;
        p = pn;
        v = vn;
        L[o_a22 + t * ROWS_SAMPLES * st] = a22;
        if (a < 2) {
          L[o_noise + t * ROWS_SAMPLES * st] = p;
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
          if (lane == 0) prog[a] = t + 1;
        }
        xi = xi_n;
        u = u_n;
      }
    }
#if RATO_DIAG == 4
    tl2 = wall_clock64();
#endif
    if (a == 2 && Z && RATO_DIAG != 1 && (S % row_split) == part_id) write_Z();
    // d x_S / d u_s = A_{S-1} ... A_{s+1} B_s for this wave's axis: rows (P, V), summed over the block's
    // samples; the same sweep accumulates (d x_S / d u) . u for the rhs  (drone_risk.py:271)
    if (RATO_DIAG != 1 && (a % row_split) == part_id) {
      float mP0 = 1.0f, mP1 = 0.0f, mV0 = 0.0f, mV1 = 1.0f, dP = 0.0f, dV = 0.0f;
      for (int s2 = S - 1; s2 >= 0; --s2) {
        const float ua = (a == 0) ? US[s2].x : ((a == 1) ? US[s2].y : UZ[s2]);
        const float eP = mP1 * dtm, eV = mV1 * dtm;
        dP += eP * ua;
        dV += eV * ua;
        const float sp = rato::wave_sum_dpp(valid ? eP : 0.0f);
        const float sv = rato::wave_sum_dpp(valid ? eV : 0.0f);
        if (lane == 0) {
          part[(size_t)tile * (6 * S + 6) + s2 * 6 + a] = sp;
          part[(size_t)tile * (6 * S + 6) + s2 * 6 + 3 + a] = sv;
        }
        if (s2 > 0) {  // mu_s = mu_{s+1} A_s
          const int slot = s2 * ROWS_SAMPLES + lane;
          const float a22 = (a < 2) ? A2f[slot * 2 + a] : AZ[slot];
          const float nP0 = mP0 + mP1 * a21, nP1 = mP0 * P.dt + mP1 * a22;
          const float nV0 = mV0 + mV1 * a21, nV1 = mV0 * P.dt + mV1 * a22;
          mP0 = nP0; mP1 = nP1; mV0 = nV0; mV1 = nV1;
        }
      }
      const float rp = rato::wave_sum_dpp(valid ? (-(p - P.x_final[a]) + dP) : 0.0f);
      const float rv = rato::wave_sum_dpp(valid ? (-(v - P.x_final[3 + a]) + dV) : 0.0f);
      if (lane == 0) {
        part[(size_t)tile * (6 * S + 6) + 6 * S + a] = rp;
        part[(size_t)tile * (6 * S + 6) + 6 * S + 3 + a] = rv;
      }
    }
  }

  // ---- phase 2: tasks from the LDS queue, in ascending row order.
  // Partition p of row_split owns the tasks congruent to p (mod row_split).  Task T < S is row t = T;
  // task T = S is Z = max_{j,t} g - tol from the p_{t+1} table.
  // One LDS fetch-add per task, issued by lane 0 and broadcast (written without `continue`:
  // hipcc 7.2 mis-structured the earlier for(;;)/continue form into a loop that re-ran task 0).
  auto next_task = [&]() -> int {
    int v = 0;
    if (lane == 0) v = atomicAdd(head, 1);
    return part_id + row_split * __builtin_amdgcn_readfirstlane(v);
  };
  int task = (RATO_DIAG == 1) ? S + 1 : next_task();
  while (task < S) {
    {
      const int t = task;
      wait_steps(t + 1);
#if RATO_DIAG == 4
      if (t >= 1 && t <= row_split && lane == 0) *tl_first = wall_clock64();   // (the first task with a store of this part)
#endif
      const float2_t pp = PP[t * ROWS_SAMPLES + lane];
      float gj[NOBS], wx[NOBS], wy[NOBS];  // g, and -(Q+Q^T) d pre-multiplied by dt/m
#pragma unroll
      for (int j = 0; j < NOBS; ++j) {
        const float dx = pp.x - P.obs_xy[j][0], dy = pp.y - P.obs_xy[j][1];
        gj[j] = 1.0f - (q00[j] * dx * dx + qs[j] * dx * dy + q11[j] * dy * dy);
        wx[j] = -(2.0f * q00[j] * dx + qs[j] * dy) * dtm;
        wy[j] = -(qs[j] * dx + 2.0f * q11[j] * dy) * dtm;
      }
      if (FACT && valid) {
#pragma unroll
        for (int j = 0; j < NOBS; ++j) {   // W[j][t][a] = dg/dp (without the dt/m folded into wx, wy)
          const float dx = pp.x - P.obs_xy[j][0], dy = pp.y - P.obs_xy[j][1];
          W[(((size_t)j * S + t) * 2 + 0) * ld + m] = -(2.0f * q00[j] * dx + qs[j] * dy);
          W[(((size_t)j * S + t) * 2 + 1) * ld + m] = -(qs[j] * dx + 2.0f * q11[j] * dy);
        }
        if (A22) {                         // the step-Jacobian table itself (implicit consumers)
          const float2_t at = A2[t * ROWS_SAMPLES + lane];
          A22[((size_t)t * 2 + 0) * ld + m] = at.x;
          A22[((size_t)t * 2 + 1) * ld + m] = at.y;
        }
      }
      float m0x = 1.0f, m0y = 1.0f, m1x = 0.0f, m1y = 0.0f;  // mu_{t+1} = e_0^T (x and y axes)
      float accx = 0.0f, accy = 0.0f;                        // sum_s mu_{s+1}[1] u_s  per axis
      float* __restrict__ Grow = Gt + (size_t)rato::pair_row_offset(t) * (RPP * RT);
      for (int k = t; k >= 1; --k) {
        const float2_t aa = A2[k * ROWS_SAMPLES + lane];
        const float2_t u2 = US[k - 1];
        const float n0x = m0x + m1x * a21, n0y = m0y + m1y * a21;
        const float n1x = m0x * P.dt + m1x * aa.x, n1y = m0y * P.dt + m1y * aa.y;
        m0x = n0x; m0y = n0y; m1x = n1x; m1y = n1y;
        accx += m1x * u2.x;
        accy += m1y * u2.y;
        if (valid && (RATO_DIAG != 3 || m1x == 123.456f)) {
          float* __restrict__ o = Grow + (k - 1) * (RPP * RT);  // column s = k-1
          // The Jacobian is written once and never read back by this kernel.  STREAMING stores (nt_stores; chosen by the
          // launcher when the output is far beyond the 256 MB memory-side cache and the batch's inputs fit into it) do not
          // allocate there: the inputs then survive from one linearization to the next and the noise is not re-read from
          // HBM in the middle of the store stream (metric configuration -4.3 %, driving C5 shard -11.7 %)
          if (nt_stores) {
            static_assert(NOBS == 3, "the streaming stores are written out for three obstacles");
            if (FACT) {
              rato::store_streaming<0>(o, m1x * dtm);
              rato::store_streaming<RT * 4>(o, m1y * dtm);
            } else {
              rato::store_streaming<0 * RT * 4>(o, wx[0] * m1x);
              rato::store_streaming<1 * RT * 4>(o, wx[1] * m1x);
              rato::store_streaming<2 * RT * 4>(o, wx[2] * m1x);
              rato::store_streaming<3 * RT * 4>(o, wy[0] * m1y);
              rato::store_streaming<4 * RT * 4>(o, wy[1] * m1y);
              rato::store_streaming<5 * RT * 4>(o, wy[2] * m1y);
            }
          } else if (FACT) {
            o[0] = m1x * dtm;   // Phi[t, s, x] = d p_x(t+1) / d u_x(s)
            o[RT] = m1y * dtm;
          } else {
#pragma unroll
            for (int j = 0; j < NOBS; ++j) {
              o[j * RT] = wx[j] * m1x;
              o[(NOBS + j) * RT] = wy[j] * m1y;
            }
          }
        }
      }
      if (valid) {
#pragma unroll
        for (int j = 0; j < NOBS; ++j)  // g_up = -g + (grad g) . u   (drone_risk.py:278); rows_out = 1: g itself
          g_up[((size_t)j * S + t) * ld + m] = P.rows_out ? gj[j] : (-gj[j] + wx[j] * accx + wy[j] * accy);
      }
    }
    task = next_task();
  }
#if RATO_DIAG == 4
  __syncthreads();   // all of the block's row tasks issued (stores may still be in flight)
  const unsigned long long tl_issued = wall_clock64();
  if (flags & 4) {   // RATO_DIAG_WAIT=1: also the clock at which every store of the unit has been acknowledged
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    // (row-split parts of one tile share the tile's record: the part that ends last wins -- the timeline tool runs with
    //  whole tiles, RATO_DYN_TAIL_SPLIT=1, when it wants every unit)
    unsigned long long* tl = reinterpret_cast<unsigned long long*>(part + (size_t)tile * (6 * S + 6));
    unsigned xcc = 0;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    tl[0] = tl0; tl[1] = tl1; tl[2] = tl2; tl[3] = tl_issued;
    tl[4] = *tl_first;
    tl[6] = wall_clock64();
    tl[7] = tl_kernel_in;
    tl[8] = 0;
    tl_last_tile = tile;
    tl[5] = (unsigned long long)bid | ((unsigned long long)(xcc & 0xf) << 32) | ((unsigned long long)part_id << 40);
  }
#endif
  // ---- next tile
  if (tile_queue) {
    __syncthreads();   // every wave has finished this tile's rows: the tables are dead, head[] may be rewritten
    if (threadIdx.x == 0)
      next_tile[0] = n_prod + (int)__hip_atomic_fetch_add(tile_queue, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    unit = next_tile[0];
  } else if (tile_stride > 0) {
    unit += tile_stride;
  } else {
    break;
  }
  }  // tile loop
  if (tile_queue && threadIdx.x == 0) {
    // the queue cleans up after itself: the workgroup that leaves last (its own final fetch came back empty, like
    // everybody's before it) resets both words for the next launch
    const unsigned gone = __hip_atomic_fetch_add(tile_queue + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (gone == (unsigned)n_prod - 1u) {
      __hip_atomic_store(tile_queue, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(tile_queue + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
#if RATO_DIAG == 4
  if (threadIdx.x == 0 && tl_last_tile >= 0)   // the workgroup's last instruction, into the record of its last unit
    reinterpret_cast<unsigned long long*>(part + (size_t)tl_last_tile * (6 * P.S + 6))[8] = wall_clock64();
#endif
}


unsigned* resolve_tile_queues() {
  void* sym = nullptr;
  return hipGetSymbolAddress(&sym, HIP_SYMBOL(g_tile_queues)) == hipSuccess ? static_cast<unsigned*>(sym) : nullptr;
}
rato::TileQueuePool g_queue_pool;

unsigned* take_tile_queue(hipStream_t stream) { return g_queue_pool.take(stream, resolve_tile_queues); }

int device_cus() { return g_queue_pool.cus(); }

bool params_ok(const rato_drone_params* p) {
  return p && p->M > 0 && p->ld >= p->M && p->S > 0 && p->S <= 4096 && p->dt > 0.0f;
}
// the entry points that compute in fp64 read the double constants: they must have been filled (consistently)
bool params64_ok(const rato_drone_params* p) {
  return p->dt64 > 0.0 && fabs(p->dt64 - (double)p->dt) <= 1e-6 * p->dt64 && fabs(p->kp64 - (double)p->kp) <= 1e-6 * fabs(p->kp64) + 1e-30;
}

}  // namespace

namespace {
// rato_drone_eval without trajectories: the tiled kernel (one wave per 64 samples, noise batches in flight) at EVERY batch
// size -- small batches because it is bound by its own arithmetic instead of a load per step, large ones because it does
// not read the vertical axis' noise at all (8 instead of 12 bytes per sample-step: M = 1e7 1.115 -> 0.787 ms).
// RATO_EVAL_TILES_MAX_M overrides (A/B).
constexpr int64_t EVAL_TILES_MAX_M = 0x7fffffff, EVAL_STATS_IN_LAUNCH_MAX_M = 65536;
int eval_tiles_max_m() {
  static const int64_t v = [] { const char* e = getenv("RATO_EVAL_TILES_MAX_M"); return e ? (int64_t)atoll(e) : EVAL_TILES_MAX_M; }();
  return (int)v;
}
// the statistics of Z in extra workgroups of an eval launch with `grid` producer workgroups of NT threads
template <int NT>
int stats_tail_for(const void* workspace, double* out, double alpha, float thr, int64_t M, int grid, rato_sel::StatsTail& tail,
                   int& grid_launch, size_t& lds_launch) {
  int Gs = 0;
  const int extra = rato_sel::stats_tail_workgroups<NT>(M, Gs);
  if (extra < 0) return RATO_EINVAL;
  tail.ws = static_cast<rato_sel::Workspace*>(const_cast<void*>(workspace));
  tail.out = out;
  tail.alpha = alpha;
  tail.thr = thr;
  tail.G = Gs;
  tail.n_prod = grid;
  rato_sel::stats_rank(M, alpha, tail.k, tail.var_is_max);
  grid_launch = grid + extra;
  if (lds_launch < rato_sel::rs_body_lds_bytes<NT>()) lds_launch = rato_sel::rs_body_lds_bytes<NT>();
  return RATO_OK;
}
}  // namespace

// 1 when rato_drone_eval with params.stats_* AND RATO_STATS_IN_LAUNCH in params.stats_flags would compute the statistics in
// its own launch for this batch
extern "C" int rato_drone_eval_stats_in_launch(int32_t M) { return M > 0 && M <= EVAL_STATS_IN_LAUNCH_MAX_M && M <= eval_tiles_max_m(); }

namespace {
int drone_eval_tiles_launch(const rato_drone_params* p, int K, const float* us, const float* dW, const float* mass,
                            const float* Qsym, float* Z, int64_t ldz, float* g, const rato_sel::StatsTail& tail, int grid_launch,
                            size_t lds, int n_tiles, hipStream_t st) {
  if (g)
    hipLaunchKernelGGL(drone_eval_tiles_kernel<true>, dim3(grid_launch, K), dim3(RATO_BLOCK), lds, st, *p, us, dW, mass, Qsym, Z,
                       (long)ldz, g, n_tiles, tail);
  else
    hipLaunchKernelGGL(drone_eval_tiles_kernel<false>, dim3(grid_launch, K), dim3(RATO_BLOCK), lds, st, *p, us, dW, mass, Qsym, Z,
                       (long)ldz, g, n_tiles, tail);
  RATO_LAUNCH_CHECK();
  return RATO_OK;
}
}  // namespace

extern "C" int rato_drone_eval(const rato_drone_params* p, const float* us, const float* dW, const float* mass,
                               const float* Qsym, float* Z, float* xs, float* g, void* stream) {
  RATO_CLEAR_ERROR();
  if (!params_ok(p) || !us || !dW || !mass || !Qsym) return RATO_EINVAL;
  if (p->stats_workspace && (!Z || !p->stats_out || !(p->stats_alpha > 0.0) || !(p->stats_alpha <= 1.0))) return RATO_EINVAL;
  hipStream_t st = rato::as_stream(stream);
  if (!xs && p->M <= eval_tiles_max_m()) {
    const int n_tiles = (int)((p->M + RATO_WAVE - 1) / RATO_WAVE);
    const int grid = (n_tiles + EV_NW - 1) / EV_NW;
    rato_sel::StatsTail tail = {};
    int grid_launch = grid;
    size_t lds = 0;
    // The statistics behind the kernel by default: measured (profiles/r05_*_mc_step.txt) a second node costs 3.4 us of a
    // replayed graph and rs_small (1024 threads) 3.8, while the selection as ONE 256-thread workgroup of this launch (the
    // producers' block size) + the hand-off took 16.  RATO_STATS_IN_LAUNCH in stats_flags asks for the one-launch form.
    const bool in_launch = p->stats_workspace && (p->stats_flags & RATO_STATS_IN_LAUNCH) && rato_drone_eval_stats_in_launch(p->M);
    if (in_launch) {
      const int rc = stats_tail_for<RATO_BLOCK>(p->stats_workspace, p->stats_out, p->stats_alpha, p->stats_thr, p->M, grid, tail,
                                                grid_launch, lds);
      if (rc != RATO_OK) return rc;
    }
    const int rc = drone_eval_tiles_launch(p, 1, us, dW, mass, Qsym, Z, p->ld, g, tail, grid_launch, lds, n_tiles, st);
    if (rc != RATO_OK) return rc;
    if (p->stats_workspace && !in_launch)
      return rato_risk_stats(Z, p->M, p->stats_alpha, p->stats_thr, p->stats_workspace,
                             rato_risk_stats_workspace_bytes(p->M), p->stats_out, stream);
    return RATO_OK;
  }
  const int nblk = rato::nblocks_for(p->M);
  // The block queue buys the READ-bound eval kernel nothing (same box, alternating, M = 1e6 / 1e7: 0.1264-0.1281 /
  // 1.159-1.168 ms without, 0.1257-0.1276 / 1.153-1.165 ms with): the XCD asymmetry that the row kernels' queue
  // removes belongs to the store stream.  Off by default (RATO_EVAL_DYNAMIC=1 for A/B runs).
  static const int eval_dyn = [] { const char* e = getenv("RATO_EVAL_DYNAMIC"); return e ? atoi(e) : 0; }();
  unsigned* queue = nullptr;
  int grid_x = nblk;
  if (eval_dyn && nblk > 4 * 8 * device_cus()) {   // more than four rounds of 8 workgroups per CU
    queue = take_tile_queue(st);
    if (queue) grid_x = 8 * device_cus();
  }
  dim3 grid(grid_x), block(RATO_BLOCK);
  hipLaunchKernelGGL(drone_eval_kernel<false>, grid, block, 0, st, *p, us, dW, (uint64_t)0, 0.0f,
                     mass, Qsym, Z, xs, g, queue, nblk);
  RATO_LAUNCH_CHECK();
  if (p->stats_workspace)
    return rato_risk_stats(Z, p->M, p->stats_alpha, p->stats_thr, p->stats_workspace, rato_risk_stats_workspace_bytes(p->M),
                           p->stats_out, stream);
  return RATO_OK;
}

// K control sequences against ONE resident batch in one call: us [K][S][3] -> Z [K][ldz] (+ stats_out [K][RATO_N_STATS]).
extern "C" int rato_drone_eval_batch(const rato_drone_params* p, int32_t K, const float* us, const float* dW, const float* mass,
                                     const float* Qsym, float* Z, int64_t ldz, double alpha, float thr, void* workspace,
                                     size_t workspace_bytes, double* stats_out, void* stream) {
  RATO_CLEAR_ERROR();
  if (!params_ok(p) || K < 1 || K > 65535 || !us || !dW || !mass || !Qsym || !Z || ldz < p->M) return RATO_EINVAL;
  if (stats_out && (!(alpha > 0.0) || !(alpha <= 1.0) || !workspace)) return RATO_EINVAL;
  hipStream_t st = rato::as_stream(stream);
  const int n_tiles = (int)((p->M + RATO_WAVE - 1) / RATO_WAVE);
  const int grid = (n_tiles + EV_NW - 1) / EV_NW;
  rato_sel::StatsTail tail = {};
  const int rc = drone_eval_tiles_launch(p, K, us, dW, mass, Qsym, Z, ldz, nullptr, tail, grid, 0, n_tiles, st);
  if (rc != RATO_OK || !stats_out) return rc;
  return rato_risk_stats_batch(Z, p->M, ldz, K, alpha, thr, workspace, workspace_bytes, stats_out, stream);
}

extern "C" int rato_drone_eval_philox(const rato_drone_params* p, const float* us, uint64_t seed, float sampler_dt,
                                      const float* mass, const float* Qsym, float* Z, float* xs, float* g,
                                      void* stream) {
  RATO_CLEAR_ERROR();
  if (!params_ok(p) || !us || !mass || !Qsym || !(sampler_dt >= 0.0f) || p->S > 65535) return RATO_EINVAL;
  dim3 grid(rato::nblocks_for(p->M)), block(RATO_BLOCK);
  hipLaunchKernelGGL(drone_eval_kernel<true>, grid, block, 0, rato::as_stream(stream), *p, us, (const float*)nullptr,
                     seed, sqrtf(sampler_dt), mass, Qsym, Z, xs, g, (unsigned*)nullptr, 0);
  RATO_LAUNCH_CHECK();
  return RATO_OK;
}

extern "C" int rato_drone_obstacle_constraints(const rato_drone_params* p, const float* xs, const float* Qsym,
                                               float* g, void* stream) {
  RATO_CLEAR_ERROR();
  if (!params_ok(p) || !xs || !Qsym || !g) return RATO_EINVAL;
  dim3 grid(rato::nblocks_for(p->M), p->S), block(RATO_BLOCK);
  hipLaunchKernelGGL(drone_obstacle_kernel, grid, block, 0, rato::as_stream(stream), *p, xs, Qsym, g);
  RATO_LAUNCH_CHECK();
  return RATO_OK;
}

namespace {
template <int CPT, int SPL>
int launch_linearize(const rato_drone_params* p, const float* us, const float* dW, const float* mass,
                     const float* Qsym, float* G, float* g_up, float* Z, float* part, hipStream_t stream) {
  const int ngroups = (p->S + CPT - 1) / CPT;
  dim3 grid((p->M + RATO_BLOCK * SPL - 1) / (RATO_BLOCK * SPL), ngroups), block(RATO_BLOCK);
  hipLaunchKernelGGL((drone_linearize_kernel<CPT, SPL>), grid, block, 0, stream, *p, us, dW, mass, Qsym, G, g_up, Z,
                     part);
  RATO_LAUNCH_CHECK();
  return RATO_OK;
}

// Resolve cols_per_thread / samples_per_lane (0 = choose).  Tuned on MI355X
// (profiles/): wide lanes once there are enough samples to fill the chip.
size_t rows_lds_bytes(int S) { return rows_lds_floats(S) * sizeof(float); }
constexpr size_t ROWS_LDS_MAX = 160 * 1024;

bool plan(int32_t M, int32_t S, int32_t ld, int32_t* cpt, int32_t* spl) {
  if (M <= 0 || S <= 0 || ld < M) return false;
  if (*cpt == 0 && *spl == 0) {  // default: the row-parallel kernel whenever its LDS tables fit
    if (S >= 2 && rows_lds_bytes(S) <= ROWS_LDS_MAX) *cpt = -1;
  }
  if (*cpt == -1) {
    *spl = 1;
    return S >= 2 && rows_lds_bytes(S) <= ROWS_LDS_MAX;
  }
  if (*spl == 0) {
    *spl = (M >= 65536 && ld % 4 == 0) ? 4 : ((M >= 32768 && ld % 2 == 0) ? 2 : 1);
    // a given column count restricts the lane widths that exist for it
    if (*cpt == 16 || *cpt == 32) *spl = 1;
    if (*cpt == 8 && *spl == 4) *spl = (ld % 2 == 0) ? 2 : 1;
    if (*cpt == 2 && *spl != 4) *cpt = 0;  // 2 columns exist for 4-sample lanes only: let the default pick
  }
  if (*cpt == 0) *cpt = (*spl == 4) ? 4 : ((*spl == 2) ? 8 : (M >= 65536 ? 8 : 4));
  if (*spl != 1 && *spl != 2 && *spl != 4) return false;
  if (ld % *spl != 0) return false;
  const bool ok = (*spl == 1 && (*cpt == 4 || *cpt == 8 || *cpt == 16 || *cpt == 32)) ||
                  (*spl == 2 && (*cpt == 4 || *cpt == 8)) || (*spl == 4 && (*cpt == 2 || *cpt == 4));
  return ok;
}
}  // namespace

extern "C" int rato_drone_linearize_generators(const rato_drone_params* p, const float* us, const float* dW,
                                               const float* mass, const float* Qsym, float* A22, float* W,
                                               float* g_up, float* Z, float* part, void* stream) {
  RATO_CLEAR_ERROR();
  // W and g_up may BOTH be NULL: only A22 (the lane's own scratch for the backward pass), Z and the sample sums
  if (!params_ok(p) || !params64_ok(p) || !us || !dW || !mass || !Qsym || !A22 || (!W != !g_up) || !part) return RATO_EINVAL;
  dim3 grid(rato::nblocks_for(p->M)), block(RATO_BLOCK);
  const size_t lds = (size_t)(RATO_BLOCK / RATO_WAVE) * ((size_t)(p->S + 1) * 6 + (RATO_GEN_LDS_REDUCE ? 24 * 65 : 0)) * sizeof(double);
  if (lds > 160 * 1024) return RATO_EINVAL;   // S <= 570
  static rato::DynamicLdsLimit gen_lds_limit;
  {
    const hipError_t e = gen_lds_limit.ensure(lds, [](size_t bytes) {
      const void* kernels[3] = {reinterpret_cast<const void*>(drone_linearize_generators_kernel<true, true>),
                                reinterpret_cast<const void*>(drone_linearize_generators_kernel<false, true>),
                                reinterpret_cast<const void*>(drone_linearize_generators_kernel<false, false>)};
      hipError_t err = hipSuccess;
      for (int i = 0; i < 3 && err == hipSuccess; ++i)
        err = hipFuncSetAttribute(kernels[i], hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
      return err;
    });
    if (e != hipSuccess) return RATO_EHIP - (int)e;
  }
  if (W)
    hipLaunchKernelGGL((drone_linearize_generators_kernel<true, true>), grid, block, lds, rato::as_stream(stream), *p, us,
                       dW, mass, Qsym, A22, W, g_up, Z, part);
  else if (Z)
    hipLaunchKernelGGL((drone_linearize_generators_kernel<false, true>), grid, block, lds, rato::as_stream(stream), *p, us,
                       dW, mass, Qsym, A22, W, g_up, Z, part);
  else
    hipLaunchKernelGGL((drone_linearize_generators_kernel<false, false>), grid, block, lds, rato::as_stream(stream), *p, us,
                       dW, mass, Qsym, A22, W, g_up, Z, part);
  RATO_LAUNCH_CHECK();
  return RATO_OK;
}

extern "C" int rato_drone_linearize_plan(int32_t M, int32_t S, int32_t ld, int32_t* cols_per_thread,
                                         int32_t* samples_per_lane, int32_t* tile) {
  if (!cols_per_thread || !samples_per_lane || !tile) return RATO_EINVAL;
  if (!plan(M, S, ld, cols_per_thread, samples_per_lane)) return RATO_EINVAL;
  *tile = (*cols_per_thread == -1) ? ROWS_SAMPLES : RATO_TILE;
  if (*cols_per_thread == -1) return (M + ROWS_SAMPLES - 1) / ROWS_SAMPLES;
  return (M + RATO_BLOCK * *samples_per_lane - 1) / (RATO_BLOCK * *samples_per_lane);
}

namespace {
// dW == NULL: the noise is regenerated from (seed, noise_scale) -- the row-parallel kernel only
int drone_linearize_impl(const rato_drone_params* p, const float* us, const float* dW, uint64_t seed, float noise_scale,
                         const float* mass, const float* Qsym, float* G, float* W, float* A22, float* g_up, float* Z,
                         float* part, int32_t cols_per_thread, int32_t samples_per_lane, void* stream,
                         int noise_tiled = 0) {
  RATO_CLEAR_ERROR();
  if (!params_ok(p) || !us || !mass || !Qsym || !G || !g_up || !part) return RATO_EINVAL;
  int32_t cpt = cols_per_thread, spl = samples_per_lane;
  if (!plan(p->M, p->S, p->ld, &cpt, &spl)) return RATO_EINVAL;
  if (!dW && cpt != -1) return RATO_EINVAL;
  if (noise_tiled && (cpt != -1 || !dW)) return RATO_EINVAL;   // the tiled noise layout: row-parallel kernel only
  if (W && cpt != -1) return RATO_EINVAL;  // the factored output exists for the row-parallel kernel only
  if (p->stats_workspace && (cpt != -1 || !Z || !p->stats_out || !(p->stats_alpha > 0.0) || !(p->stats_alpha <= 1.0)))
    return RATO_EINVAL;   // statistics in the same launch: row-parallel kernel, Z requested
  if (A22 && !W) return RATO_EINVAL;       // the step-Jacobian table goes with the factored output
  hipStream_t st = rato::as_stream(stream);
  if (cpt == -1) {
    const size_t lds = rows_lds_bytes(p->S);
    // Device properties are cached so that a launch inside a hipGraph capture makes no non-stream
    // runtime call (the first, uncaptured call sets them).
    static rato::DynamicLdsLimit lds_limit;   // per device
    {
      const hipError_t e = lds_limit.ensure(lds, [](size_t bytes) {
        hipError_t err = hipSuccess;
        const void* kernels[4] = {reinterpret_cast<const void*>(drone_linearize_rows_kernel<true, false>),
                                  reinterpret_cast<const void*>(drone_linearize_rows_kernel<false, false>),
                                  reinterpret_cast<const void*>(drone_linearize_rows_kernel<true, true>),
                                  reinterpret_cast<const void*>(drone_linearize_rows_kernel<false, true>)};
        for (int i = 0; i < 4 && err == hipSuccess; ++i)
          err = hipFuncSetAttribute(kernels[i], hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        return err;
      });
      if (e != hipSuccess) return RATO_EHIP - (int)e;
    }
    // Small batches (fewer tiles than resident workgroup slots) deal each tile's row tasks out to
    // row_split workgroups so that the chip is filled (M = 1e4, S = 50: 81 -> 72 us).  Splitting only
    // the tiles of an incomplete last round of a large batch was measured and does not pay
    // (M = 1e5: 0.627 -> 0.648 ms), so large batches use one workgroup per tile.
    const int cus = device_cus();
    int per_cu = (int)(ROWS_LDS_MAX / lds);
    if (per_cu > 32 / ROWS_NW) per_cu = 32 / ROWS_NW;
    if (per_cu < 1) per_cu = 1;
    // RATO_ROWS_SLOTS_PER_CU: A/B knob -- fewer queue workgroups than the LDS allows (one per CU = 256 store streams)
    static const int slots_env = [] { const char* e = getenv("RATO_ROWS_SLOTS_PER_CU"); return e ? atoi(e) : 0; }();
    if (slots_env >= 1 && slots_env < per_cu) per_cu = slots_env;
    const int slots = cus * per_cu;
    const int n_tiles = (p->M + ROWS_SAMPLES - 1) / ROWS_SAMPLES;
    const int max_split = (p->S + 3) / 4 < 1 ? 1 : (p->S + 3) / 4;   // keep >= 4 row tasks per workgroup
    int split = 1, n_whole = n_tiles;
    if (n_tiles < slots) {   // small batch: every tile split so that the chip is filled (M = 1e4, S = 50: 81 -> 72 us)
      static const int small_split_env = [] { const char* e = getenv("RATO_SMALL_SPLIT"); return e ? atoi(e) : 0; }();   // A/B
      // Re-measured with the tiles on 2 MiB boundaries (RATO_SMALL_SPLIT sweep, kernel ms, products / factored):
      // M = 2000 (32 tiles): split 1 0.0439 / 0.0387, 2 0.0339 / 0.0336, 4 0.0342 / 0.0340, slots / n_tiles 0.0378 / 0.0376;
      // M = 5000: 1 0.0456 / 0.0397, 2 0.0389 / 0.0349, 4 0.0453 / 0.0387; M = 1e4 (C2): 1 0.0609, 2 0.0645, 3 0.0653;
      // M = 2e4: 1 0.1208 / 0.0632, 2 0.1213 / 0.0731 -> two parts while that still leaves one workgroup per CU, else none.
      split = small_split_env > 0 ? small_split_env : (2 * n_tiles <= cus ? 2 : 1);
      if (split > max_split) split = max_split;
      if (split < 1) split = 1;
      n_whole = split > 1 ? 0 : n_tiles;
    }
    // (Splitting the tiles of the last round of a STATIC grid was measured twice and rejected: every extra work unit
    // spends ~15-25 us staging and rolling out in one of only 512 LDS-limited slots -- profiles/README.md.)
    int grid = n_whole + (n_tiles - n_whole) * split, stride = 0;
    unsigned* queue = nullptr;
    // Large batches: a grid that fills every slot once + a global tile counter (see the kernel).  Why: with one tile
    // per workgroup the timeline (tools/timeline.py, -DRATO_DIAG=4, M = 1e5) shows the workgroups with an even block
    // index -- every other XCD -- running their tiles in 145-151 us and the odd ones in 174-176 us, the hardware
    // having dealt the grid out to the XCDs in advance: the fast half of the chip is done at 500-518 us and idles
    // until the slow half finishes at 585-590 us.  RATO_ROWS_DYNAMIC=0 switches it off (the bit-identity test's base).
    static const int dynamic_env = [] { const char* e = getenv("RATO_ROWS_DYNAMIC"); return e ? atoi(e) : 1; }();
    if (split == 1 && n_tiles > slots) {
      if (dynamic_env >= 1) {
        // Measured, same box, alternating (profiles/r02_ab_rows.txt): factored output -7 % at M = 1e5, -9 % at M = 1e6;
        // products output -1.5 % at M = 1e5, and -- since its tiles start on 2 MiB boundaries (rato_packed_tile_stride)
        // -- also at large batches: M = 4e5 2.148 / 2.147 ms against 2.176 / 2.213 static, M = 1e6 5.298 / 5.369 against
        // 5.344 / 5.477 (tools/ab_big_products.sh; with the tiles back to back the queue had cost +1.3 % / +5 % there).
        // 64 two-word queues in device memory, handed out round robin: launches that overlap on different streams get
        // different queues; each launch leaves its queue zeroed.  (Address looked up once, outside any capture.)
        queue = take_tile_queue(st);
        if (queue) {
          // Products output, four or more rounds of tiles: ONE queue workgroup per CU (256 store streams instead of
          // 512) is as fast or faster than the two the LDS allows -- same box, alternating (tools/ab_slots.sh,
          // ab_slots2.sh), 2 -> 1 per CU: M = 1e5 0.5600 -> 0.5605 ms (noise read) / 0.5355 -> 0.5256 (regenerated),
          // 2e5 1.110 -> 1.102 / 1.050 -> 1.025, 1e6 5.425 -> 5.311 / 5.097 -> 5.025; at 5e4 +1.2 % / -1.2 %.
          // The factored output needs the second workgroup (its tiles are a third as long: 0.2455 -> 0.2648 ms).
          // RATO_ROWS_SLOTS_PER_CU overrides.
          static const int qslots_env = [] { const char* e = getenv("RATO_ROWS_QSLOTS"); return e ? atoi(e) : 0; }();   // A/B: absolute
          const int qslots = qslots_env > 0 ? qslots_env : ((!W && slots_env < 1 && n_tiles >= 1024) ? cus : slots);
          grid = qslots;
          // Products output: the LAST tiles are handed out in row-interleaved parts (round 3: quarters of the last slots / 2
          // tiles; round 6: halves of the last `slots` tiles, below).
          // The drain at the end of the launch is bounded per workgroup (~19 GB/s each, whatever the residency), so
          // shorter last units shorten it; the re-staging they cost is paid while the chip is still full.  Same box,
          // alternating, 100 steps (tools/dyn_tail_sweep.sh): 0.5543-0.5576 -> 0.5415-0.5440 ms (-2.4 %, 0.704-0.708
          // of 8 TB/s); halves over the last 1024 tiles -1 %; thirds / sixths / eighths no better.  The factored
          // output loses with any split (its tiles are short already) and keeps whole tiles.
          // RATO_DYN_TAIL_SPLIT x RATO_DYN_TAIL_TILES override (split 1 = whole tiles only).
          static const int dts = [] { const char* e = getenv("RATO_DYN_TAIL_SPLIT"); return e ? atoi(e) : 0; }();
          static const int dtt = [] { const char* e = getenv("RATO_DYN_TAIL_TILES"); return e ? atoi(e) : 0; }();
          // Round 6, re-measured on three boards (same board, alternating, kernel ms by events; tools/ab.sh): quarters over
          // the last 128 tiles (the round-3 choice) 0.5139 / 0.5076 / 0.5135, whole tiles 0.5117 / 0.5036 / 0.5117, HALVES
          // over the last 128 / 256 / 384 tiles 0.5059 / 0.4980 / 0.5004, 0.5067 / 0.5055 / 0.4986, 0.4984 / 0.5046 / 0.4969;
          // eighths 0.523-0.538.  Since the streaming stores (round 4) a re-staged unit costs more than it did (its noise is
          // no longer re-read from HBM by anyone else in between): halves over the last round of tiles are the default.
          int want_split = dts > 0 ? dts : (W ? 1 : 2);
          int want_tiles = dtt > 0 ? dtt : qslots;
          if (want_split > max_split) want_split = max_split;
          if (want_split > 1 && want_tiles > 0) {
            split = want_split;
            n_whole = n_tiles - (want_tiles < n_tiles - qslots ? want_tiles : n_tiles - qslots);
          }
        }
      }
    }
    // statistics of Z in extra workgroups of this launch (params.stats_*)
    rato_sel::StatsTail tail = {};
    size_t lds_launch = lds;
    int grid_launch = grid;
    // (in the launch only while the whole grid is resident at once -- no queue; otherwise behind it, below)
    const bool stats_behind = p->stats_workspace && queue;
    if (p->stats_workspace && !stats_behind) {
      int Gs = 0;
      const int extra = rato_sel::stats_tail_workgroups<ROWS_NW * RATO_WAVE>(p->M, Gs);
      if (extra < 0) return RATO_EINVAL;   // beyond the one-launch forms of the selection: use rato_risk_stats
      tail.ws = static_cast<rato_sel::Workspace*>(p->stats_workspace);
      tail.out = p->stats_out;
      tail.alpha = p->stats_alpha;
      tail.thr = p->stats_thr;
      tail.G = Gs;
      tail.n_prod = grid;
      rato_sel::stats_rank(p->M, p->stats_alpha, tail.k, tail.var_is_max);
      grid_launch = grid + extra;
      if (lds_launch < rato_sel::rs_body_lds_bytes<ROWS_NW * RATO_WAVE>()) lds_launch = rato_sel::rs_body_lds_bytes<ROWS_NW * RATO_WAVE>();
    }
    // streaming stores for the Jacobian: when the output cannot stay in the 256 MB memory-side cache anyway and the batch's
    // inputs (the noise) can -- RATO_NT_STORES=0 never / 2 always (A/B).  M = 1e5, S = 50: 3 GB out, 60 MB in: yes.
    const bool nt_stores = rato_drone_rows_streaming_stores(p->M, p->S, W ? 1 : 0) != 0;
#define RATO_ROWS_LAUNCH(F, PH)                                                                                     \
  hipLaunchKernelGGL((drone_linearize_rows_kernel<F, PH>), dim3(grid_launch), dim3(ROWS_NW * RATO_WAVE), lds_launch, st, \
                     *p, n_whole, split, stride, n_tiles, queue, seed, noise_scale, us, dW, mass, Qsym, G, W, A22, g_up, \
                     Z, part, tail, (noise_tiled ? 1 : 0) | (nt_stores ? 2 : 0) | ((RATO_DIAG == 4 && getenv("RATO_DIAG_WAIT")) ? 4 : 0))
    if (W) {
      if (dW) RATO_ROWS_LAUNCH(true, false); else RATO_ROWS_LAUNCH(true, true);
    } else {
      if (dW) RATO_ROWS_LAUNCH(false, false); else RATO_ROWS_LAUNCH(false, true);
    }
#undef RATO_ROWS_LAUNCH
    RATO_LAUNCH_CHECK();
    if (stats_behind)
      return rato_risk_stats(Z, p->M, p->stats_alpha, p->stats_thr, p->stats_workspace,
                             rato_risk_stats_workspace_bytes(p->M), p->stats_out, stream);
    return RATO_OK;
  }
#define RATO_CASE(C, L) \
  if (cpt == C && spl == L) return launch_linearize<C, L>(p, us, dW, mass, Qsym, G, g_up, Z, part, st)
  RATO_CASE(4, 1);
  RATO_CASE(8, 1);
  RATO_CASE(16, 1);
  RATO_CASE(32, 1);
  RATO_CASE(4, 2);
  RATO_CASE(8, 2);
  RATO_CASE(2, 4);
  RATO_CASE(4, 4);
#undef RATO_CASE
  return RATO_EINVAL;
}
}  // namespace

// The launcher's store policy for the row-parallel kernel (1: the Jacobian goes out as streaming stores): when the output
// cannot stay in the 256 MB memory-side cache anyway (>= 256 MB) and the batch's noise can (<= 128 MB).
extern "C" int rato_drone_rows_streaming_stores(int64_t M, int32_t S, int32_t factored) {
  static const int nt_env = [] { const char* e = getenv("RATO_NT_STORES"); return e ? atoi(e) : 1; }();
  const double out_bytes = (double)M * (double)rato::pair_row_offset(S) * (factored ? 2.0 : 6.0) * 4.0;
  const double in_bytes = (double)M * S * 3.0 * 4.0;
  return (nt_env == 2 || (nt_env == 1 && out_bytes >= 256e6 && in_bytes <= 128e6)) ? 1 : 0;
}

extern "C" int rato_drone_linearize(const rato_drone_params* p, const float* us, const float* dW,
                                    const float* mass, const float* Qsym, float* G, float* W, float* A22,
                                    float* g_up, float* Z, float* part, int32_t cols_per_thread,
                                    int32_t samples_per_lane, void* stream) {
  if (!dW) return RATO_EINVAL;
  return drone_linearize_impl(p, us, dW, 0, 0.0f, mass, Qsym, G, W, A22, g_up, Z, part, cols_per_thread,
                              samples_per_lane, stream);
}

extern "C" int rato_drone_linearize_philox(const rato_drone_params* p, const float* us, uint64_t seed,
                                           float sampler_dt, const float* mass, const float* Qsym, float* G, float* W,
                                           float* A22, float* g_up, float* Z, float* part, void* stream) {
  if (!(sampler_dt > 0.0f)) return RATO_EINVAL;
  return drone_linearize_impl(p, us, nullptr, seed, sqrtf(sampler_dt), mass, Qsym, G, W, A22, g_up, Z, part, -1, 0,
                              stream);
}

namespace {
__global__ __launch_bounds__(RATO_BLOCK) void drone_tile_noise_kernel(const float* __restrict__ dW, long M, long ld, int nrows,
                                                                     float* __restrict__ out) {
  const long i = (long)blockIdx.x * RATO_BLOCK + threadIdx.x;        // over [tile][row][lane]
  const long n_tiles = (M + ROWS_SAMPLES - 1) / ROWS_SAMPLES;
  if (i >= n_tiles * nrows * ROWS_SAMPLES) return;
  // a tile's block is the image the row kernel keeps in LDS: [S][64] (xi_x, xi_y) pairs, then [S][64] xi_z
  const int S = nrows / 3;
  const long tile = i / ((long)nrows * ROWS_SAMPLES);
  const int o = (int)(i - tile * (long)nrows * ROWS_SAMPLES);
  int t, lane, a;
  if (o < S * 2 * ROWS_SAMPLES) {
    t = o / (2 * ROWS_SAMPLES);
    lane = (o % (2 * ROWS_SAMPLES)) / 2;
    a = o & 1;
  } else {
    const int o2 = o - S * 2 * ROWS_SAMPLES;
    t = o2 / ROWS_SAMPLES;
    lane = o2 % ROWS_SAMPLES;
    a = 2;
  }
  const long m = tile * ROWS_SAMPLES + lane;
  out[i] = (m < M) ? dW[(size_t)(t * 3 + a) * ld + m] : 0.0f;
}
}  // namespace

extern "C" size_t rato_drone_tiled_noise_floats(int64_t M, int32_t S) {
  return (M > 0 && S > 0) ? (size_t)((M + ROWS_SAMPLES - 1) / ROWS_SAMPLES) * (size_t)(3 * S) * ROWS_SAMPLES : 0;
}

extern "C" int rato_drone_tile_noise(const float* dW, int64_t M, int64_t ld, int32_t S, float* dW_tiled, void* stream) {
  RATO_CLEAR_ERROR();
  if (!dW || !dW_tiled || M <= 0 || ld < M || S <= 0) return RATO_EINVAL;
  const size_t n = rato_drone_tiled_noise_floats(M, S);
  hipLaunchKernelGGL(drone_tile_noise_kernel, dim3((unsigned)((n + RATO_BLOCK - 1) / RATO_BLOCK)), dim3(RATO_BLOCK), 0,
                     rato::as_stream(stream), dW, (long)M, (long)ld, 3 * S, dW_tiled);
  RATO_LAUNCH_CHECK();
  return RATO_OK;
}

extern "C" int rato_drone_linearize_tiled(const rato_drone_params* p, const float* us, const float* dW_tiled,
                                          const float* mass, const float* Qsym, float* G, float* W, float* A22,
                                          float* g_up, float* Z, float* part, void* stream) {
  if (!dW_tiled) return RATO_EINVAL;
  return drone_linearize_impl(p, us, dW_tiled, 0, 0.0f, mass, Qsym, G, W, A22, g_up, Z, part, -1, 0, stream, 1);
}

// Would rato_drone_linearize (row-parallel kernel) with params.stats_* compute the statistics IN its launch?  (Small
// batches: fewer tiles than resident workgroup slots, i.e. no tile queue; the facades then ask for them there and fold the
// statistics of larger batches into their partial-sum launch instead.)  1 yes, 0 no.
extern "C" int rato_drone_stats_in_launch(int32_t M, int32_t S) {
  if (M <= 0 || S < 2 || rows_lds_bytes(S) > ROWS_LDS_MAX) return 0;
  int per_cu = (int)(ROWS_LDS_MAX / rows_lds_bytes(S));
  if (per_cu > 32 / ROWS_NW) per_cu = 32 / ROWS_NW;
  if (per_cu < 1) per_cu = 1;
  const int n_tiles = (M + ROWS_SAMPLES - 1) / ROWS_SAMPLES;
  int G = 0;
  return n_tiles <= device_cus() * per_cu && rato_sel::stats_tail_workgroups<ROWS_NW * RATO_WAVE>(M, G) > 0;
}
