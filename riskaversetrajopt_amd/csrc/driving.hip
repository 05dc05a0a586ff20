// Driving (ego car + pedestrian) SAA kernels (gfx950).  Replaces
// driving.py:145-236 (social-force dynamics, rollout, separation distance),
// :260-313 (control-Jacobian + mean) and :630-638 (Monte-Carlo closure).
//
// Structure exploited (verified in tests/test_oracle_driving.py): the ego
// sub-state x[0:4] and its control sensitivity E_t = d x_ego,t / d u are bit-
// identical across samples, so one small prologue kernel computes them once per
// call into `ego_scratch`; the sample kernels read them through wave-uniform
// (scalar) loads.  Only the pedestrian's 4-state sensitivity is per sample:
//   dpp+ = dpp + dt dvp
//   dvp+ = dvp + dt [ w_r H (dpp - dE_pos) - w_s dvy (1,1)^T ],  H = (I - n n^T)/r
//   d g_t / d u_c = -n_{t+1}^T (E_{t+1}[pos, c] - dpp_{t+1})
// Lane = sample; grid.y = groups of control steps s (both controls of a step
// share one slot because they share the activity window t > s).
#include <stdlib.h>

#include <atomic>

#include "philox.h"
#include "rato_common.h"
#include "rato_select.h"

namespace {

// ego_scratch layout (floats):
//   ego   [(S+1)][4]          ego trajectory
//   Eu    [(S+1)][2]          E_t[pos,:] . u   (tangent of the ego position along u)
//   Epos  [(S+1)][2][2S]      E_t[pos, c],  c = s*2 + i
__host__ __device__ inline size_t ego_off_Eu(int S) { return (size_t)(S + 1) * 4; }
__host__ __device__ inline size_t ego_off_Epos(int S) { return ego_off_Eu(S) + (size_t)(S + 1) * 2; }
__host__ __device__ inline size_t ego_total(int S) { return ego_off_Epos(S) + (size_t)(S + 1) * 2 * 2 * S; }
inline size_t ego_lds_bytes(int S) {   // car_ego_kernel's tables: 6S+4 floats, then 3(S+1) doubles
  return (size_t)((6 * S + 4 + 1) & ~1) * sizeof(float) + (size_t)(S + 1) * 3 * sizeof(double);
}

// One block, the sample-independent prologue of every driving call (driving.py:166-173 and its control
// sensitivity).  The first version rolled the ego out on ONE thread, re-evaluated sincosf in every (column, step) and
// went through global memory between its stages: 30 us, 18 % of a linearize call at M = 1e5.  Here
//   A  thread t folds the trajectory up to step t in fp64 (rounded once) and evaluates sin / cos(phi_t),
//   D  one thread per control column propagates E (4 rows) through A^e_t = d ego_{t+1} / d ego_t from the tables,
// with the ego table in LDS; the per-step tangents Epos / Eu are written only when the caller needs them
// (want_E: the forward/column linearize kernel; not the row-parallel kernel, not eval).
__global__ __launch_bounds__(RATO_BLOCK) void car_ego_kernel(rato_car_params P, const float* __restrict__ us,
                                                            float* __restrict__ scratch,
                                                            float* __restrict__ final_du,
                                                            float* __restrict__ final_rhs, int want_E) {
  extern __shared__ __attribute__((aligned(8))) float ego_lds[];   // v | ph | x | y [S+1] each | cs | sn [S] each | fp64 v, cos, sin [S+1] each
  const int S = P.S, NC = 2 * S;
  float* sv = ego_lds;
  float* sph = sv + (S + 1);
  float* sx = sph + (S + 1);
  float* sy = sx + (S + 1);
  float* scs = sy + (S + 1);
  float* ssn = scs + S;
  double* dv = reinterpret_cast<double*>(ego_lds + ((6 * S + 4 + 1) & ~1));
  double* dcs = dv + (S + 1);
  double* dsn = dcs + (S + 1);
  float* ego = scratch;
  float* Eu = scratch + ego_off_Eu(S);
  float* Epos = scratch + ego_off_Epos(S);
  // one thread per step folds the trajectory in fp64 and rounds once (identical to the row-parallel kernel's
  // in-kernel tables, so eval and linearize see the same ego)
  const double dt = P.dt;
  for (int t = threadIdx.x; t <= S; t += RATO_BLOCK) {
    double v = P.ego_init[2], ph = P.ego_init[3];
    for (int k = 0; k < t; ++k) {
      v += dt * (double)us[k * 2 + 0];
      ph += dt * (double)us[k * 2 + 1];
    }
    double sn, cs;
    sincos(ph, &sn, &cs);
    dv[t] = v;
    dcs[t] = cs;
    dsn[t] = sn;
    sv[t] = (float)v;
    sph[t] = (float)ph;
    if (t < S) {
      scs[t] = (float)cs;
      ssn[t] = (float)sn;
    }
  }
  __syncthreads();
  for (int t = threadIdx.x; t <= S; t += RATO_BLOCK) {
    double x = P.ego_init[0], y = P.ego_init[1];
    for (int k = 0; k < t; ++k) {
      x += dt * dv[k] * dcs[k];
      y += dt * dv[k] * dsn[k];
    }
    sx[t] = (float)x;
    sy[t] = (float)y;
    ego[t * 4 + 0] = sx[t]; ego[t * 4 + 1] = sy[t]; ego[t * 4 + 2] = sv[t]; ego[t * 4 + 3] = sph[t];
  }
  __syncthreads();
  if (!final_du && !final_rhs && !want_E) return;   // eval: the trajectory is all that is needed
  // D: control columns
  __shared__ float red[RATO_BLOCK / RATO_WAVE][4];
  float rhs_acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int c = threadIdx.x; c < NC; c += RATO_BLOCK) {
    const int s = c >> 1, i = c & 1;
    const float uc = us[c];
    float e0 = 0.f, e1 = 0.f, e2 = 0.f, e3 = 0.f;
    if (want_E) {
      Epos[(size_t)(0 * 2 + 0) * NC + c] = 0.f;
      Epos[(size_t)(0 * 2 + 1) * NC + c] = 0.f;
    }
    for (int t = 0; t < S; ++t) {
      const float v = sv[t], sn = ssn[t], cs = scs[t];
      const float n0 = e0 + P.dt * cs * e2 - P.dt * v * sn * e3;
      const float n1 = e1 + P.dt * sn * e2 + P.dt * v * cs * e3;
      float n2 = e2, n3 = e3;
      if (t == s) {
        if (i == 0) n2 += P.dt; else n3 += P.dt;
      }
      e0 = n0; e1 = n1; e2 = n2; e3 = n3;
      if (want_E) {
        Epos[(size_t)((t + 1) * 2 + 0) * NC + c] = e0;
        Epos[(size_t)((t + 1) * 2 + 1) * NC + c] = e1;
      }
    }
    if (final_du) {
      final_du[0 * NC + c] = e0; final_du[1 * NC + c] = e1; final_du[2 * NC + c] = e2; final_du[3 * NC + c] = e3;
    }
    rhs_acc[0] += e0 * uc; rhs_acc[1] += e1 * uc; rhs_acc[2] += e2 * uc; rhs_acc[3] += e3 * uc;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float sres = rato::wave_sum(rhs_acc[r]);
    if (lane == 0) red[wave][r] = sres;
  }
  __syncthreads();
  if (final_rhs && threadIdx.x < 4) {
    float acc = 0.f;
#pragma unroll
    for (int w = 0; w < RATO_BLOCK / RATO_WAVE; ++w) acc += red[w][threadIdx.x];
    // val_final = -(x_S[:4] - goal) + v_final_du . u   (driving.py:288)
    const float xS = (threadIdx.x == 0) ? sx[S] : ((threadIdx.x == 1) ? sy[S] : ((threadIdx.x == 2) ? sv[S] : sph[S]));
    final_rhs[threadIdx.x] = -(xS - P.ego_goal[threadIdx.x]) + acc;
  }
  if (!want_E) return;
  __threadfence_block();
  __syncthreads();
  // Eu[t] = Epos[t] . u  (one thread per (t, axis); fixed summation order)
  for (int idx = threadIdx.x; idx < (S + 1) * 2; idx += RATO_BLOCK) {
    float acc = 0.f;
    for (int c = 0; c < NC; ++c) acc += Epos[(size_t)idx * NC + c] * us[c];
    Eu[idx] = acc;
  }
}

struct PedConsts {
  float w_s, w_r, cn;
};

// Pedestrian step (driving.py:145-158,160-178,196-203) given the ego position at t.
// Also returns the unit normal / inverse distance at time t for the Jacobian.
__device__ __forceinline__ void ped_step(const rato_car_params& P, const PedConsts& c, float ex, float ey,
                                         float xi0, float xi1, float& px, float& py, float& vx, float& vy,
                                         float& n0, float& n1, float& rinv) {
  const float dx = ex - px, dy = ey - py;
  const float r2 = dx * dx + dy * dy;
  rinv = __builtin_amdgcn_rsqf(r2);  // v_rsq_f32 (1 ulp); r2 = squared ego-pedestrian distance, O(1..100) m^2
  n0 = dx * rinv;
  n1 = dy * rinv;
  const float common = c.w_s * (P.speed_ped_des - vy);  // added to BOTH components (:156-157)
  const float F0 = -c.w_r * n0 + common, F1 = -c.w_r * n1 + common;
  const float pxn = px + P.dt * vx, pyn = py + P.dt * vy;
  const float vxn = vx + P.dt * F0 + c.cn * xi0, vyn = vy + P.dt * F1 + c.cn * xi1;
  px = pxn; py = pyn; vx = vxn; vy = vyn;
}

// The pedestrian step and the distance row with every rounding spelled out (see step_axis_exact in drone.hip): what
// car_eval_kernel AND car_eval_tiles_kernel call, so that their Z / g / trajectories agree to the bit by construction.
// (-ffp-contract=fast lets the BACK END fuse any multiply into a following add whatever the pragma says: a product that
//  must stay a product goes through an empty asm statement, which hides where it came from)
__device__ __forceinline__ float unfused(float x) {
  asm volatile("" : "+v"(x));
  return x;
}
__device__ __forceinline__ void ped_step_exact(const rato_car_params& P, const PedConsts& c, float ex, float ey, float xi0,
                                               float xi1, float& px, float& py, float& vx, float& vy) {
#pragma clang fp contract(off)
  const float dx = ex - px, dy = ey - py;
  const float r2 = unfused(dx * dx) + unfused(dy * dy);
  const float rinv = __builtin_amdgcn_rsqf(r2);
  const float common = c.w_s * (P.speed_ped_des - vy);
  const float F0 = __builtin_fmaf(-c.w_r, dx * rinv, common), F1 = __builtin_fmaf(-c.w_r, dy * rinv, common);
  const float pxn = __builtin_fmaf(P.dt, vx, px), pyn = __builtin_fmaf(P.dt, vy, py);
  vx = __builtin_fmaf(c.cn, xi0, __builtin_fmaf(P.dt, F0, vx));
  vy = __builtin_fmaf(c.cn, xi1, __builtin_fmaf(P.dt, F1, vy));
  px = pxn;
  py = pyn;
}
__device__ __forceinline__ float separation_row_exact(const rato_car_params& P, float ex, float ey, float px, float py) {
#pragma clang fp contract(off)
  const float dx = ex - px, dy = ey - py;
  const float d2 = unfused(dx * dx) + unfused(dy * dy);
  return -__builtin_fmaf(d2, __builtin_amdgcn_rsqf(d2), -P.d_min);
}

// PHILOX: regenerate the two pedestrian noise components of step t in the kernel (philox.h; bit-identical to
// rato_car_sample's dW) instead of reading 8 B per sample-step.
template <bool PHILOX>
__global__ __launch_bounds__(RATO_BLOCK) void car_eval_kernel(
    rato_car_params P, const float* __restrict__ dW, uint64_t seed, float noise_scale,
    const float* __restrict__ x0_ped, const float* __restrict__ w_speed, const float* __restrict__ w_rep,
    const float* __restrict__ scratch, float* __restrict__ Z, float* __restrict__ xs, float* __restrict__ g) {
  const size_t M = (size_t)P.M;
  const size_t m = (size_t)blockIdx.x * RATO_BLOCK + threadIdx.x;
  if (m >= M) return;
  const int S = P.S;
  const float* __restrict__ ego = scratch;
  PedConsts c;
  c.w_s = w_speed[m];
  c.w_r = w_rep[m];
  c.cn = sqrtf(P.dt) * P.beta;  // sqrt(dt) * beta: driving.py:183,200
  float px = x0_ped[0 * M + m], py = x0_ped[1 * M + m], vx = x0_ped[2 * M + m], vy = x0_ped[3 * M + m];
  if (xs) {
#pragma unroll
    for (int k = 0; k < 4; ++k) xs[(size_t)k * M + m] = ego[k];
    xs[(size_t)4 * M + m] = px; xs[(size_t)5 * M + m] = py; xs[(size_t)6 * M + m] = vx; xs[(size_t)7 * M + m] = vy;
  }
  float zmax = -INFINITY;
  float xi0 = 0.0f, xi1 = 0.0f;
  if (!PHILOX) {
    xi0 = dW[m];
    xi1 = dW[M + m];
  }
  for (int t = 0; t < S; ++t) {
    float nx0 = 0.0f, nx1 = 0.0f;
    if (PHILOX) {
      const rato::u32x4 r = rato::philox_at(seed, rato::PHILOX_STREAM_DW, (uint32_t)t, (uint64_t)m);
      rato::box_muller(r.x, r.y, xi0, xi1);
      xi0 *= noise_scale;
      xi1 *= noise_scale;
    } else {
      const int tn = (t + 1 < S) ? t + 1 : t;
      nx0 = dW[(size_t)(tn * 2 + 0) * M + m];
      nx1 = dW[(size_t)(tn * 2 + 1) * M + m];
    }
    ped_step_exact(P, c, ego[t * 4 + 0], ego[t * 4 + 1], xi0, xi1, px, py, vx, vy);
    const float gt = separation_row_exact(P, ego[(t + 1) * 4 + 0], ego[(t + 1) * 4 + 1], px, py);   // -(||p_e - p_p|| - d_min): driving.py:223-230,269
    zmax = fmaxf(zmax, gt);
    if (g) g[(size_t)t * M + m] = gt;
    if (xs) {
#pragma unroll
      for (int k = 0; k < 4; ++k) xs[((size_t)(t + 1) * 8 + k) * M + m] = ego[(t + 1) * 4 + k];
      xs[((size_t)(t + 1) * 8 + 4) * M + m] = px; xs[((size_t)(t + 1) * 8 + 5) * M + m] = py;
      xs[((size_t)(t + 1) * 8 + 6) * M + m] = vx; xs[((size_t)(t + 1) * 8 + 7) * M + m] = vy;
    }
    if (!PHILOX) {
      xi0 = nx0;
      xi1 = nx1;
    }
  }
  if (Z) Z[m] = zmax - P.tol;
}

// The Monte-Carlo form for small batches (driving.py:618-740: M = 1e4), for calls that want Z (and g) but no
// trajectories -- see drone_eval_tiles_kernel (drone.hip) for the structure: one wave per tile of 64 samples, the noise of
// 2 x CEV_TB steps in flight before the first step, the statistics of Z in the same launch.  The ego trajectory
// (sample-independent) is folded by every workgroup into its own LDS exactly as car_ego_kernel does (fp64, rounded
// once): ONE launch instead of the ego kernel + the rollout + the statistics, and the same Z / g to the bit.
constexpr int CEV_TB = 16;
constexpr int CEV_NW = RATO_BLOCK / RATO_WAVE;
__host__ __device__ inline size_t car_eval_tiles_lds_bytes(int S) {   // ego xy (float2) [S+1] | us [2S] | fp64 v, cos, sin [S+1]
  return (((size_t)(S + 1) * 2 + (size_t)S * 2 + 1) & ~size_t(1)) * sizeof(float) + (size_t)(S + 1) * 3 * sizeof(double);
}

// blockIdx.y = control sequence k of a batch of K (rato_car_eval_batch: us [K][S][2], Z [K][ldz]); K = 1: rato_car_eval.
template <bool WANT_G>
__global__ __launch_bounds__(RATO_BLOCK) void car_eval_tiles_kernel(
    rato_car_params P, const float* __restrict__ us_base, const float* __restrict__ dW, const float* __restrict__ x0_ped,
    const float* __restrict__ w_speed, const float* __restrict__ w_rep, float* __restrict__ Z_base, long ldz,
    float* __restrict__ g, int n_tiles, const rato_sel::StatsTail tail) {
  extern __shared__ __attribute__((aligned(16))) unsigned char cev_lds[];
  if (tail.is_stats((int)blockIdx.x)) {
    rato_sel::stats_tail_run<RATO_BLOCK>(tail, Z_base, (long)P.M, cev_lds);
    return;
  }
  const int S = P.S;
  const float* __restrict__ us = us_base + (size_t)blockIdx.y * S * 2;
  float* __restrict__ Z = Z_base ? Z_base + (size_t)blockIdx.y * ldz : nullptr;
  float* EGO = reinterpret_cast<float*>(cev_lds);                 // [S+1][2]
  float* US = EGO + (size_t)(S + 1) * 2;                          // [S][2]
  double* DV = reinterpret_cast<double*>(cev_lds + ((((size_t)(S + 1) * 2 + (size_t)S * 2 + 1) & ~size_t(1)) * sizeof(float)));
  double* DCS = DV + (S + 1);
  double* DSN = DCS + (S + 1);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int tile = (int)blockIdx.x * CEV_NW + wave;
  const bool has_tile = tile < n_tiles;            // (wave-uniform; every wave helps with the ego tables)
  const size_t M = (size_t)P.M;
  const size_t m_raw = (size_t)(has_tile ? tile : 0) * RATO_WAVE + lane;
  const bool valid = has_tile && m_raw < M;
  const size_t m = m_raw < M ? m_raw : M - 1;
  float xa[CEV_TB][2], xb[CEV_TB][2];
  auto load = [&](float (&xi)[CEV_TB][2], int t0) {
#pragma unroll
    for (int i = 0; i < CEV_TB; ++i) {
      const int t = (t0 + i < S) ? t0 + i : S - 1;
      xi[i][0] = dW[(size_t)(t * 2 + 0) * M + m];
      xi[i][1] = dW[(size_t)(t * 2 + 1) * M + m];
    }
  };
  load(xa, 0);                                     // in flight while the ego tables are folded
  load(xb, CEV_TB);
  PedConsts c;
  c.w_s = w_speed[m];
  c.w_r = w_rep[m];
  c.cn = sqrtf(P.dt) * P.beta;
  float px = x0_ped[0 * M + m], py = x0_ped[1 * M + m], vx = x0_ped[2 * M + m], vy = x0_ped[3 * M + m];
  for (int i = threadIdx.x; i < 2 * S; i += RATO_BLOCK) US[i] = us[i];
  __syncthreads();
  {
    // the fold of car_ego_kernel, statement for statement (one thread per step, fp64, rounded once)
    const double dt = P.dt;
    for (int t = threadIdx.x; t <= S; t += RATO_BLOCK) {
      double v = P.ego_init[2], ph = P.ego_init[3];
      for (int k = 0; k < t; ++k) {
        v += dt * (double)US[k * 2 + 0];
        ph += dt * (double)US[k * 2 + 1];
      }
      double sn, cs;
      sincos(ph, &sn, &cs);
      DV[t] = v;
      DCS[t] = cs;
      DSN[t] = sn;
    }
    __syncthreads();
    for (int t = threadIdx.x; t <= S; t += RATO_BLOCK) {
      double x = P.ego_init[0], y = P.ego_init[1];
      for (int k = 0; k < t; ++k) {
        x += dt * DV[k] * DCS[k];
        y += dt * DV[k] * DSN[k];
      }
      EGO[t * 2 + 0] = (float)x;
      EGO[t * 2 + 1] = (float)y;
    }
    __syncthreads();
  }
  if (!has_tile) return;
  // the ego positions of 64 steps live in the lanes of two registers (lane l: position at step 64 c + l + 1) and reach
  // the step as a v_readlane instead of an LDS read in front of every step
  float egx = 0.0f, egy = 0.0f;
  auto load_ego = [&](int t0) {
    const int t = (t0 + lane + 1 <= S) ? t0 + lane + 1 : S;
    egx = EGO[t * 2 + 0];
    egy = EGO[t * 2 + 1];
  };
  load_ego(0);
  float ex = EGO[0], ey = EGO[1];                 // ego position at the current step (carried)
  float zmax = -INFINITY;
  auto steps = [&](const float (&xi)[CEV_TB][2], int t0) {
#pragma unroll
    for (int i = 0; i < CEV_TB; ++i) {
      const int t = t0 + i;
      if (t < S) {
        const float nx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(egx), t & 63));
        const float ny = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(egy), t & 63));
        ped_step_exact(P, c, ex, ey, xi[i][0], xi[i][1], px, py, vx, vy);
        const float gt = separation_row_exact(P, nx, ny, px, py);
        ex = nx;
        ey = ny;
        zmax = fmaxf(zmax, gt);
        if (WANT_G && valid) g[(size_t)t * M + m] = gt;
      }
    }
  };
  for (int t0 = 0; t0 < S; t0 += 2 * CEV_TB) {
    if (t0 && (t0 & 63) == 0) load_ego(t0);
    steps(xa, t0);
    if (t0 + 2 * CEV_TB < S) load(xa, t0 + 2 * CEV_TB);
    steps(xb, t0 + CEV_TB);
    if (t0 + 3 * CEV_TB < S) load(xb, t0 + 3 * CEV_TB);
  }
  if (!Z) return;
  if (!tail.ws) {
    if (valid) Z[m] = zmax - P.tol;
    return;
  }
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

// separation_distances_at_all_times on given trajectories (driving.py:223-236): xs [S+1][8][M] -> dist [S][M]
__global__ __launch_bounds__(RATO_BLOCK) void car_distance_kernel(rato_car_params P, const float* __restrict__ xs,
                                                                 float* __restrict__ dist) {
  const size_t M = (size_t)P.M;
  const size_t m = (size_t)blockIdx.x * RATO_BLOCK + threadIdx.x;
  const int t = blockIdx.y;
  if (m >= M) return;
  const float* __restrict__ x = xs + (size_t)(t + 1) * 8 * M + m;
  const float dx = x[0] - x[4 * M], dy = x[M] - x[5 * M];
  dist[(size_t)t * M + m] = sqrtf(dx * dx + dy * dy) - P.d_min;
}

__device__ __forceinline__ int step_of(int k, int grp, int ngroups) {
  return k * ngroups + ((k & 1) ? (ngroups - 1 - grp) : grp);
}

// SPT = control steps per thread (each carries both controls: 2 x 4 sensitivities).
template <int SPT>
__global__ __launch_bounds__(RATO_BLOCK) void car_linearize_kernel(
    rato_car_params P, const float* __restrict__ dW, const float* __restrict__ x0_ped,
    const float* __restrict__ w_speed, const float* __restrict__ w_rep, const float* __restrict__ scratch,
    float* __restrict__ G, float* __restrict__ g_up, float* __restrict__ Z) {
  const size_t M = (size_t)P.M;
  const size_t m_raw = (size_t)blockIdx.x * RATO_BLOCK + threadIdx.x;
  const bool valid = m_raw < M;
  const size_t m = valid ? m_raw : M - 1;
  const int S = P.S, NC = 2 * S;
  const int grp = blockIdx.y, ngroups = gridDim.y;
  const bool lead = (grp == 0);
  const float* __restrict__ ego = scratch;
  const float* __restrict__ Eu = scratch + ego_off_Eu(S);
  const float* __restrict__ Epos = scratch + ego_off_Epos(S);
  PedConsts c;
  c.w_s = w_speed[m];
  c.w_r = w_rep[m];
  c.cn = sqrtf(P.dt) * P.beta;
  float px = x0_ped[0 * M + m], py = x0_ped[1 * M + m], vx = x0_ped[2 * M + m], vy = x0_ped[3 * M + m];
  // this workgroup's tile of the packed Jacobian: [n_pairs*2 rows][RATO_TILE lanes]
  float* __restrict__ Gt = G + (size_t)blockIdx.x * rato::packed_tile_stride((size_t)rato::pair_row_offset(S) * 2 * RATO_TILE);

  int stp[SPT];
#pragma unroll
  for (int k = 0; k < SPT; ++k) {
    const int s = step_of(k, grp, ngroups);
    stp[k] = (s < S) ? s : 0x7fffffff;
  }
  // sens[k][i] = d(ped px,py,vx,vy)_t / d u_{stp[k], i}
  float spx[SPT][2], spy[SPT][2], svx[SPT][2], svy[SPT][2];
#pragma unroll
  for (int k = 0; k < SPT; ++k)
#pragma unroll
    for (int i = 0; i < 2; ++i) spx[k][i] = spy[k][i] = svx[k][i] = svy[k][i] = 0.0f;
  float tpx = 0.f, tpy = 0.f, tvx = 0.f, tvy = 0.f;  // tangent along u

  float zmax = -INFINITY;
  float xi0 = dW[m], xi1 = dW[M + m];
  for (int t = 0; t < S; ++t) {
    const int tn = (t + 1 < S) ? t + 1 : t;
    const float nx0 = dW[(size_t)(tn * 2 + 0) * M + m], nx1 = dW[(size_t)(tn * 2 + 1) * M + m];
    float n0, n1, rinv;
    ped_step(P, c, ego[t * 4 + 0], ego[t * 4 + 1], xi0, xi1, px, py, vx, vy, n0, n1, rinv);
    // dt * w_r * H at time t
    const float k00 = P.dt * c.w_r * (1.0f - n0 * n0) * rinv;
    const float k01 = -P.dt * c.w_r * n0 * n1 * rinv;
    const float k11 = P.dt * c.w_r * (1.0f - n1 * n1) * rinv;
    const float ks = P.dt * c.w_s;
    {  // tangent along u, forced by Eu[t]
      const float ax = tpx - Eu[t * 2 + 0], ay = tpy - Eu[t * 2 + 1];
      const float npx = tpx + P.dt * tvx, npy = tpy + P.dt * tvy;
      const float nvx = tvx + (k00 * ax + k01 * ay) - ks * tvy;
      const float nvy = tvy + (k01 * ax + k11 * ay) - ks * tvy;
      tpx = npx; tpy = npy; tvx = nvx; tvy = nvy;
    }
#pragma unroll
    for (int k = 0; k < SPT; ++k) {
      if (stp[k] + 1 < t) {  // wave-uniform: E_t[pos, (s,i)] is nonzero only for t >= s+2
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int col = stp[k] * 2 + i;
          const float ax = spx[k][i] - Epos[(size_t)(t * 2 + 0) * NC + col];
          const float ay = spy[k][i] - Epos[(size_t)(t * 2 + 1) * NC + col];
          const float npx = spx[k][i] + P.dt * svx[k][i], npy = spy[k][i] + P.dt * svy[k][i];
          const float nvx = svx[k][i] + (k00 * ax + k01 * ay) - ks * svy[k][i];
          const float nvy = svy[k][i] + (k01 * ax + k11 * ay) - ks * svy[k][i];
          spx[k][i] = npx; spy[k][i] = npy; svx[k][i] = nvx; svy[k][i] = nvy;
        }
      }
    }
    // row t: g_t = -(|p_e - p_p|_{t+1} - d_min), gradient -n_{t+1}
    const float dx = ego[(t + 1) * 4 + 0] - px, dy = ego[(t + 1) * 4 + 1] - py;
    const float r = sqrtf(dx * dx + dy * dy);
    const float m0 = dx / r, m1 = dy / r;
    const float gt = -(r - P.d_min);
    zmax = fmaxf(zmax, gt);
    if (lead && valid) {
      const float dirv = -(m0 * (Eu[(t + 1) * 2 + 0] - tpx) + m1 * (Eu[(t + 1) * 2 + 1] - tpy));
      g_up[(size_t)t * M + m] = P.rows_out ? gt : (-gt + dirv);  // driving.py:295; rows_out = 1: g itself
    }
    const int row = rato::pair_row_offset(t);
#pragma unroll
    for (int k = 0; k < SPT; ++k) {
      if (stp[k] < t) {  // wave-uniform
        if (valid) {
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            const int col = stp[k] * 2 + i;
            const float e0 = Epos[(size_t)((t + 1) * 2 + 0) * NC + col];
            const float e1 = Epos[(size_t)((t + 1) * 2 + 1) * NC + col];
            Gt[((row + stp[k]) * 2 + i) * RATO_TILE + (int)threadIdx.x] =
                -(m0 * (e0 - spx[k][i]) + m1 * (e1 - spy[k][i]));
          }
        }
      }
    }
    xi0 = nx0; xi1 = nx1;
  }
  if (lead && valid && Z) Z[m] = zmax - P.tol;
}

// ---------------------------------------------------------------------------
// Row-parallel (adjoint) linearization for the driving problem — default.
// Same structure as drone_linearize_rows_kernel: a workgroup owns 64 samples;
//   phase 0  stage the noise tile, the controls and the (sample-independent) ego tables in LDS;
//   phase 1  wave 0 rolls the 4-state pedestrian out and leaves K_k = dt w_r (I - n n^T)/r (3 floats)
//            and the pedestrian position q_{k+1} (2 floats) in LDS: 20 B per sample-step, publishing
//            its progress in LDS after every step, then joins phase 2;
//   phase 2  (overlapped with phase 1: row t only needs steps 0..t; rows are taken shortest first)
//            all waves pull row tasks from an LDS queue and sweep row t with the 8-state adjoint
//            eta_{t+1} = grad_x g_t = (-n, 0, 0, +n, 0, 0),  eta_k = eta_{k+1} J_k,
//            d g_t / d u_{s,i} = dt * eta_{s+1}[2 + i]   (u_0 drives v_ego, u_1 drives phi_ego),
//            accumulating the row's dot product with u for g_up (driving.py:295).
// The forward/column kernel above re-rolls the pedestrian once per column group and carries
// 8 registers per control step; it stays as the fallback when the LDS tables do not fit.
#ifndef RATO_CDIAG
#define RATO_CDIAG 0             // diagnostic builds: 1 no Jacobian stores; 4 phase times of every workgroup into g_up; 5 both;
#endif                           // 6 the timeline of every workgroup (tools/car_timeline.py)
#define RATO_CDIAG_PHASES (RATO_CDIAG == 4 || RATO_CDIAG == 5)
#ifndef RATO_CRAMP
#define RATO_CRAMP 1             // A/B: 0 = the launch ramp of round 3 (first tile's noise requested after the ego tables, pedestrian
#endif                           // state requested at the rollout, progress published every step, rollout wave at default priority)
#ifndef RATO_CROLL_PUBLISH
#define RATO_CROLL_PUBLISH (RATO_CRAMP ? 4 : 1)   // the rollout wave publishes its progress every this many steps
#endif
#ifndef RATO_CROWS_NW
#define RATO_CROWS_NW 8          // waves per workgroup (A/B builds: tools/ab.sh ... -DRATO_CROWS_NW=16)
#endif
constexpr int CROWS_NW = RATO_CROWS_NW;
constexpr int CROWS_SAMPLES = 64;

typedef float cfloat2_t __attribute__((ext_vector_type(2)));
typedef float cfloat4_t __attribute__((ext_vector_type(4)));

__host__ __device__ inline size_t car_rows_lds_floats(int S) {
  // K4 float4 + QP float2 per (k, lane) | EGOP float2[S+1] | EC float4[S] | EC2 float4[S] | US float2[S] | head (+pad)
  // | ego v, phi [S+1] each | 8 x 4 reduction slots (final rows, workgroup 0) | fp64 v, cos, sin [S+1] each
  return (size_t)S * CROWS_SAMPLES * 6 + (size_t)(S + 1) * 2 + (size_t)S * 8 + (size_t)S * 2 + 4 +
         (size_t)(S + 1) * 2 + CROWS_NW * 4 + (size_t)(S + 1) * 6 + 2;
}

// LOOP: several tiles per workgroup through the global tile queue (large batches).
// PHILOX: the tile's noise is regenerated while it is staged (the numbers rato_car_sample would have written) instead
// of read: no noise array, no reads in the middle of the store stream.
template <bool LOOP, bool PHILOX = false>
__global__ __launch_bounds__(CROWS_NW* RATO_WAVE) void car_linearize_rows_kernel(
    rato_car_params P, uint64_t seed, float noise_scale, const float* __restrict__ us, const float* __restrict__ dW,
    const float* __restrict__ x0_ped, const float* __restrict__ w_speed, const float* __restrict__ w_rep,
    float* __restrict__ final_du, float* __restrict__ final_rhs, float* __restrict__ G, float* __restrict__ g_up,
    float* __restrict__ Z, int n_tiles_total, unsigned* __restrict__ tile_queue, int split, int n_whole,
    const rato_sel::StatsTail tail, int flags) {
  extern __shared__ __attribute__((aligned(16))) unsigned char car_lds_raw[];
  // Statistics in the same launch (rato_saa.h: params.stats_*): the workgroups behind the producer's own wait until every
  // tile's Z has been counted in, then run the exact selection on it -- beside the workgroups still storing the Jacobian.
  const int n_prod = tail.ws ? tail.n_prod : (int)gridDim.x;
  if (tail.is_stats((int)blockIdx.x)) {
    rato_sel::stats_tail_run<CROWS_NW * RATO_WAVE>(tail, Z, (long)P.M, car_lds_raw);
    return;
  }
  const int pbid = (int)blockIdx.x;
  const bool noise_tiled = (flags & 1) != 0;   // dW is the re-tiled copy (rato_car_tile_noise)
  const bool nt_stores = (flags & 2) != 0;     // the Jacobian goes out as streaming (non-temporal) stores (see drone.hip)
  unsigned* const z_signal = tail.ws ? tail.ws->sig : nullptr;
  const size_t M = (size_t)P.M;
  const int S = P.S;
  constexpr int NT = CROWS_NW * RATO_WAVE;
  const int lane = threadIdx.x & (RATO_WAVE - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / RATO_WAVE);   // scalar: wave-level branches stay scalar
  // (no __restrict__: QP[t] holds step t's noise until the rollout overwrites it with q_{t+1})
  cfloat4_t* EC = reinterpret_cast<cfloat4_t*>(car_lds_raw);                 // [S] (dt c, dt s, -dt v s, dt v c)
  cfloat4_t* EC2 = EC + S;                                                   // [S] -dt (EC.x, EC.z | EC.y, EC.w): the row sweep's form
  cfloat4_t* K4 = EC2 + S;                                                   // [S][64] (k00, k01 | k01, k11)
  cfloat2_t* QP = reinterpret_cast<cfloat2_t*>(K4 + (size_t)S * CROWS_SAMPLES);   // [S][64] q_{k+1}
  cfloat2_t* EGOP = QP + (size_t)S * CROWS_SAMPLES;                          // [S+1] ego position
  cfloat2_t* US = EGOP + (S + 1);                                            // [S]
  int* head = reinterpret_cast<int*>(US + S);
  float* SV = reinterpret_cast<float*>(head + 4);                            // [S+1] ego speed
  float* SPH = SV + (S + 1);                                                 // [S+1] ego heading
  float* RED = SPH + (S + 1);                                                // [CROWS_NW][4]
  double* DV = reinterpret_cast<double*>(car_lds_raw + (((reinterpret_cast<unsigned char*>(RED + CROWS_NW * 4) - car_lds_raw) + 7) & ~size_t(7)));
  double* DCS = DV + (S + 1);                                                // fp64 speed | cos | sin of the ego
  double* DSN = DCS + (S + 1);

  // ---- once per workgroup: the controls and the (sample-independent) ego tables.  With ONE tile per workgroup
  // (small batches, latency bound) the noise tile is requested first and the tables are computed while those loads
  // are in flight; with the tile loop they are requested per tile (keeping 16 more registers live across the fp64 ego
  // code would cost a workgroup per CU).
  constexpr int MAXR = 16;
  const int nrows = 2 * S;
  float tmp0[MAXR];   // batch 0 of the noise rows (S = 40: all 80 rows)
  constexpr int MAX4 = 4;             // the tiled form: 16-byte requests per thread and batch (S <= 64: one batch)
  cfloat4_t pre4[MAX4];
  const int n4 = nrows * (CROWS_SAMPLES / 4);
  bool first_unit = true;   // (scalar) the first unit's noise is requested here, in front of the ego tables
  if ((!LOOP || RATO_CRAMP) && !PHILOX) {
    const size_t tile0 = (size_t)((LOOP && pbid < n_whole) ? pbid : (LOOP ? n_whole + (pbid - n_whole) / split : pbid / split));
    const size_t mr = tile0 * CROWS_SAMPLES + lane;
    const size_t mm = mr < M ? mr : M - 1;
    if (noise_tiled) {
      // noise_tiled: dW is [tile] blocks of 2S x 64 floats (rato_car_tile_noise), each the image this kernel keeps in LDS
      // ([S][64] (xi_0, xi_1) pairs): ONE contiguous block per tile instead of 2S rows of 256 B that lie M floats apart
      // (reads beside the store stream: DESIGN.md 4.2), staged as a straight 16-byte-per-lane copy
      const cfloat4_t* __restrict__ src4 = reinterpret_cast<const cfloat4_t*>(dW + tile0 * nrows * CROWS_SAMPLES);
#pragma unroll
      for (int i = 0; i < MAX4; ++i) {
        const int idx = (int)threadIdx.x + i * NT;
        pre4[i] = src4[idx < n4 ? idx : 0];
      }
    } else {
#pragma unroll
    for (int i = 0; i < MAXR; ++i) {
      const int r = wave + i * CROWS_NW;
      tmp0[i] = dW[(size_t)((r < nrows) ? r : 0) * M + mm];
    }
    }
  }
  for (int t = threadIdx.x; t < S; t += NT) {
    cfloat2_t u2;
    u2.x = us[t * 2 + 0];
    u2.y = us[t * 2 + 1];
    US[t] = u2;
  }
  __syncthreads();
  {
    // ego trajectory (driving.py:166-173), one thread per step, folded in fp64 and rounded once: the ego is
    // sample independent, so this costs a microsecond per workgroup and keeps the 40-step accumulation error of
    // the positions (x ~ 20 m, fp32 ulp 2e-6) out of every sample's distance / normal / Jacobian
    const double dt = P.dt;
    for (int t = threadIdx.x; t <= S; t += NT) {
      double v = P.ego_init[2], ph = P.ego_init[3];
      for (int k = 0; k < t; ++k) {
        const cfloat2_t u2 = US[k];
        v += dt * (double)u2.x;
        ph += dt * (double)u2.y;
      }
      double sn, cs;
      sincos(ph, &sn, &cs);   // once per step
      DV[t] = v;
      DCS[t] = cs;
      DSN[t] = sn;
      SV[t] = (float)v;
      SPH[t] = (float)ph;
      if (t < S) {
        cfloat4_t c;
        c.x = (float)(dt * cs);
        c.y = (float)(dt * sn);
        c.z = (float)(-dt * v * sn);
        c.w = (float)(dt * v * cs);
        EC[t] = c;
        cfloat4_t c2;                       // the sweep propagates E = dt (eta_v, eta_phi) and eta_p_ego = -eta_q:
        c2.x = (float)(-dt * dt * cs);      //   E -= dt (q_x (c.x, c.z) + q_y (c.y, c.w))
        c2.y = (float)(dt * dt * v * sn);
        c2.z = (float)(-dt * dt * sn);
        c2.w = (float)(-dt * dt * v * cs);
        EC2[t] = c2;
      }
    }
    __syncthreads();
    for (int t = threadIdx.x; t <= S; t += NT) {
      double x = P.ego_init[0], y = P.ego_init[1];
      for (int k = 0; k < t; ++k) {
        x += dt * DV[k] * DCS[k];
        y += dt * DV[k] * DSN[k];
      }
      cfloat2_t e;
      e.x = (float)x;
      e.y = (float)y;
      EGOP[t] = e;
    }
  }
  __syncthreads();

  // tile_queue != NULL (large batches): the grid fills every workgroup slot once and a workgroup that has finished a
  // tile takes the next one from a global counter, so that XCDs that run this store stream faster take more tiles
  // (see drone_linearize_rows_kernel); the sample-independent ego tables are built once per workgroup.
  // split > 1 (small batches, !LOOP): every tile is dealt out to `split` workgroups that each build the tables and take
  // the row tasks congruent to their part (mod split), so that a batch of a few hundred tiles still fills the chip.
  // LOOP with split > 1: the queue hands the first n_whole tiles out whole and the LAST ones as `split` row-interleaved
  // parts each (shorter last units shorten the drain of the launch, as in the drone kernel).
  const int n_units = LOOP ? n_whole + (n_tiles_total - n_whole) * split : n_tiles_total * split;
#if RATO_CDIAG_PHASES   // diagnostic builds 4 / 5 (tools/car_phases.py): where does a tile's time go?  100 MHz ticks, thread 0 / wave 0
  unsigned long long dg_t0 = wall_clock64(), dg_prologue = 0, dg_stage = 0, dg_roll = 0, dg_rows = 0, dg_next = 0, dg_mark = 0;
  int dg_tiles = 0;
  dg_mark = wall_clock64();
#endif
#if RATO_CDIAG == 6   // per tile: start, rows-after-the-rollout start, end (100 MHz ticks, low 24 bits), flushed into g_up at the end
  __shared__ unsigned dg_tl[64];
  int dg_n = 0;
#endif
  for (int unit = pbid; unit < n_units;) {
#if RATO_CDIAG == 6
  if (threadIdx.x == 0 && dg_n < 20) dg_tl[1 + 3 * dg_n] = (unsigned)wall_clock64();
#endif
  int tile, part_id, row_split;
  if (LOOP && unit < n_whole) {
    tile = unit;
    part_id = 0;
    row_split = 1;
  } else {
    const int v = LOOP ? unit - n_whole : unit;
    tile = (LOOP ? n_whole : 0) + v / split;
    part_id = v - (v / split) * split;
    row_split = split;
  }
  const size_t m_raw = (size_t)tile * CROWS_SAMPLES + lane;
  const bool valid = m_raw < M;
  const size_t m = valid ? m_raw : M - 1;
  const float w_s = w_speed[m], w_r = w_rep[m];
  const float ks = P.dt * w_s;
  // the pedestrian's initial state, requested by the rollout wave HERE so that it arrives under the staging of the noise
  // tile (requested at the rollout it cost a round trip to memory per tile, behind the workgroup's own row stores)
  float ped0[4] = {0.f, 0.f, 0.f, 0.f};
  if (RATO_CRAMP && wave == 0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) ped0[i] = x0_ped[(size_t)i * M + m];
  }

  // ---- phase 0: queue state and the noise tile (the ego tables were built by this workgroup before the loop: the
  // separate one-workgroup ego prologue launch, 10-30 us in front of every linearize call, is gone)
  {
    if (threadIdx.x == 0) {
      head[0] = 0;
      head[1] = 0;   // rollout progress: number of finished steps
    }
  }
  __syncthreads();
  {
    float* QPf = reinterpret_cast<float*>(QP);
    if (PHILOX) {
      for (int t = wave; t < S; t += CROWS_NW) {
        const rato::u32x4 rr = rato::philox_at(seed, rato::PHILOX_STREAM_DW, (uint32_t)t, (uint64_t)m);
        float x0, x1;
        rato::box_muller(rr.x, rr.y, x0, x1);
        QPf[(t * CROWS_SAMPLES + lane) * 2 + 0] = x0 * noise_scale;
        QPf[(t * CROWS_SAMPLES + lane) * 2 + 1] = x1 * noise_scale;
      }
    }
    if (!PHILOX && noise_tiled) {
      const cfloat4_t* __restrict__ src4 = reinterpret_cast<const cfloat4_t*>(dW + (size_t)tile * nrows * CROWS_SAMPLES);
      cfloat4_t* dst4 = reinterpret_cast<cfloat4_t*>(QPf);
      if (!((!LOOP || RATO_CRAMP) && first_unit)) {
#pragma unroll
        for (int i = 0; i < MAX4; ++i) {
          const int idx = (int)threadIdx.x + i * NT;
          pre4[i] = src4[idx < n4 ? idx : 0];
        }
      }
#pragma unroll
      for (int i = 0; i < MAX4; ++i) {
        const int idx = (int)threadIdx.x + i * NT;
        if (idx < n4) dst4[idx] = pre4[i];
      }
      for (int b0 = (int)threadIdx.x + MAX4 * NT; b0 < n4; b0 += MAX4 * NT) {   // long horizons: further batches
        cfloat4_t t4[MAX4];
#pragma unroll
        for (int i = 0; i < MAX4; ++i) {
          const int idx = b0 + i * NT;
          t4[i] = src4[idx < n4 ? idx : 0];
        }
#pragma unroll
        for (int i = 0; i < MAX4; ++i) {
          const int idx = b0 + i * NT;
          if (idx < n4) dst4[idx] = t4[i];
        }
      }
    }
    if (LOOP && !PHILOX && !noise_tiled && !(RATO_CRAMP && first_unit)) {
#pragma unroll
      for (int i = 0; i < MAXR; ++i) {
        const int r = wave + i * CROWS_NW;
        tmp0[i] = dW[(size_t)((r < nrows) ? r : 0) * M + m];
      }
    }
    first_unit = false;
    if (!PHILOX && !noise_tiled) {
#pragma unroll
      for (int i = 0; i < MAXR; ++i) {
        const int r = wave + i * CROWS_NW;
        if (r < nrows) QPf[((r >> 1) * CROWS_SAMPLES + lane) * 2 + (r & 1)] = tmp0[i];
      }
    }
    for (int r0 = wave + CROWS_NW * MAXR; !PHILOX && !noise_tiled && r0 < nrows; r0 += CROWS_NW * MAXR) {   // long horizons: further batches
      float tmp[MAXR];
#pragma unroll
      for (int i = 0; i < MAXR; ++i) {
        const int r = r0 + i * CROWS_NW;
        tmp[i] = (r < nrows) ? dW[(size_t)r * M + m] : 0.0f;
      }
#pragma unroll
      for (int i = 0; i < MAXR; ++i) {
        const int r = r0 + i * CROWS_NW;
        if (r < nrows) QPf[((r >> 1) * CROWS_SAMPLES + lane) * 2 + (r & 1)] = tmp[i];
      }
    }
  }
  __syncthreads();
#if RATO_CDIAG_PHASES
  if (threadIdx.x == 0) {
    const unsigned long long now = wall_clock64();
    dg_stage += now - dg_mark;
    dg_mark = now;
    ++dg_tiles;
  }
#endif
  // final rows (sample independent: driving.py:283-288, :311): workgroup 0 propagates one control column per thread
  if (tile == 0 && part_id == 0 && (final_du || final_rhs)) {
    const int NC = 2 * S;
    float rhs_acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int c = threadIdx.x; c < NC; c += NT) {
      const int s = c >> 1, i = c & 1;
      float e0 = 0.f, e1 = 0.f, e2 = 0.f, e3 = 0.f;
      for (int t = 0; t < S; ++t) {
        const cfloat4_t k = EC[t];
        const float n0 = e0 + k.x * e2 + k.z * e3;
        const float n1 = e1 + k.y * e2 + k.w * e3;
        float n2 = e2, n3 = e3;
        if (t == s) {
          if (i == 0) n2 += P.dt; else n3 += P.dt;
        }
        e0 = n0; e1 = n1; e2 = n2; e3 = n3;
      }
      if (final_du) {
        final_du[0 * NC + c] = e0; final_du[1 * NC + c] = e1; final_du[2 * NC + c] = e2; final_du[3 * NC + c] = e3;
      }
      const float uc = (i == 0) ? US[s].x : US[s].y;
      rhs_acc[0] += e0 * uc; rhs_acc[1] += e1 * uc; rhs_acc[2] += e2 * uc; rhs_acc[3] += e3 * uc;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float sres = rato::wave_sum(rhs_acc[r]);
      if (lane == 0) RED[wave * 4 + r] = sres;
    }
    __syncthreads();
    if (final_rhs && threadIdx.x < 4) {
      float acc = 0.f;
#pragma unroll
      for (int w = 0; w < CROWS_NW; ++w) acc += RED[w * 4 + threadIdx.x];
      const float xS = (threadIdx.x == 0) ? EGOP[S].x : ((threadIdx.x == 1) ? EGOP[S].y
                                                        : ((threadIdx.x == 2) ? SV[S] : SPH[S]));
      final_rhs[threadIdx.x] = -(xS - P.ego_goal[threadIdx.x]) + acc;   // driving.py:288
    }
  }

  // ---- phase 1 (wave 0) overlapped with phase 2 (the other waves, then everybody): row t only needs steps
  // 0..t, so the rows are swept shortest first behind the rollout, which publishes its progress in LDS.
  typedef __attribute__((address_space(3))) volatile int lds_vint;
  lds_vint* prog = (lds_vint*)(head + 1);
  if (wave == 0) {
    PedConsts c;
    c.w_s = w_s;
    c.w_r = w_r;
    c.cn = sqrtf(P.dt) * P.beta;
    float px, py, vx, vy;
    if (RATO_CRAMP) {
      px = ped0[0]; py = ped0[1]; vx = ped0[2]; vy = ped0[3];
      __builtin_amdgcn_s_setprio(3);   // the rollout is the critical path of a tile: its wave wins issue against the sweeps
    } else {
      px = x0_ped[0 * M + m]; py = x0_ped[1 * M + m]; vx = x0_ped[2 * M + m]; vy = x0_ped[3 * M + m];
    }
    cfloat2_t xi = QP[lane], e = EGOP[0];
    for (int t = 0; t < S; ++t) {
      const int slot = t * CROWS_SAMPLES + lane;
      const int tn = (t + 1 < S) ? t + 1 : t;   // next step's inputs are fetched before this step's table stores
      const cfloat2_t xi_n = QP[tn * CROWS_SAMPLES + lane], e_n = EGOP[tn];
      float n0, n1, rinv;
      ped_step(P, c, e.x, e.y, xi.x, xi.y, px, py, vx, vy, n0, n1, rinv);
      const float kr = P.dt * w_r * rinv;  // dt w_r (I - n n^T)/r at state t
      cfloat4_t kk;
      kk.x = kr * (1.0f - n0 * n0);
      kk.y = -kr * n0 * n1;
      kk.z = kk.y;
      kk.w = kr * (1.0f - n1 * n1);
      K4[slot] = kk;
      cfloat2_t q;
      q.x = px;
      q.y = py;
      QP[slot] = q;
      if (RATO_CROLL_PUBLISH == 1 || ((t + 1) % RATO_CROLL_PUBLISH) == 0 || t + 1 == S) {   // (scalar condition)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // waits for the table stores: not on every step
        if (lane == 0) *prog = t + 1;
      }
      xi = xi_n;
      e = e_n;
    }
    if (RATO_CRAMP) __builtin_amdgcn_s_setprio(0);
    // Z = max_t g_t - tol from the q table (driving.py:630-638), by the rollout wave itself as soon as the rollout is
    // done -- BEFORE the rows are swept, so that the statistics workgroups of this launch (params.stats_*) can select on
    // Z while the Jacobian is still being stored (until round 4: the LAST task of the row queue).  Row-split parts: the
    // part that owned that task writes.
    if (Z && (S % row_split) == part_id) {
      float zmax = -INFINITY;
      for (int t = 0; t < S; ++t) {
        const cfloat2_t q = QP[t * CROWS_SAMPLES + lane], e = EGOP[t + 1];
        const float dx = e.x - q.x, dy = e.y - q.y;
        zmax = fmaxf(zmax, -(sqrtf(dx * dx + dy * dy) - P.d_min));
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
    }
#if RATO_CDIAG_PHASES
    if (threadIdx.x == 0) {
      const unsigned long long now = wall_clock64();
      dg_roll += now - dg_mark;
      dg_mark = now;
    }
#endif
#if RATO_CDIAG == 6
    if (threadIdx.x == 0 && dg_n < 20) dg_tl[2 + 3 * dg_n] = (unsigned)wall_clock64();
#endif
  }

  // ---- phase 2: row tasks in ascending order (task S: Z = max_t g_t - tol from the q table)
  auto wait_steps = [&](int need) {
    while (*prog < need) __builtin_amdgcn_s_sleep(4);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  };
  constexpr int RT = CROWS_SAMPLES;
  const size_t tile_floats = rato::packed_tile_stride((size_t)rato::pair_row_offset(S) * 2 * RT);
  float* __restrict__ Gt = G + (size_t)tile * tile_floats;   // wave-uniform: the stores take it as a scalar base + lane
  auto next_task = [&]() -> int {
    int v = 0;
    if (lane == 0) v = atomicAdd(head, 1);
    return part_id + row_split * __builtin_amdgcn_readfirstlane(v);
  };
  int task = next_task();
  while (task < S) {
    {
      const int t = task;
      wait_steps(t + 1);
      const cfloat2_t q = QP[t * CROWS_SAMPLES + lane], e1 = EGOP[t + 1];
      const float dx = e1.x - q.x, dy = e1.y - q.y;
      const float r = sqrtf(dx * dx + dy * dy);
      const float n0 = dx / r, n1 = dy / r;
      const float gt = -(r - P.d_min);                       // driving.py:223-230,269
      // eta = (eta_p_ego, eta_v, eta_phi | eta_q, eta_qv); eta_p_ego = -eta_q throughout (both start at -+n and
      // receive -+ the same increment), so only q = eta_q is carried; E = dt (eta_v, eta_phi) IS the Jacobian entry.
      // Everything is kept in pairs: the compiler issues packed fp32 math (v_pk_fma_f32 / v_pk_mul_f32 /
      // v_pk_add_f32), 9 vector instructions per step for the two stored numbers (the scalar form took 20).
      cfloat2_t q2, qv2 = {0.0f, 0.0f}, E2 = {0.0f, 0.0f}, acc2 = {0.0f, 0.0f};
      q2.x = n0;
      q2.y = n1;
      float* __restrict__ Grow = Gt + (size_t)rato::pair_row_offset(t) * (2 * RT);
      for (int k = t; k >= 1; --k) {
        const int slot = k * CROWS_SAMPLES + lane;
        const cfloat4_t kk = K4[slot];
        const cfloat4_t c = EC2[k];
        const cfloat2_t u2 = US[k - 1];
        cfloat2_t f2 = qv2.y * kk.zw;                                   // (eta_qv) K
        f2 = qv2.x * kk.xy + f2;
        E2 = q2.y * c.zw + E2;                                          // uses eta_q BEFORE its update
        E2 = q2.x * c.xy + E2;
        cfloat2_t nqv2 = P.dt * q2 + qv2;
        nqv2.y -= ks * (qv2.x + qv2.y);
        q2 += f2;
        qv2 = nqv2;
        acc2 = E2 * u2 + acc2;                                          // d g_t / d u_{k-1, 0|1} = E2
#if RATO_CDIAG == 1 || RATO_CDIAG == 5   // diagnostic builds: everything but the Jacobian stores (the condition is never true)
        if (valid && E2.x == 123.456f) {
#else
        if (valid) {
#endif
          float* __restrict__ o = Grow + (k - 1) * (2 * RT);
          if (nt_stores) {   // streaming: the Jacobian is never read back here; the memory-side cache keeps the batch's inputs
            rato::store_streaming<0>(o + lane, E2.x);
            rato::store_streaming<RT * 4>(o + lane, E2.y);
          } else {
            o[lane] = E2.x;
            o[RT + lane] = E2.y;
          }
        }
      }
      const float acc = acc2.x + acc2.y;
      if (valid) g_up[(size_t)t * M + m] = P.rows_out ? gt : (-gt + acc);        // driving.py:295; rows_out = 1: g itself
    }
    task = next_task();
  }
  // ---- next tile
#if RATO_CDIAG == 6
  if (!LOOP) {   // (one unit per workgroup: its end stamp, behind a barrier so that every wave's rows are in)
    __syncthreads();
    if (threadIdx.x == 0 && dg_n < 20) dg_tl[3 + 3 * dg_n] = (unsigned)wall_clock64();
    ++dg_n;
  }
#endif
#if RATO_CDIAG_PHASES
  if (!LOOP) {
    __syncthreads();
    if (threadIdx.x == 0) {
      const unsigned long long now = wall_clock64();
      dg_rows += now - dg_mark;
      dg_mark = now;
    }
  }
#endif
  if (!LOOP) break;
  __syncthreads();   // every wave has finished this tile's rows: the sample tables are dead, head[] may be rewritten
#if RATO_CDIAG == 6
  if (threadIdx.x == 0 && dg_n < 20) dg_tl[3 + 3 * dg_n] = (unsigned)wall_clock64();
  ++dg_n;
#endif
#if RATO_CDIAG_PHASES
  if (threadIdx.x == 0) {
    const unsigned long long now = wall_clock64();
    dg_rows += now - dg_mark;
    dg_mark = now;
  }
#endif
  if (threadIdx.x == 0)
    head[2] = n_prod + (int)__hip_atomic_fetch_add(tile_queue, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  unit = head[2];
#if RATO_CDIAG_PHASES
  if (threadIdx.x == 0) {
    const unsigned long long now = wall_clock64();
    dg_next += now - dg_mark;
    dg_mark = now;
  }
#endif
  }  // unit loop
#if RATO_CDIAG_PHASES
  if (threadIdx.x == 0 && (size_t)(blockIdx.x + 1) * 8 <= (size_t)S * M) {
    float* o = g_up + (size_t)blockIdx.x * 8;
    o[0] = (float)dg_tiles; o[1] = (float)dg_stage; o[2] = (float)dg_roll; o[3] = (float)dg_rows; o[4] = (float)dg_next;
    o[5] = (float)(wall_clock64() - dg_t0); o[6] = (float)dg_prologue; o[7] = 0.f;
  }
#endif
#if RATO_CDIAG == 6
  __syncthreads();
  if (threadIdx.x == 0) dg_tl[0] = (unsigned)dg_n;
  __syncthreads();
  if (threadIdx.x < 64 && (size_t)(blockIdx.x + 1) * 64 <= (size_t)S * M)
    g_up[(size_t)blockIdx.x * 64 + threadIdx.x] = (float)(dg_tl[threadIdx.x] & 0xffffffu);
#endif
  if (LOOP && threadIdx.x == 0) {   // the workgroup that leaves last zeroes the queue for the next launch
    const unsigned gone = __hip_atomic_fetch_add(tile_queue + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (gone == (unsigned)n_prod - 1u) {
      __hip_atomic_store(tile_queue, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(tile_queue + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}


bool params_ok(const rato_car_params* p) {
  return p && p->M > 0 && p->S > 0 && p->S <= 1024 && p->dt > 0.0f;   // 1024: the ego prologue's LDS tables (48 KB)
}

template <int SPT>
int launch_car_linearize(const rato_car_params* p, const float* dW, const float* x0_ped, const float* w_speed,
                         const float* w_rep, const float* scratch, float* G, float* g_up, float* Z,
                         hipStream_t st) {
  const int ngroups = (p->S + SPT - 1) / SPT;
  dim3 grid(rato::nblocks_for(p->M), ngroups), block(RATO_BLOCK);
  hipLaunchKernelGGL(car_linearize_kernel<SPT>, grid, block, 0, st, *p, dW, x0_ped, w_speed, w_rep, scratch, G,
                     g_up, Z);
  RATO_LAUNCH_CHECK();
  return RATO_OK;
}

}  // namespace

extern "C" size_t rato_car_ego_scratch_floats(int32_t S) { return S > 0 ? ego_total(S) : 0; }

namespace {
// (beyond ~1e6 samples the launch is bandwidth bound either way and the plain kernel behind the ego prologue is a few
//  per cent faster: M = 1e7 0.614 against 0.650 ms; the driving noise has no unused row to skip)
constexpr int64_t CAR_EVAL_TILES_MAX_M = 1 << 20, CAR_EVAL_STATS_IN_LAUNCH_MAX_M = 65536;
int car_eval_tiles_max_m() {
  static const int64_t v = [] { const char* e = getenv("RATO_EVAL_TILES_MAX_M"); return e ? (int64_t)atoll(e) : CAR_EVAL_TILES_MAX_M; }();
  return (int)v;
}
}  // namespace

extern "C" int rato_car_eval_stats_in_launch(int32_t M) {
  return M > 0 && M <= CAR_EVAL_STATS_IN_LAUNCH_MAX_M && M <= car_eval_tiles_max_m();
}

namespace {
int car_eval_tiles_launch(const rato_car_params* p, int K, const float* us, const float* dW, const float* x0_ped,
                          const float* w_speed, const float* w_rep, float* Z, int64_t ldz, float* g,
                          const rato_sel::StatsTail& tail, int grid_launch, size_t lds, int n_tiles, hipStream_t st) {
  if (g)
    hipLaunchKernelGGL(car_eval_tiles_kernel<true>, dim3(grid_launch, K), dim3(RATO_BLOCK), lds, st, *p, us, dW, x0_ped, w_speed,
                       w_rep, Z, (long)ldz, g, n_tiles, tail);
  else
    hipLaunchKernelGGL(car_eval_tiles_kernel<false>, dim3(grid_launch, K), dim3(RATO_BLOCK), lds, st, *p, us, dW, x0_ped, w_speed,
                       w_rep, Z, (long)ldz, g, n_tiles, tail);
  RATO_LAUNCH_CHECK();
  return RATO_OK;
}
bool car_eval_tiles_ok(const rato_car_params* p) {
  return p->M <= car_eval_tiles_max_m() && car_eval_tiles_lds_bytes(p->S) <= 48 * 1024;
}
}  // namespace

extern "C" int rato_car_eval(const rato_car_params* p, const float* us, const float* dW, const float* x0_ped,
                             const float* w_speed, const float* w_rep, float* ego_scratch, float* Z, float* xs,
                             float* g, void* stream) {
  RATO_CLEAR_ERROR();
  if (!params_ok(p) || !us || !dW || !x0_ped || !w_speed || !w_rep || !ego_scratch) return RATO_EINVAL;
  if (p->stats_workspace && (!Z || !p->stats_out || !(p->stats_alpha > 0.0) || !(p->stats_alpha <= 1.0))) return RATO_EINVAL;
  hipStream_t st = rato::as_stream(stream);
  if (!xs && car_eval_tiles_ok(p)) {
    // no trajectories wanted: ONE launch for the rollout (ego tables in every workgroup, tiled rollout); the statistics
    // behind it (in the launch only on request: see rato_drone_eval)
    const int n_tiles = (int)((p->M + RATO_WAVE - 1) / RATO_WAVE);
    const int grid = (n_tiles + CEV_NW - 1) / CEV_NW;
    rato_sel::StatsTail tail = {};
    int grid_launch = grid;
    size_t lds = car_eval_tiles_lds_bytes(p->S);
    const bool in_launch = p->stats_workspace && (p->stats_flags & RATO_STATS_IN_LAUNCH) && rato_car_eval_stats_in_launch(p->M);
    if (in_launch) {
      int Gs = 0;
      const int extra = rato_sel::stats_tail_workgroups<RATO_BLOCK>(p->M, Gs);
      if (extra < 0) return RATO_EINVAL;
      tail.ws = static_cast<rato_sel::Workspace*>(p->stats_workspace);
      tail.out = p->stats_out;
      tail.alpha = p->stats_alpha;
      tail.thr = p->stats_thr;
      tail.G = Gs;
      tail.n_prod = grid;
      rato_sel::stats_rank(p->M, p->stats_alpha, tail.k, tail.var_is_max);
      grid_launch = grid + extra;
      if (lds < rato_sel::rs_body_lds_bytes<RATO_BLOCK>()) lds = rato_sel::rs_body_lds_bytes<RATO_BLOCK>();
    }
    const int rc = car_eval_tiles_launch(p, 1, us, dW, x0_ped, w_speed, w_rep, Z, p->M, g, tail, grid_launch, lds, n_tiles, st);
    if (rc != RATO_OK) return rc;
    if (p->stats_workspace && !in_launch)
      return rato_risk_stats(Z, p->M, p->stats_alpha, p->stats_thr, p->stats_workspace,
                             rato_risk_stats_workspace_bytes(p->M), p->stats_out, stream);
    return RATO_OK;
  }
  hipLaunchKernelGGL(car_ego_kernel, dim3(1), dim3(RATO_BLOCK), ego_lds_bytes(p->S), st, *p, us, ego_scratch,
                     (float*)nullptr, (float*)nullptr, 0);
  hipLaunchKernelGGL(car_eval_kernel<false>, dim3(rato::nblocks_for(p->M)), dim3(RATO_BLOCK), 0, st, *p, dW,
                     (uint64_t)0, 0.0f, x0_ped, w_speed, w_rep, ego_scratch, Z, xs, g);
  RATO_LAUNCH_CHECK();
  if (p->stats_workspace)
    return rato_risk_stats(Z, p->M, p->stats_alpha, p->stats_thr, p->stats_workspace, rato_risk_stats_workspace_bytes(p->M),
                           p->stats_out, stream);
  return RATO_OK;
}

// K control sequences against ONE resident batch in one call: us [K][S][2] -> Z [K][ldz] (+ stats_out [K][RATO_N_STATS]).
extern "C" int rato_car_eval_batch(const rato_car_params* p, int32_t K, const float* us, const float* dW, const float* x0_ped,
                                   const float* w_speed, const float* w_rep, float* Z, int64_t ldz, double alpha, float thr,
                                   void* workspace, size_t workspace_bytes, double* stats_out, void* stream) {
  RATO_CLEAR_ERROR();
  if (!params_ok(p) || K < 1 || K > 65535 || !us || !dW || !x0_ped || !w_speed || !w_rep || !Z || ldz < p->M ||
      car_eval_tiles_lds_bytes(p->S) > 48 * 1024)
    return RATO_EINVAL;
  if (stats_out && (!(alpha > 0.0) || !(alpha <= 1.0) || !workspace)) return RATO_EINVAL;
  hipStream_t st = rato::as_stream(stream);
  const int n_tiles = (int)((p->M + RATO_WAVE - 1) / RATO_WAVE);
  const int grid = (n_tiles + CEV_NW - 1) / CEV_NW;
  rato_sel::StatsTail tail = {};
  const int rc = car_eval_tiles_launch(p, K, us, dW, x0_ped, w_speed, w_rep, Z, ldz, nullptr, tail, grid,
                                       car_eval_tiles_lds_bytes(p->S), n_tiles, st);
  if (rc != RATO_OK || !stats_out) return rc;
  return rato_risk_stats_batch(Z, p->M, ldz, K, alpha, thr, workspace, workspace_bytes, stats_out, stream);
}

extern "C" int rato_car_eval_philox(const rato_car_params* p, const float* us, uint64_t seed, float sampler_dt,
                                    const float* x0_ped, const float* w_speed, const float* w_rep,
                                    float* ego_scratch, float* Z, float* xs, float* g, void* stream) {
  RATO_CLEAR_ERROR();
  if (!params_ok(p) || !us || !x0_ped || !w_speed || !w_rep || !ego_scratch || !(sampler_dt >= 0.0f) || p->S > 65535)
    return RATO_EINVAL;
  hipStream_t st = rato::as_stream(stream);
  hipLaunchKernelGGL(car_ego_kernel, dim3(1), dim3(RATO_BLOCK), ego_lds_bytes(p->S), st, *p, us, ego_scratch,
                     (float*)nullptr, (float*)nullptr, 0);
  hipLaunchKernelGGL(car_eval_kernel<true>, dim3(rato::nblocks_for(p->M)), dim3(RATO_BLOCK), 0, st, *p,
                     (const float*)nullptr, seed, sqrtf(sampler_dt), x0_ped, w_speed, w_rep, ego_scratch, Z, xs, g);
  RATO_LAUNCH_CHECK();
  return RATO_OK;
}

extern "C" int rato_car_separation_distances(const rato_car_params* p, const float* xs, float* dist, void* stream) {
  RATO_CLEAR_ERROR();
  if (!params_ok(p) || !xs || !dist) return RATO_EINVAL;
  hipLaunchKernelGGL(car_distance_kernel, dim3(rato::nblocks_for(p->M), p->S), dim3(RATO_BLOCK), 0,
                     rato::as_stream(stream), *p, xs, dist);
  RATO_LAUNCH_CHECK();
  return RATO_OK;
}

namespace {
__device__ unsigned g_car_tile_queues[RATO_QUEUES_TOTAL * 2];   // {next tile, workgroups gone} per queue; every launch leaves
                                                               // its queue zeroed (rato::TileQueuePool hands them out)
unsigned* resolve_car_tile_queues() {
  void* sym = nullptr;
  return hipGetSymbolAddress(&sym, HIP_SYMBOL(g_car_tile_queues)) == hipSuccess ? static_cast<unsigned*>(sym) : nullptr;
}
rato::TileQueuePool g_car_queue_pool;
constexpr size_t CAR_ROWS_LDS_MAX = 160 * 1024;
size_t car_rows_lds_bytes(int S) { return car_rows_lds_floats(S) * sizeof(float); }
}  // namespace

// the launcher's store policy for the row-parallel kernel (see rato_drone_rows_streaming_stores)
extern "C" int rato_car_rows_streaming_stores(int64_t M, int32_t S) {
  static const int nt_env = [] { const char* e = getenv("RATO_NT_STORES"); return e ? atoi(e) : 1; }();
  const double out_bytes = (double)M * (double)rato::pair_row_offset(S) * 2.0 * 4.0;
  const double in_bytes = (double)M * S * 2.0 * 4.0;
  return (nt_env == 2 || (nt_env == 1 && out_bytes >= 256e6 && in_bytes <= 128e6)) ? 1 : 0;
}

extern "C" int rato_car_linearize_plan(int32_t M, int32_t S, int32_t* cols_per_thread, int32_t* tile) {
  if (M <= 0 || S <= 0 || !cols_per_thread || !tile) return RATO_EINVAL;
  int c = *cols_per_thread;
  if (c == 0) c = (S >= 2 && car_rows_lds_bytes(S) <= CAR_ROWS_LDS_MAX) ? -1 : 0;
  if (c == -1 && !(S >= 2 && car_rows_lds_bytes(S) <= CAR_ROWS_LDS_MAX)) return RATO_EINVAL;
  if (c == 0) {
    const long waves_per_group = (long)rato::nblocks_for(M) * (RATO_BLOCK / RATO_WAVE);
    c = 4;
    const int cands[2] = {16, 8};
    for (int i = 0; i < 2; ++i) {
      const int ng = (S + cands[i] - 1) / cands[i];
      if (waves_per_group * ng >= 2048) {
        c = cands[i];
        break;
      }
    }
  }
  if (c != -1 && c != 4 && c != 8 && c != 16) return RATO_EINVAL;
  *cols_per_thread = c;
  *tile = (c == -1) ? CROWS_SAMPLES : RATO_TILE;
  return (c == -1) ? (M + CROWS_SAMPLES - 1) / CROWS_SAMPLES : rato::nblocks_for(M);
}

namespace {
// dW == NULL: the noise is regenerated from (seed, noise_scale) -- the row-parallel kernel only
int car_linearize_impl(const rato_car_params* p, const float* us, const float* dW, uint64_t seed, float noise_scale,
                       const float* x0_ped, const float* w_speed, const float* w_rep, float* ego_scratch, float* G,
                       float* g_up, float* Z, float* final_du, float* final_rhs, int32_t cols_per_thread,
                       void* stream, int noise_tiled = 0) {
  RATO_CLEAR_ERROR();
  if (!params_ok(p) || !us || !x0_ped || !w_speed || !w_rep || !ego_scratch || !G || !g_up)
    return RATO_EINVAL;
  hipStream_t st = rato::as_stream(stream);
  int32_t spt = cols_per_thread, tile = 0;
  if (rato_car_linearize_plan(p->M, p->S, &spt, &tile) < 0) return RATO_EINVAL;
  if (!dW && spt != -1) return RATO_EINVAL;
  if (noise_tiled && (spt != -1 || !dW)) return RATO_EINVAL;   // the tiled noise layout: row-parallel kernel only
  if (p->stats_workspace && (spt != -1 || !Z || !p->stats_out || !(p->stats_alpha > 0.0) || !(p->stats_alpha <= 1.0)))
    return RATO_EINVAL;   // statistics in the same launch: row-parallel kernel, Z requested
  // the ego prologue (trajectory + per-step tangents Epos, Eu) serves the forward/column kernel; the row-parallel
  // kernel builds its ego tables itself, under the latency of its noise loads
  if (spt != -1)
    hipLaunchKernelGGL(car_ego_kernel, dim3(1), dim3(RATO_BLOCK), ego_lds_bytes(p->S), st, *p, us, ego_scratch,
                       final_du, final_rhs, 1);
  if (spt == -1) {
    const size_t lds = car_rows_lds_bytes(p->S);
    static rato::DynamicLdsLimit lds_limit;   // per device; cached: capture-safe after the first call
    {
      const hipError_t e = lds_limit.ensure(lds, [](size_t bytes) {
        hipError_t err = hipSuccess;
        const void* kernels[4] = {reinterpret_cast<const void*>(car_linearize_rows_kernel<false, false>),
                                  reinterpret_cast<const void*>(car_linearize_rows_kernel<true, false>),
                                  reinterpret_cast<const void*>(car_linearize_rows_kernel<false, true>),
                                  reinterpret_cast<const void*>(car_linearize_rows_kernel<true, true>)};
        for (int i = 0; i < 4 && err == hipSuccess; ++i)
          err = hipFuncSetAttribute(kernels[i], hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        return err;
      });
      if (e != hipSuccess) return RATO_EHIP - (int)e;
    }
    const int n_tiles = (p->M + CROWS_SAMPLES - 1) / CROWS_SAMPLES;
    // large batches: one workgroup per slot + a global tile queue (XCD load balance, see the kernel)
    const int cus = g_car_queue_pool.cus();
    int per_cu = (int)(CAR_ROWS_LDS_MAX / lds);
    if (per_cu > 32 / CROWS_NW) per_cu = 32 / CROWS_NW;
    if (per_cu < 1) per_cu = 1;
    static const int slots_env = [] { const char* e = getenv("RATO_CAR_SLOTS_PER_CU"); return e ? atoi(e) : 0; }();   // A/B knob
    if (slots_env >= 1 && slots_env < per_cu) per_cu = slots_env;
    const int slots = cus * per_cu;
    static const int dynamic_env = [] { const char* e = getenv("RATO_ROWS_DYNAMIC"); return e ? atoi(e) : 1; }();
    unsigned* queue = nullptr;
    int grid_x = n_tiles;
    if (dynamic_env && n_tiles > slots)
      queue = g_car_queue_pool.take(st, resolve_car_tile_queues);   // per stream / per captured launch; none left: static form
    // Two queue workgroups per CU, not the three the LDS allows: same box, alternating (tools/ab_car_slots.sh), 3 -> 2:
    // C5 shard (M = 125,000) 0.1923-0.1938 -> 0.1846-0.1877 ms (noise read), 0.1768-0.1771 -> 0.1728-0.1734 (regenerated);
    // M = 1e6 1.162-1.175 -> 1.158-1.163 / 1.105-1.108 -> 1.102-1.111; one per CU: +17 %.  RATO_CAR_SLOTS_PER_CU overrides.
    const int qslots = (slots_env < 1 && per_cu > 2) ? cus * 2 : slots;
    if (queue) {
      grid_x = qslots;
    }
    // small batches (fewer tiles than workgroup slots): every tile split over several workgroups (>= 4 row tasks each)
    int split = 1;
    static const int small_split = [] { const char* e = getenv("RATO_CAR_SMALL_SPLIT"); return e ? atoi(e) : -1; }();
    if (!queue && n_tiles < slots) {
      // C3 (M = 1e4: 157 tiles on 768 slots), same box, alternating, kern_ms: split 1 / 2 / 3 / 4 = 0.0302-0.0307 /
      // 0.0279-0.0282 / 0.0270-0.0271 (one run 0.0411) / 0.0360-0.0361: every part rebuilds the fp64 ego tables and
      // re-stages the noise tile, so two parts per tile is where it stops paying reliably.
      split = small_split >= 1 ? small_split : (slots / n_tiles >= 2 ? 2 : 1);
      const int max_split = (p->S + 3) / 4 < 1 ? 1 : (p->S + 3) / 4;
      if (split > max_split) split = max_split;
      if (split < 1) split = 1;
      grid_x = n_tiles * split;
    }
    // statistics of Z in extra workgroups of this launch (params.stats_*)
    rato_sel::StatsTail tail = {};
    size_t lds_launch = lds;
    int grid_launch = grid_x;
    // (in the launch only while the whole grid is resident at once -- no queue; otherwise behind it, below)
    const bool stats_behind = p->stats_workspace && queue;
    if (p->stats_workspace && !stats_behind) {
      int Gs = 0;
      const int extra = rato_sel::stats_tail_workgroups<CROWS_NW * RATO_WAVE>(p->M, Gs);
      if (extra < 0) return RATO_EINVAL;   // beyond the one-launch forms of the selection: use rato_risk_stats
      tail.ws = static_cast<rato_sel::Workspace*>(p->stats_workspace);
      tail.out = p->stats_out;
      tail.alpha = p->stats_alpha;
      tail.thr = p->stats_thr;
      tail.G = Gs;
      tail.n_prod = grid_x;
      rato_sel::stats_rank(p->M, p->stats_alpha, tail.k, tail.var_is_max);
      grid_launch = grid_x + extra;
      if (lds_launch < rato_sel::rs_body_lds_bytes<CROWS_NW * RATO_WAVE>()) lds_launch = rato_sel::rs_body_lds_bytes<CROWS_NW * RATO_WAVE>();
    }
    dim3 grid(grid_launch), block(CROWS_NW * RATO_WAVE);
    // streaming stores for the Jacobian when the output is far beyond the memory-side cache and the inputs fit (drone.hip):
    // C5 shard (800 MB out, 40 MB in) -11.7 %; M = 1e6 (6.4 GB out, 320 MB in) ordinary stores (+2 % with streaming ones)
    const bool nt_stores = rato_car_rows_streaming_stores(p->M, p->S) != 0;
    if (queue) {
      // the last `tail_tiles` tiles of the queue as `tail_split` parts each (RATO_CAR_TAIL_SPLIT / RATO_CAR_TAIL_TILES).
      // OFF by default: unlike the drone's products output it does not pay here -- C5 shard (M = 125,000, 1954 tiles
      // on 768 slots), same box, alternating (tools/ab_car_tail.sh), kernel ms: whole tiles 0.1813-0.1816 | halves over
      // the last 384 / 768 tiles 0.1817-0.1823 / 0.1844-0.1856 | thirds 0.1904-0.1907 | quarters 0.1994-0.2012.
      static const int tail_split_env = [] { const char* e = getenv("RATO_CAR_TAIL_SPLIT"); return e ? atoi(e) : 1; }();
      static const int tail_tiles_env = [] { const char* e = getenv("RATO_CAR_TAIL_TILES"); return e ? atoi(e) : -1; }();
      int tail_split = tail_split_env < 1 ? 1 : tail_split_env;
      const int max_split = (p->S + 3) / 4 < 1 ? 1 : (p->S + 3) / 4;
      if (tail_split > max_split) tail_split = max_split;
      int tail_tiles = tail_tiles_env >= 0 ? tail_tiles_env : qslots / 2;
      if (tail_tiles > n_tiles) tail_tiles = n_tiles;
      const int n_whole = tail_split > 1 ? n_tiles - tail_tiles : n_tiles;
      if (dW)
        hipLaunchKernelGGL((car_linearize_rows_kernel<true, false>), grid, block, lds_launch, st, *p, seed, noise_scale, us,
                           dW, x0_ped, w_speed, w_rep, final_du, final_rhs, G, g_up, Z, n_tiles, queue, tail_split, n_whole, tail, (noise_tiled ? 1 : 0) | (nt_stores ? 2 : 0));
      else
        hipLaunchKernelGGL((car_linearize_rows_kernel<true, true>), grid, block, lds_launch, st, *p, seed, noise_scale, us,
                           dW, x0_ped, w_speed, w_rep, final_du, final_rhs, G, g_up, Z, n_tiles, queue, tail_split, n_whole, tail, (noise_tiled ? 1 : 0) | (nt_stores ? 2 : 0));
    } else {
      if (dW)
        hipLaunchKernelGGL((car_linearize_rows_kernel<false, false>), grid, block, lds_launch, st, *p, seed, noise_scale, us,
                           dW, x0_ped, w_speed, w_rep, final_du, final_rhs, G, g_up, Z, n_tiles, queue, split, 0, tail, (noise_tiled ? 1 : 0) | (nt_stores ? 2 : 0));
      else
        hipLaunchKernelGGL((car_linearize_rows_kernel<false, true>), grid, block, lds_launch, st, *p, seed, noise_scale, us,
                           dW, x0_ped, w_speed, w_rep, final_du, final_rhs, G, g_up, Z, n_tiles, queue, split, 0, tail, (noise_tiled ? 1 : 0) | (nt_stores ? 2 : 0));
    }
    RATO_LAUNCH_CHECK();
    if (stats_behind)
      return rato_risk_stats(Z, p->M, p->stats_alpha, p->stats_thr, p->stats_workspace,
                             rato_risk_stats_workspace_bytes(p->M), p->stats_out, stream);
    return RATO_OK;
  }
  switch (spt) {
    case 4: return launch_car_linearize<4>(p, dW, x0_ped, w_speed, w_rep, ego_scratch, G, g_up, Z, st);
    case 8: return launch_car_linearize<8>(p, dW, x0_ped, w_speed, w_rep, ego_scratch, G, g_up, Z, st);
    case 16: return launch_car_linearize<16>(p, dW, x0_ped, w_speed, w_rep, ego_scratch, G, g_up, Z, st);
    default: return RATO_EINVAL;
  }
}
}  // namespace

extern "C" int rato_car_linearize(const rato_car_params* p, const float* us, const float* dW,
                                  const float* x0_ped, const float* w_speed, const float* w_rep,
                                  float* ego_scratch, float* G, float* g_up, float* Z, float* final_du,
                                  float* final_rhs, int32_t cols_per_thread, void* stream) {
  if (!dW) return RATO_EINVAL;
  return car_linearize_impl(p, us, dW, 0, 0.0f, x0_ped, w_speed, w_rep, ego_scratch, G, g_up, Z, final_du, final_rhs,
                            cols_per_thread, stream);
}

extern "C" int rato_car_linearize_philox(const rato_car_params* p, const float* us, uint64_t seed, float sampler_dt,
                                         const float* x0_ped, const float* w_speed, const float* w_rep,
                                         float* ego_scratch, float* G, float* g_up, float* Z, float* final_du,
                                         float* final_rhs, void* stream) {
  if (!(sampler_dt > 0.0f)) return RATO_EINVAL;
  return car_linearize_impl(p, us, nullptr, seed, sqrtf(sampler_dt), x0_ped, w_speed, w_rep, ego_scratch, G, g_up, Z,
                            final_du, final_rhs, -1, stream);
}

// Would rato_car_linearize (row-parallel kernel) with params.stats_* compute the statistics IN its launch?  (see
// rato_drone_stats_in_launch)  1 yes, 0 no.
namespace {
__global__ __launch_bounds__(RATO_BLOCK) void car_tile_noise_kernel(const float* __restrict__ dW, long M, int nrows,
                                                                   float* __restrict__ out) {
  const long i = (long)blockIdx.x * RATO_BLOCK + threadIdx.x;        // over [tile][row][lane]
  const long n_tiles = (M + CROWS_SAMPLES - 1) / CROWS_SAMPLES;
  if (i >= n_tiles * nrows * CROWS_SAMPLES) return;
  // a tile's block is the image the row kernel keeps in LDS: [S][64] (xi_0, xi_1) pairs
  const long tile = i / ((long)nrows * CROWS_SAMPLES);
  const int o = (int)(i - tile * (long)nrows * CROWS_SAMPLES);
  const int t = o / (2 * CROWS_SAMPLES), lane = (o % (2 * CROWS_SAMPLES)) / 2, a = o & 1;
  const long m = tile * CROWS_SAMPLES + lane;
  out[i] = (m < M) ? dW[(size_t)(t * 2 + a) * M + m] : 0.0f;
}
}  // namespace

extern "C" size_t rato_car_tiled_noise_floats(int64_t M, int32_t S) {
  return (M > 0 && S > 0) ? (size_t)((M + CROWS_SAMPLES - 1) / CROWS_SAMPLES) * (size_t)(2 * S) * CROWS_SAMPLES : 0;
}

extern "C" int rato_car_tile_noise(const float* dW, int64_t M, int32_t S, float* dW_tiled, void* stream) {
  RATO_CLEAR_ERROR();
  if (!dW || !dW_tiled || M <= 0 || S <= 0) return RATO_EINVAL;
  const size_t n = rato_car_tiled_noise_floats(M, S);
  hipLaunchKernelGGL(car_tile_noise_kernel, dim3((unsigned)((n + RATO_BLOCK - 1) / RATO_BLOCK)), dim3(RATO_BLOCK), 0,
                     rato::as_stream(stream), dW, (long)M, 2 * S, dW_tiled);
  RATO_LAUNCH_CHECK();
  return RATO_OK;
}

extern "C" int rato_car_linearize_tiled(const rato_car_params* p, const float* us, const float* dW_tiled,
                                        const float* x0_ped, const float* w_speed, const float* w_rep, float* ego_scratch,
                                        float* G, float* g_up, float* Z, float* final_du, float* final_rhs, void* stream) {
  return car_linearize_impl(p, us, dW_tiled, 0, 0.0f, x0_ped, w_speed, w_rep, ego_scratch, G, g_up, Z, final_du, final_rhs,
                            -1, stream, 1);
}

extern "C" int rato_car_stats_in_launch(int32_t M, int32_t S) {
  if (M <= 0 || S < 2 || car_rows_lds_bytes(S) > CAR_ROWS_LDS_MAX) return 0;
  int per_cu = (int)(CAR_ROWS_LDS_MAX / car_rows_lds_bytes(S));
  if (per_cu > 32 / CROWS_NW) per_cu = 32 / CROWS_NW;
  if (per_cu < 1) per_cu = 1;
  const int n_tiles = (M + CROWS_SAMPLES - 1) / CROWS_SAMPLES;
  int G = 0;
  return n_tiles <= g_car_queue_pool.cus() * per_cu && rato_sel::stats_tail_workgroups<CROWS_NW * RATO_WAVE>(M, G) > 0;
}
