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
#include "rato_common.h"

namespace {

constexpr int NOBS = RATO_DRONE_NOBS;

struct SampleConsts {
  float inv_m, a21, cn, dtm;
  float q00[NOBS], qs[NOBS], q11[NOBS];
};

__device__ __forceinline__ SampleConsts load_consts(const rato_drone_params& P, const float* __restrict__ mass,
                                                    const float* __restrict__ Qsym, size_t M, size_t m) {
  SampleConsts c;
  c.inv_m = 1.0f / mass[m];
  c.a21 = -P.kp * P.dt * c.inv_m;
  c.cn = sqrtf(P.dt) * P.beta * c.inv_m;  // sqrt(dt) * (beta/m): drone_risk.py:136,151
  c.dtm = P.dt * c.inv_m;
#pragma unroll
  for (int j = 0; j < NOBS; ++j) {
    c.q00[j] = Qsym[(size_t)(j * 3 + 0) * M + m];
    c.qs[j] = Qsym[(size_t)(j * 3 + 1) * M + m];
    c.q11[j] = Qsym[(size_t)(j * 3 + 2) * M + m];
  }
  return c;
}

// One Euler–Maruyama step of one axis (drone_risk.py:122-131,148-153).
__device__ __forceinline__ void step_axis(const rato_drone_params& P, const SampleConsts& c, float u, float xi,
                                          float& p, float& v) {
  const float acc = (u - (P.kp * p + P.kd * v)) * c.inv_m - P.drag * fabsf(v) * v * c.inv_m;
  const float pn = p + P.dt * v;
  const float vn = v + P.dt * acc + c.cn * xi;
  p = pn;
  v = vn;
}

__global__ __launch_bounds__(RATO_BLOCK) void drone_eval_kernel(
    rato_drone_params P, const float* __restrict__ us, const float* __restrict__ dW,
    const float* __restrict__ mass, const float* __restrict__ Qsym, float* __restrict__ Z,
    float* __restrict__ xs, float* __restrict__ g) {
  const size_t M = (size_t)P.M;
  const size_t m = (size_t)blockIdx.x * RATO_BLOCK + threadIdx.x;
  if (m >= M) return;
  const int S = P.S;
  const SampleConsts c = load_consts(P, mass, Qsym, M, m);
  float p[3], v[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    p[a] = P.x_init[a];
    v[a] = P.x_init[3 + a];
  }
  if (xs) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      xs[(size_t)a * M + m] = p[a];
      xs[(size_t)(3 + a) * M + m] = v[a];
    }
  }
  float zmax = -INFINITY;
  float xi[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) xi[a] = dW[(size_t)a * M + m];
  for (int t = 0; t < S; ++t) {
    float nxt[3];
    const int tn = (t + 1 < S) ? t + 1 : t;  // prefetch next step's noise
#pragma unroll
    for (int a = 0; a < 3; ++a) nxt[a] = dW[(size_t)(tn * 3 + a) * M + m];
#pragma unroll
    for (int a = 0; a < 3; ++a) step_axis(P, c, us[t * 3 + a], xi[a], p[a], v[a]);
    if (xs) {
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        xs[((size_t)(t + 1) * 6 + a) * M + m] = p[a];
        xs[((size_t)(t + 1) * 6 + 3 + a) * M + m] = v[a];
      }
    }
#pragma unroll
    for (int j = 0; j < NOBS; ++j) {
      const float dx = p[0] - P.obs_xy[j][0], dy = p[1] - P.obs_xy[j][1];
      const float gj = 1.0f - (c.q00[j] * dx * dx + c.qs[j] * dx * dy + c.q11[j] * dy * dy);
      zmax = fmaxf(zmax, gj);
      if (g) g[((size_t)j * S + t) * M + m] = gj;
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) xi[a] = nxt[a];
  }
  if (Z) Z[m] = zmax - P.tol;
}

__global__ __launch_bounds__(RATO_BLOCK) void drone_obstacle_kernel(rato_drone_params P,
                                                                    const float* __restrict__ xs,
                                                                    const float* __restrict__ Qsym,
                                                                    float* __restrict__ g) {
  const size_t M = (size_t)P.M;
  const size_t m = (size_t)blockIdx.x * RATO_BLOCK + threadIdx.x;
  const int t = blockIdx.y;
  if (m >= M) return;
  const float px = xs[((size_t)(t + 1) * 6 + 0) * M + m];
  const float py = xs[((size_t)(t + 1) * 6 + 1) * M + m];
#pragma unroll
  for (int j = 0; j < NOBS; ++j) {
    const float q00 = Qsym[(size_t)(j * 3 + 0) * M + m], qs = Qsym[(size_t)(j * 3 + 1) * M + m],
                q11 = Qsym[(size_t)(j * 3 + 2) * M + m];
    const float dx = px - P.obs_xy[j][0], dy = py - P.obs_xy[j][1];
    g[((size_t)j * P.S + t) * M + m] = 1.0f - (q00 * dx * dx + qs * dx * dy + q11 * dy * dy);
  }
}

// Column owned by slot k of column-group grp (serpentine, so that the
// triangular work S-1-s is balanced over groups).  >= S means "no column".
__device__ __forceinline__ int column_of(int k, int grp, int ngroups) {
  return k * ngroups + ((k & 1) ? (ngroups - 1 - grp) : grp);
}

template <int CPT>
__global__ __launch_bounds__(RATO_BLOCK) void drone_linearize_kernel(
    rato_drone_params P, const float* __restrict__ us, const float* __restrict__ dW,
    const float* __restrict__ mass, const float* __restrict__ Qsym, float* __restrict__ G,
    float* __restrict__ g_up, float* __restrict__ Z, float* __restrict__ part_du,
    float* __restrict__ part_rhs) {
  const size_t M = (size_t)P.M;
  const size_t m_raw = (size_t)blockIdx.x * RATO_BLOCK + threadIdx.x;
  const bool valid = m_raw < M;
  const size_t m = valid ? m_raw : M - 1;  // clamp loads; stores are predicated
  const int S = P.S;
  const int grp = blockIdx.y, ngroups = gridDim.y;
  const bool lead = (grp == 0);  // group 0 also emits g_up, Z and the rhs partial
  const SampleConsts c = load_consts(P, mass, Qsym, M, m);

  int col[CPT];
#pragma unroll
  for (int k = 0; k < CPT; ++k) {
    const int s = column_of(k, grp, ngroups);
    col[k] = (s < S) ? s : 0x7fffffff;
  }
  // Phi[k][a] = (P, V) = d(p_a, v_a)_t / d u_{col[k], a}
  float phiP[CPT][3], phiV[CPT][3];
#pragma unroll
  for (int k = 0; k < CPT; ++k)
#pragma unroll
    for (int a = 0; a < 3; ++a) phiP[k][a] = phiV[k][a] = 0.0f;

  float p[3], v[3], dp[3], dv[3];  // state and its tangent in the direction u (for g_up / rhs)
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    p[a] = P.x_init[a];
    v[a] = P.x_init[3 + a];
    dp[a] = dv[a] = 0.0f;
  }
  float zmax = -INFINITY;
  float xi[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) xi[a] = dW[(size_t)a * M + m];

  for (int t = 0; t < S; ++t) {
    float nxt[3];
    const int tn = (t + 1 < S) ? t + 1 : t;
#pragma unroll
    for (int a = 0; a < 3; ++a) nxt[a] = dW[(size_t)(tn * 3 + a) * M + m];

    float a22[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) a22[a] = 1.0f - P.dt * (P.kd + 2.0f * P.drag * fabsf(v[a])) * c.inv_m;

    // tangent along u: d x_{t+1} = A_t d x_t + B u_t
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const float u = us[t * 3 + a];
      const float dpn = dp[a] + P.dt * dv[a];
      const float dvn = c.a21 * dp[a] + a22[a] * dv[a] + c.dtm * u;
      dp[a] = dpn;
      dv[a] = dvn;
    }
    // column sensitivities
#pragma unroll
    for (int k = 0; k < CPT; ++k) {
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        const float Pn = phiP[k][a] + P.dt * phiV[k][a];
        const float Vn = c.a21 * phiP[k][a] + a22[a] * phiV[k][a];
        phiP[k][a] = Pn;
        phiV[k][a] = (col[k] == t) ? c.dtm : Vn;  // Phi_{t+1,t} = B_t
      }
    }
    // state
#pragma unroll
    for (int a = 0; a < 3; ++a) step_axis(P, c, us[t * 3 + a], xi[a], p[a], v[a]);

    // constraint row t (uses p_{t+1}) and its gradient w = -(Q+Q^T) d
    float wx[NOBS], wy[NOBS];
#pragma unroll
    for (int j = 0; j < NOBS; ++j) {
      const float dx = p[0] - P.obs_xy[j][0], dy = p[1] - P.obs_xy[j][1];
      const float gj = 1.0f - (c.q00[j] * dx * dx + c.qs[j] * dx * dy + c.q11[j] * dy * dy);
      wx[j] = -(2.0f * c.q00[j] * dx + c.qs[j] * dy);
      wy[j] = -(c.qs[j] * dx + 2.0f * c.q11[j] * dy);
      zmax = fmaxf(zmax, gj);
      if (lead && valid) g_up[((size_t)j * S + t) * M + m] = -gj + wx[j] * dp[0] + wy[j] * dp[1];
    }
    const size_t row = (size_t)rato::pair_row_offset(t);
#pragma unroll
    for (int k = 0; k < CPT; ++k) {
      if (col[k] < t) {  // wave-uniform
        const size_t base = ((row + (size_t)col[k]) * 2) * NOBS;
        if (valid) {
#pragma unroll
          for (int j = 0; j < NOBS; ++j) {
            G[(base + j) * M + m] = wx[j] * phiP[k][0];
            G[(base + NOBS + j) * M + m] = wy[j] * phiP[k][1];
          }
        }
      }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) xi[a] = nxt[a];
  }

  if (lead && valid && Z) Z[m] = zmax - P.tol;

  // per-block sums for the sample mean of the final-constraint linearization
  __shared__ float red[RATO_BLOCK / RATO_WAVE][CPT * 6 + 6];
  const int lane = threadIdx.x & (RATO_WAVE - 1), wave = threadIdx.x / RATO_WAVE;
#pragma unroll
  for (int k = 0; k < CPT; ++k) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const float sp = rato::wave_sum(valid ? phiP[k][a] : 0.0f);
      const float sv = rato::wave_sum(valid ? phiV[k][a] : 0.0f);
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
      const float rp = rato::wave_sum(valid ? (-(p[a] - P.x_final[a]) + dp[a]) : 0.0f);
      const float rv = rato::wave_sum(valid ? (-(v[a] - P.x_final[3 + a]) + dv[a]) : 0.0f);
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
      part_du[((size_t)blockIdx.x * S + s) * 6 + e] = acc;
    }
  } else if (lead && tid < CPT * 6 + 6) {
    float acc = 0.0f;
#pragma unroll
    for (int w = 0; w < RATO_BLOCK / RATO_WAVE; ++w) acc += red[w][tid];
    part_rhs[(size_t)blockIdx.x * 6 + (tid - CPT * 6)] = acc;
  }
}

bool params_ok(const rato_drone_params* p) {
  return p && p->M > 0 && p->S > 0 && p->S <= 4096 && p->dt > 0.0f;
}

}  // namespace

extern "C" int rato_drone_eval(const rato_drone_params* p, const float* us, const float* dW, const float* mass,
                               const float* Qsym, float* Z, float* xs, float* g, void* stream) {
  RATO_CLEAR_ERROR();
  if (!params_ok(p) || !us || !dW || !mass || !Qsym) return RATO_EINVAL;
  dim3 grid(rato::nblocks_for(p->M)), block(RATO_BLOCK);
  hipLaunchKernelGGL(drone_eval_kernel, grid, block, 0, rato::as_stream(stream), *p, us, dW, mass, Qsym, Z, xs, g);
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

extern "C" int rato_drone_linearize_nblocks(int32_t M) { return M > 0 ? rato::nblocks_for(M) : RATO_EINVAL; }

namespace {
template <int CPT>
int launch_linearize(const rato_drone_params* p, const float* us, const float* dW, const float* mass,
                     const float* Qsym, float* G, float* g_up, float* Z, float* part_du, float* part_rhs,
                     hipStream_t stream) {
  const int ngroups = (p->S + CPT - 1) / CPT;
  dim3 grid(rato::nblocks_for(p->M), ngroups), block(RATO_BLOCK);
  hipLaunchKernelGGL(drone_linearize_kernel<CPT>, grid, block, 0, stream, *p, us, dW, mass, Qsym, G, g_up, Z,
                     part_du, part_rhs);
  RATO_LAUNCH_CHECK();
  return RATO_OK;
}
}  // namespace

extern "C" int rato_drone_linearize(const rato_drone_params* p, const float* us, const float* dW,
                                    const float* mass, const float* Qsym, float* G, float* g_up, float* Z,
                                    float* part_du, float* part_rhs, int32_t cols_per_thread, void* stream) {
  RATO_CLEAR_ERROR();
  if (!params_ok(p) || !us || !dW || !mass || !Qsym || !G || !g_up || !part_du || !part_rhs) return RATO_EINVAL;
  int cpt = cols_per_thread;
  if (cpt == 0) {
    // largest column group that still yields >= 2 waves per SIMD (1024 SIMDs)
    const long waves_per_group = (long)rato::nblocks_for(p->M) * (RATO_BLOCK / RATO_WAVE);
    cpt = 4;
    const int cands[3] = {32, 16, 8};
    for (int i = 0; i < 3; ++i) {
      const int ng = (p->S + cands[i] - 1) / cands[i];
      if (waves_per_group * ng >= 2048) {
        cpt = cands[i];
        break;
      }
    }
  }
  hipStream_t st = rato::as_stream(stream);
  switch (cpt) {
    case 4: return launch_linearize<4>(p, us, dW, mass, Qsym, G, g_up, Z, part_du, part_rhs, st);
    case 8: return launch_linearize<8>(p, us, dW, mass, Qsym, G, g_up, Z, part_du, part_rhs, st);
    case 16: return launch_linearize<16>(p, us, dW, mass, Qsym, G, g_up, Z, part_du, part_rhs, st);
    case 32: return launch_linearize<32>(p, us, dW, mass, Qsym, G, g_up, Z, part_du, part_rhs, st);
    default: return RATO_EINVAL;
  }
}
