// Device oracle of the linearized CVaR constraint (gfx950) — what a cutting-plane solve of the SCP
// subproblem needs per cut, computed where the linearization already lives (HBM):
//
//   m_i(x) = max_r [ (G_i x)_r + sign * base_{i,r} ]     rato_saa_rowmax / rato_drone_rowmax_implicit
//   sum_i w_i G_i[r*_i, :],  sum_i w_i base_{i,r*_i}     rato_saa_tail_rows_batch / rato_drone_tail_rows_implicit
//
// Eliminating the auxiliary y_i of the reference's QP (drone_risk.py:327-368: y_i >= -slack,
// y_i >= (G_i u - g_up_i)_r - t) leaves  alpha M CVaR_alpha(m(u)) - (M(1-alpha) - 1) slack <= 0  in the
// variables (u, slack) only.  The rows come in two algebraically identical forms:
//   reference form  x = u,         base = g_up = -g + G u_k,  sign = -1     (drone_risk.py:278, :357-364)
//   delta form      x = u - u_k,   base = g (the constraint value at u_k), sign = +1
// The data are the fp32 outputs of the linearize kernels; ALL arithmetic here is fp64 (the kernels are HBM-bound
// on the fp32 reads, the fp64 FMAs are free): a cut's value and its gradient are then consistent to 1e-13, which
// is what lets the cutting-plane loop reproduce the optimum of the full QP to the 1e-5 the north star asks for
// (fp32 rows of magnitude 1e2 carried 1e-5 of noise per cut).  The delta form additionally keeps the fp32 rounding
// of g_up (|g_up| ~ 1e2 against |g| ~ 1e-2 on the rows that matter) out of the rows.
#include "rato_common.h"
#include <atomic>
#include <chrono>
#include <cstdlib>
#include <type_traits>

namespace {

constexpr int RM_NW = 8;        // waves per workgroup (64 samples)

typedef double rdouble2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float tail_weight(float mv, float tstar, float lambda) {
  return (mv > tstar) ? 1.0f : ((mv == tstar) ? lambda : 0.0f);
}

// threshold t (slot 10: the Rockafellar-Uryasev minimiser; slot 0 = VaR wraps to max(Z) when floor(alpha M) = M)
// and the weight of the samples that tie with it, from a rato_risk_stats record
__device__ __forceinline__ void tail_rule(const double* __restrict__ st, double alphaM, float& tstar, float& lambda) {
  tstar = (float)st[10];
  const double n_gt = st[8], n_eq = st[9];
  const double l = (n_eq > 0.0) ? (alphaM - n_gt) / n_eq : 0.0;
  lambda = (float)fmin(fmax(l, 0.0), 1.0);
}

// The step-Jacobian table comes in two forms, told apart by its axis count: [S][2][ld] from the row-parallel linearize
// kernel holds a22 itself (its fp32 LDS table); [S][3][ld] from the generators-only linearization holds the complement
// e22 = 1 - a22 ~ 1e-3, which an fp32 number represents to 1e-10 of a22 (drone.hip).  Either way a22 in fp64:
__device__ __forceinline__ double a22_value(float stored, bool complement) {
  return complement ? 1.0 - (double)stored : (double)stored;
}

// Block-wide, order-preserving compaction of the tail samples.  Only ~alpha of the samples carry a weight; left where
// they are, every wave of the block would run the whole sweep (and its fp64 wave sums) for a handful of active lanes.
// The samples with w != 0 are moved, in sample order (deterministic sums), to lanes [0, n) of the block: afterwards
// lane L < n holds (sample index, weight, row) of the L-th tail sample and only ceil(n / 64) waves have work.
struct TailLane {
  long m;     // sample index (clamped to a valid one for idle lanes)
  float w;    // tail weight, 0 for idle lanes
  int t, r;   // arg-max row (step, row group)
};
struct TailLists {   // the compacted lists themselves (LDS), for kernels that walk them in chunks of one wave
  int* src;
  float* w;
  int* tr;
};
__device__ __forceinline__ int compact_tail(float w, int t, int r, long block_base, long M, TailLane& out,
                                            TailLists* lists = nullptr) {
  __shared__ int s_cnt[RATO_BLOCK / RATO_WAVE];
  __shared__ int s_src[RATO_BLOCK];
  __shared__ float s_w[RATO_BLOCK];
  __shared__ int s_tr[RATO_BLOCK];
  if (lists) {
    lists->src = s_src;
    lists->w = s_w;
    lists->tr = s_tr;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool act = (w != 0.0f);
  const unsigned long long bal = __ballot(act);
  const int pre = __popcll(bal & ((1ull << lane) - 1ull));
  if (lane == 0) s_cnt[wave] = __popcll(bal);
  __syncthreads();
  int base = 0, total = 0;
#pragma unroll
  for (int i = 0; i < RATO_BLOCK / RATO_WAVE; ++i) {
    const int c = s_cnt[i];
    if (i < wave) base += c;
    total += c;
  }
  if (act) {
    s_src[base + pre] = threadIdx.x;
    s_w[base + pre] = w;
    s_tr[base + pre] = t | (r << 20);
  }
  __syncthreads();
  const bool work = (int)threadIdx.x < total;
  const int src = work ? s_src[threadIdx.x] : 0;
  const long m = block_base + src;
  out.m = (m < M) ? m : M - 1;
  out.w = work ? s_w[threadIdx.x] : 0.0f;
  const int tr = work ? s_tr[threadIdx.x] : 0;
  out.t = tr & 0xfffff;
  out.r = tr >> 20;
  return total;
}

// grid = ceil(M/64) workgroups; rows t are pulled from an LDS queue, longest first.
template <int R, bool FACT>
__global__ __launch_bounds__(RM_NW* RATO_WAVE) void rowmax_kernel(const float* __restrict__ G,
                                                                   const float* __restrict__ W, int tileW, int S,
                                                                   long M, long ld, const float* __restrict__ base,
                                                                   double sign, const double* __restrict__ xs, int n_u,
                                                                   float* __restrict__ m_out,
                                                                   int* __restrict__ arg_out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char rm_lds[];
  rdouble2_t* US = reinterpret_cast<rdouble2_t*>(rm_lds);            // [S] (x_{s,0}, x_{s,1})
  double* BV = reinterpret_cast<double*>(US + S);                    // [RM_NW][64] best value per wave
  int* BI = reinterpret_cast<int*>(BV + RM_NW * 64);                 // [RM_NW][64] best row index
  int* head = BI + RM_NW * 64;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < S; i += RM_NW * RATO_WAVE) {
    rdouble2_t u2;
    u2.x = xs[i * n_u + 0];
    u2.y = xs[i * n_u + 1];
    US[i] = u2;
  }
  if (threadIdx.x == 0) *head = 0;
  __syncthreads();
  const long i0 = (long)blockIdx.x * 64;
  const long m_raw = i0 + lane;
  const bool valid = m_raw < M;
  const long m = valid ? m_raw : M - 1;
  const size_t n_pairs = (size_t)S * (S - 1) / 2;
  constexpr int RR = FACT ? 1 : R;   // row groups stored per (pair, control): factored keeps only Phi
  const float* __restrict__ Gt = G + (size_t)(i0 / tileW) * rato::packed_tile_stride(n_pairs * 2 * RR * tileW) + (i0 % tileW) + lane;
  double best = -INFINITY;
  int best_idx = 0;
  auto next_task = [&]() -> int {
    int v = 0;
    if (lane == 0) v = atomicAdd(head, 1);
    return __builtin_amdgcn_readfirstlane(v);
  };
  int task = next_task();
  while (task < S) {
    const int t = S - 1 - task;
    double acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = 0.0;
    const float* __restrict__ row = Gt + (size_t)rato::pair_row_offset(t) * (2 * RR * tileW);
    if (FACT) {
      double ax = 0.0, ay = 0.0;   // (Phi x) per axis, shared by the row groups
#pragma unroll 8
      for (int s2 = 0; s2 < t; ++s2) {
        const rdouble2_t u2 = US[s2];
        const float* __restrict__ o = row + (size_t)s2 * (2 * tileW);
        ax += (double)(valid ? o[0] : 0.0f) * u2.x;
        ay += (double)(valid ? o[tileW] : 0.0f) * u2.y;
      }
#pragma unroll
      for (int r = 0; r < R; ++r)
        acc[r] = (double)W[(((size_t)r * S + t) * 2 + 0) * ld + m] * ax + (double)W[(((size_t)r * S + t) * 2 + 1) * ld + m] * ay;
    } else {
#pragma unroll 4
      for (int s2 = 0; s2 < t; ++s2) {
        const rdouble2_t u2 = US[s2];
        const float* __restrict__ o = row + (size_t)s2 * (2 * R * tileW);
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const float g0 = valid ? o[r * tileW] : 0.0f;
          const float g1 = valid ? o[(R + r) * tileW] : 0.0f;
          acc[r] += (double)g0 * u2.x + (double)g1 * u2.y;
        }
      }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const double v = acc[r] + sign * (double)base[((size_t)r * S + t) * ld + m];
      if (v > best) {           // strict: ties keep the row found first (deterministic given the row)
        best = v;
        best_idx = r * S + t;
      } else if (v == best && (r * S + t) < best_idx) {
        best_idx = r * S + t;   // and the smallest row index among equal values, whichever wave saw it
      }
    }
    task = next_task();
  }
  BV[wave * 64 + lane] = best;
  BI[wave * 64 + lane] = best_idx;
  __syncthreads();
  if (wave == 0) {
    double b = BV[lane];
    int bi = BI[lane];
    for (int w = 1; w < RM_NW; ++w) {
      const double v = BV[w * 64 + lane];
      const int vi = BI[w * 64 + lane];
      if (v > b || (v == b && vi < bi)) {
        b = v;
        bi = vi;
      }
    }
    if (valid) {
      m_out[m] = (float)b;
      arg_out[m] = bi;
    }
  }
}

// Cuts under a linearization.  A cut of the CVaR constraint is a tail weighting (w_i in [0,1], sum = alpha M: 1 above
// the threshold of the rato_risk_stats record that was computed on the m values, lambda on ties with it) plus one row
// r_i per sample; under ANY linearization  CVaR_alpha(m(x)) >= (1/(alpha M)) sum_i w_i [(G_i x)_{r_i} + sign base_{i,r_i}],
// with equality at the x the (w, r) were computed for.  This kernel evaluates K of them in one launch: blockIdx.y = k
// picks ring slot slots[k] (slots == NULL: K = 1, slot 0); part[blk][k][0 .. 2(S-1)) = block sums of
// w_i G_i[r_i, (s,g)], part[blk][k][2(S-1)] = block sum of w_i base_{i,r_i}: fp64 throughout.
template <int R>
__global__ __launch_bounds__(RATO_BLOCK) void tail_rows_batch_kernel(
    const float* __restrict__ G, const float* __restrict__ W, long ld, int tileW, int S, long M,
    const float* __restrict__ base, const float* __restrict__ m_base, const int* __restrict__ arg_base,
    const double* __restrict__ stats_base, long stats_stride, const int* __restrict__ slots, double alphaM,
    double* __restrict__ part) {
  extern __shared__ double trb_lds[];   // [4 waves][2*(S-1) + 1]
  const int K = gridDim.y, kk = blockIdx.y;
  const long slot = slots ? slots[kk] : 0;
  const float* __restrict__ mvals = m_base + slot * M;
  const int* __restrict__ arg = arg_base + slot * M;
  float tstar, lambda;
  tail_rule(stats_base + slot * stats_stride, alphaM, tstar, lambda);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float w = 0.0f;
  int t = 0, r = 0;
  {
    const long m0 = (long)blockIdx.x * RATO_BLOCK + threadIdx.x;
    if (m0 < M) {
      w = tail_weight(mvals[m0], tstar, lambda);
      const int a = arg[m0];
      r = a / S;
      t = a - r * S;
    }
  }
  TailLane tl;
  const int n_tail = compact_tail(w, t, r, (long)blockIdx.x * RATO_BLOCK, M, tl);   // tail samples -> lanes [0, n_tail)
  const int work_waves = (n_tail + RATO_WAVE - 1) / RATO_WAVE;
  const long m = tl.m;
  w = tl.w;
  t = tl.t;
  r = tl.r;
  const size_t n_pairs = (size_t)S * (S - 1) / 2;
  const bool fact = (W != nullptr);
  const int RR = fact ? 1 : R;
  const float* __restrict__ Gm = G + (size_t)(m / tileW) * rato::packed_tile_stride(n_pairs * 2 * RR * tileW) + (m % tileW);
  const float* __restrict__ row = Gm + (size_t)rato::pair_row_offset(t) * (2 * RR * tileW);
  double w0 = w, w1 = w, wg = 0.0;
  if (w != 0.0f) {
    wg = (double)w * (double)base[((size_t)r * S + t) * ld + m];
    if (fact) {   // factored: entry = W[r,t,a] * Phi[t,s,a]
      w0 = (double)w * (double)W[(((size_t)r * S + t) * 2 + 0) * ld + m];
      w1 = (double)w * (double)W[(((size_t)r * S + t) * 2 + 1) * ld + m];
    }
  }
  const int rsel = fact ? 0 : r;
  const int nw = 2 * (S - 1), nc = nw + 1;
  constexpr int SB = 8;   // columns per batch: 16 gathers in flight, then 16 wave reductions
  for (int sb = 0; sb < S - 1 && wave < work_waves; sb += SB) {
    float g0[SB], g1[SB];
#pragma unroll
    for (int i = 0; i < SB; ++i) {
      const int s2 = sb + i;
      g0[i] = 0.0f;
      g1[i] = 0.0f;
      if (w != 0.0f && s2 < t) {
        const float* __restrict__ o = row + (size_t)s2 * (2 * RR * tileW);
        g0[i] = o[rsel * tileW];
        g1[i] = o[(RR + rsel) * tileW];
      }
    }
#pragma unroll
    for (int i = 0; i < SB; ++i) {
      const int s2 = sb + i;
      if (s2 < S - 1) {
        const double s0 = rato::wave_sum_dpp(w0 * (double)g0[i]);
        const double s1 = rato::wave_sum_dpp(w1 * (double)g1[i]);
        if (lane == 0) {
          trb_lds[wave * nc + s2 * 2 + 0] = s0;
          trb_lds[wave * nc + s2 * 2 + 1] = s1;
        }
      }
    }
  }
  if (wave < work_waves) {
    const double sg = rato::wave_sum_dpp(wg);
    if (lane == 0) trb_lds[wave * nc + nw] = sg;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < nc; i += RATO_BLOCK) {
    double acc = 0.0;
    for (int wv = 0; wv < work_waves; ++wv) acc += trb_lds[wv * nc + i];   // fixed order
    part[((size_t)blockIdx.x * K + kk) * nc + i] = acc;
  }
}

// Jacobian-free tail rows for the drone (one cut or K kept cuts, output layout of tail_rows_batch_kernel): the arg-max
// row (j*, t*) of a tail sample is  W[j*,t*,a] Phi[t*,s,a],  Phi[t*, k-1, a] = (mu_k)[1] dt/m  with  mu_{t*+1} = e_0',
// mu_k = mu_{k+1} A_k  -- the adjoint sweep of the linearize kernel, regenerated here (fp64) from A22.  The lanes walk
// k from S-1 down together; a lane joins at k = t*_i.
__global__ __launch_bounds__(RATO_BLOCK) void drone_tail_rows_implicit_kernel(
    rato_drone_params P, const float* __restrict__ mass, const float* __restrict__ A22, int a22_axes,
    const float* __restrict__ W, const float* __restrict__ base, const float* __restrict__ m_base,
    const int* __restrict__ arg_base, const double* __restrict__ stats_base, long stats_stride,
    const int* __restrict__ slots, double alphaM, double* __restrict__ part) {
  extern __shared__ double tri_lds[];   // [4 waves][2*(S-1) + 1]
  const int S = P.S;
  const long M = P.M, ld = P.ld;
  const int K = gridDim.y, kk = blockIdx.y;
  const long slot = slots ? slots[kk] : 0;
  const float* __restrict__ mvals = m_base + slot * M;
  const int* __restrict__ arg = arg_base + slot * M;
  float tstar, lambda;
  tail_rule(stats_base + slot * stats_stride, alphaM, tstar, lambda);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float w = 0.0f;
  int t = 0, r = 0;
  {
    const long m0 = (long)blockIdx.x * RATO_BLOCK + threadIdx.x;
    if (m0 < M) {
      w = tail_weight(mvals[m0], tstar, lambda);
      const int a = arg[m0];
      r = a / S;
      t = a - r * S;
    }
  }
  TailLane tl;
  const int n_tail = compact_tail(w, t, r, (long)blockIdx.x * RATO_BLOCK, M, tl);   // tail samples -> lanes [0, n_tail)
  const int work_waves = (n_tail + RATO_WAVE - 1) / RATO_WAVE;
  const long m = tl.m;
  w = tl.w;
  t = tl.t;
  r = tl.r;
  const double dt = P.dt64;
  double w0 = 0.0, w1 = 0.0, wg = 0.0, a21 = 0.0;
  if (w != 0.0f) {
    const double inv_m = 1.0 / (double)mass[m];
    a21 = -P.kp64 * dt * inv_m;
    const double dtm = dt * inv_m;
    w0 = (double)w * (double)W[(((size_t)r * S + t) * 2 + 0) * ld + m] * dtm;
    w1 = (double)w * (double)W[(((size_t)r * S + t) * 2 + 1) * ld + m] * dtm;
    wg = (double)w * (double)base[((size_t)r * S + t) * ld + m];
  }
  const int nw = 2 * (S - 1), nc = nw + 1;
  double m0x = 0.0, m1x = 0.0, m0y = 0.0, m1y = 0.0;
  constexpr int SB = 8;
  for (int kb = S - 1; kb >= 1 && wave < work_waves; kb -= SB) {
    float ax[SB], ay[SB];
#pragma unroll
    for (int i = 0; i < SB; ++i) {   // the a22 of the batch for the lanes that are in the sweep at that step
      const int k = kb - i;
      const bool on = (w != 0.0f) && k >= 1 && k <= t;
      ax[i] = on ? A22[((size_t)k * a22_axes + 0) * ld + m] : 0.0f;
      ay[i] = on ? A22[((size_t)k * a22_axes + 1) * ld + m] : 0.0f;
    }
    const bool compl22 = (a22_axes == 3);   // the generators' table holds 1 - a22 (see a22_value)
#pragma unroll
    for (int i = 0; i < SB; ++i) {
      const int k = kb - i;
      if (k >= 1) {   // wave-uniform
        const bool on = (w != 0.0f) && k <= t;
        if (on && k == t) {
          m0x = 1.0; m1x = 0.0; m0y = 1.0; m1y = 0.0;   // mu_{t*+1} = e_0'
        }
        double cx = 0.0, cy = 0.0;
        if (on) {
          const double n0x = m0x + m1x * a21, n1x = m0x * dt + m1x * a22_value(ax[i], compl22);
          const double n0y = m0y + m1y * a21, n1y = m0y * dt + m1y * a22_value(ay[i], compl22);
          m0x = n0x; m1x = n1x; m0y = n0y; m1y = n1y;
          cx = w0 * m1x;
          cy = w1 * m1y;
        }
        const double s0 = rato::wave_sum_dpp(cx);
        const double s1 = rato::wave_sum_dpp(cy);
        if (lane == 0) {
          tri_lds[wave * nc + (k - 1) * 2 + 0] = s0;
          tri_lds[wave * nc + (k - 1) * 2 + 1] = s1;
        }
      }
    }
  }
  if (wave < work_waves) {
    const double sg = rato::wave_sum_dpp(wg);
    if (lane == 0) tri_lds[wave * nc + nw] = sg;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < nc; i += RATO_BLOCK) {
    double acc = 0.0;
    for (int wv = 0; wv < work_waves; ++wv) acc += tri_lds[wv * nc + i];   // fixed order
    part[((size_t)blockIdx.x * K + kk) * nc + i] = acc;
  }
}

// Jacobian-free form of rowmax for the drone: one lane per sample, one pass over the step-Jacobian table
// A22 [S][axes][ld], W [3][S][2][ld] and base [3][S][ld] (11 S floats per sample).  (G_i x)_{j,t} =
// W[j,t,x] dp_x(t+1) + W[j,t,y] dp_y(t+1) with d x_{k+1} = A_k d x_k + B x_k, d x_0 = 0 — the forward form of
// the adjoint sweep that produced Phi (drone.hip), evaluated in fp64 on the fp32 tables.
__global__ __launch_bounds__(RATO_BLOCK) void drone_rowmax_implicit_kernel(
    rato_drone_params P, const float* __restrict__ mass, const float* __restrict__ A22, int a22_axes,
    const float* __restrict__ W, const float* __restrict__ base, double sign, const double* __restrict__ xs,
    float* __restrict__ m_out, int* __restrict__ arg_out) {
  const long m = (long)blockIdx.x * RATO_BLOCK + threadIdx.x;
  if (m >= P.M) return;
  const size_t ld = (size_t)P.ld;
  const int S = P.S;
  const double dt = P.dt64;
  const double inv_m = 1.0 / (double)mass[m];
  const double a21 = -P.kp64 * dt * inv_m, dtm = dt * inv_m;
  const bool compl22 = (a22_axes == 3);            // the generators' table holds 1 - a22 (see a22_value)
  double px = 0.0, vx = 0.0, py = 0.0, vy = 0.0;   // d x_t (t = 0)
  double best = -INFINITY;
  int best_idx = 0;
  constexpr int TB = 8;   // steps per batch: the 11 loads of a step do not depend on the recursion, so a batch
                          // issues 88 loads back to back and only then runs its 8 dependent steps
  for (int t0 = 0; t0 < S; t0 += TB) {
    float a2[TB][2], wv[TB][3][2], gu[TB][3];
#pragma unroll
    for (int i = 0; i < TB; ++i) {
      const int t = (t0 + i < S) ? t0 + i : S - 1;
      a2[i][0] = A22[((size_t)t * a22_axes + 0) * ld + m];
      a2[i][1] = A22[((size_t)t * a22_axes + 1) * ld + m];
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        wv[i][j][0] = W[(((size_t)j * S + t) * 2 + 0) * ld + m];
        wv[i][j][1] = W[(((size_t)j * S + t) * 2 + 1) * ld + m];
        gu[i][j] = base[((size_t)j * S + t) * ld + m];
      }
    }
#pragma unroll
    for (int i = 0; i < TB; ++i) {
      const int t = t0 + i;
      if (t < S) {
        const double ux = xs[t * 3 + 0], uy = xs[t * 3 + 1];
        const double npx = px + dt * vx, npy = py + dt * vy;
        const double nvx = a21 * px + a22_value(a2[i][0], compl22) * vx + dtm * ux;
        const double nvy = a21 * py + a22_value(a2[i][1], compl22) * vy + dtm * uy;
        px = npx; py = npy; vx = nvx; vy = nvy;             // d x_{t+1}
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          const double v = (double)wv[i][j][0] * px + (double)wv[i][j][1] * py + sign * (double)gu[i][j];
          const int r = j * S + t;
          if (v > best || (v == best && r < best_idx)) {   // smallest row index among equal values
            best = v;
            best_idx = r;
          }
        }
      }
    }
  }
  m_out[m] = (float)best;
  arg_out[m] = best_idx;
}

// ---------------------------------------------------------------------------------------------------------------------
// Table-free form of the drone oracle ("rollout" form).  rows(u) = g(u_k) + grad g(u_k) . (u - u_k) needs, per sample,
// the rollout at the linearization point u_k; instead of reading what a linearize kernel stored of it in fp32 (A22, W,
// g: 44 S bytes per sample, each number carrying 6e-8 of ITS magnitude -- 6e-6 absolute on rows with |g| ~ 1e2, which
// is what an SCP subproblem with an O(1) step saw as 1e-5 in u), the oracle RE-RUNS that rollout in fp64 from the
// samples themselves (noise of the two horizontal axes, mass, Q: 8 S + 40 bytes per sample) while it propagates the
// response to x = u - u_k.  5 x less HBM traffic than the tables, and no stored intermediate at all: what is left of the
// fp32 device path in the rows is the rounding of its INPUTS.  u_k and x are doubles ([S][3]).
// BYVAL: x travels in the kernel arguments (S n_u <= XARG_MAX doubles) instead of device memory -- the one-call round
// trip (rato_cut_oracle_rollout) then needs no upload in front of its first launch (an asynchronous 1.2 KB copy costs
// ~25 us of host time on this stack, a quarter of the device time of the whole trip).
// v_max_f64 as it is (fmax() first canonicalises both operands: two more 64-bit instructions per call); a NaN operand
// yields the other one, like fmax
__device__ __forceinline__ double max_f64(double a, double b) {
  double r;
  asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
constexpr int XARG_MAX = 192;
struct XArg {
  double v[XARG_MAX];
};

template <bool BYVAL>
__global__ __launch_bounds__(RATO_BLOCK) void drone_rowmax_rollout_kernel(
    rato_drone_params P, const double* __restrict__ uk, const float* __restrict__ dW, const float* __restrict__ mass,
    const float* __restrict__ Qsym, const double* __restrict__ xs_mem, const XArg xv, float* __restrict__ m_out,
    int* __restrict__ arg_out) {
  const double* __restrict__ xs = BYVAL ? xv.v : xs_mem;
  const long m = (long)blockIdx.x * RATO_BLOCK + threadIdx.x;
  if (m >= P.M) return;
  const size_t ld = (size_t)P.ld;
  const int S = P.S;
  const double dt = P.dt64, kp = P.kp64, kd = P.kd64, drag = P.drag64;
  const double inv_m = 1.0 / (double)mass[m];
  const double a21 = -kp * dt * inv_m, dtm = dt * inv_m, cn = sqrt(dt) * P.beta64 * inv_m;
  // The kernel is bound by fp64 instruction ISSUE (round 4: two lanes per sample made it slower), so the step is written
  // for the fewest operations: per axis, with c1 = dt kd / m, c2 = dt drag / m,
  //   a22 = (1 - c1) - 2 c2 |v|,       v' = ((1 - c1) - c2 |v|) v + (dt/m) u - (kp dt/m) p + cn xi,       p' = p + dt v
  // (the reference's  v + dt ((u - (kp p + kd v)) / m - drag |v| v / m)  multiplied out: 12 operations instead of 19),
  // and an obstacle row as  1 - [dx (q00 (dx + 2 dp_x) + qs (dy + dp_y)) + dy (q11 (dy + 2 dp_y) + qs dp_x)]
  // (= g + grad g . dp: 11 operations instead of 16).  One running maximum per obstacle, merged at the end.
  const double one_c1 = 1.0 - dt * kd * inv_m, c2 = dt * drag * inv_m, c22 = 2.0 * c2;
  // (round 5: the obstacle's centre leaves the two inner sums as per-sample constants -- kx = q00 ox + qs oy,
  //  ky = q11 oy: 8 operations per row instead of 11 -- and the running maximum is a v_max_f64 with the step index selected
  //  beside it: 3 instead of 4.  The kernel is bound by the ~10 cycles between two instructions of the only wave or two
  //  on a SIMD, tools/fp64bench.hip: every instruction less is time.)
  double q00[3], qs[3], q11[3], kx[3], ky[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    q00[j] = (double)Qsym[(size_t)(j * 3 + 0) * ld + m];
    qs[j] = (double)Qsym[(size_t)(j * 3 + 1) * ld + m];
    q11[j] = (double)Qsym[(size_t)(j * 3 + 2) * ld + m];
    kx[j] = fma(q00[j], P.obs_xy64[j][0], qs[j] * P.obs_xy64[j][1]);
    ky[j] = q11[j] * P.obs_xy64[j][1];
  }
  double p[2] = {P.x_init64[0], P.x_init64[1]}, v[2] = {P.x_init64[3], P.x_init64[4]};
  double dp[2] = {0.0, 0.0}, dv[2] = {0.0, 0.0};
  double best[3] = {-INFINITY, -INFINITY, -INFINITY};
  int best_t[3] = {0, 0, 0};
  // The loads are batched: the noise of 8 steps is requested as one batch, and the NEXT batch is already in flight while
  // a batch is consumed (two register buffers, the loop unrolled by two).
  constexpr int TB = 8;
  auto load = [&](float (&xi)[TB][2], int t0) {
#pragma unroll
    for (int i = 0; i < TB; ++i) {
      const int t = (t0 + i < S) ? t0 + i : S - 1;
      xi[i][0] = dW[((size_t)t * 3 + 0) * ld + m];
      xi[i][1] = dW[((size_t)t * 3 + 1) * ld + m];
    }
  };
  auto steps = [&](const float (&xi)[TB][2], int t0, auto guarded) {
#pragma unroll
    for (int i = 0; i < TB; ++i) {
      const int t = t0 + i;
      if (!decltype(guarded)::value || t < S) {
#pragma unroll
        for (int a = 0; a < 2; ++a) {
          const double av = fabs(v[a]);                           // at the state BEFORE the step
          const double a22 = fma(-c22, av, one_c1);
          const double ndp = fma(dt, dv[a], dp[a]);
          dv[a] = fma(a21, dp[a], fma(a22, dv[a], dtm * xs[t * 3 + a]));
          dp[a] = ndp;
          const double sv = fma(-c2, av, one_c1);
          const double tv = fma(a21, p[a], fma(dtm, uk[t * 3 + a], cn * (double)xi[i][a]));
          const double pn = fma(dt, v[a], p[a]);
          v[a] = fma(sv, v[a], tv);
          p[a] = pn;
        }
        const double px2 = fma(2.0, dp[0], p[0]), py2 = fma(2.0, dp[1], p[1]), sy = p[1] + dp[1];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          const double dx = p[0] - P.obs_xy64[j][0], dy = p[1] - P.obs_xy64[j][1];
          const double ix = fma(qs[j], sy, fma(q00[j], px2, -kx[j]));     // q00 (dx + 2 dp_x) + qs (dy + dp_y)
          const double iy = fma(qs[j], dp[0], fma(q11[j], py2, -ky[j]));  // q11 (dy + 2 dp_y) + qs dp_x
          const double val = fma(-dy, iy, fma(-dx, ix, 1.0));
          best_t[j] = (val > best[j]) ? t : best_t[j];            // ascending t: the smallest t among equal values
          best[j] = max_f64(best[j], val);
        }
      }
    }
  };
  float xa[TB][2], xb[TB][2];
  load(xa, 0);
  int t0 = 0;
  // whole double batches without a test per step (a guarded step is a conditional block: its merges cost 64-bit moves);
  // the last S mod 16 steps behind their guards
  for (; t0 + 2 * TB <= S; t0 += 2 * TB) {
    load(xb, t0 + TB);
    steps(xa, t0, std::false_type{});
    load(xa, t0 + 2 * TB);
    steps(xb, t0 + TB, std::false_type{});
  }
  if (t0 < S) {
    load(xb, t0 + TB);
    steps(xa, t0, std::true_type{});
    steps(xb, t0 + TB, std::true_type{});
  }
  // merge the three maxima: the larger value, the smaller row index r = j S + t among equal values
  double bv = best[0];
  int bi = best_t[0];
#pragma unroll
  for (int j = 1; j < 3; ++j)
    if (best[j] > bv) {
      bv = best[j];
      bi = j * S + best_t[j];
    }
  m_out[m] = (float)bv;
  arg_out[m] = bi;
}

// maximum of a non-negative int over the wave, in every lane (DPP tree: no LDS crossbar round trips)
__device__ __forceinline__ unsigned wave_max_nonneg_dpp(int v) {
  unsigned a = (unsigned)v;
  a = max(a, (unsigned)__builtin_amdgcn_update_dpp(0, (int)a, 0x111, 0xf, 0xf, false));
  a = max(a, (unsigned)__builtin_amdgcn_update_dpp(0, (int)a, 0x112, 0xf, 0xf, false));
  a = max(a, (unsigned)__builtin_amdgcn_update_dpp(0, (int)a, 0x114, 0xf, 0xf, false));
  a = max(a, (unsigned)__builtin_amdgcn_update_dpp(0, (int)a, 0x118, 0xf, 0xf, false));
  a = max(a, (unsigned)__builtin_amdgcn_update_dpp(0, (int)a, 0x142, 0xa, 0xf, false));
  a = max(a, (unsigned)__builtin_amdgcn_update_dpp(0, (int)a, 0x143, 0xc, 0xf, false));
  return (unsigned)__builtin_amdgcn_readlane((int)a, 63);
}

#ifndef RATO_TDIAG
#define RATO_TDIAG 0   // diagnostic build (tools/tail_timeline.py): 100 MHz ticks of the phases of drone_tail_rows_rollout_kernel
#endif                 // replace the block's row of `part` (thread 0: wave 0 = the x axis)
#if RATO_TDIAG
#define TSTAMP(i) do { if (threadIdx.x == 0) tdiag[i] = (double)wall_clock64(); } while (0)
__shared__ double tdiag[12];
#else
#define TSTAMP(i) do { } while (0)
#endif

constexpr int TAIL_CSTRIDE = RATO_WAVE + 1;   // row stride of the sweep's term table (doubles): column threads on distinct banks
__host__ __device__ inline size_t tail_ctab_doubles(int S) { return (size_t)2 * (S - 1) * TAIL_CSTRIDE; }

// The tail phase of ONE cut for one block of 256 samples: thread i brings its sample's tail weight and arg-max row (step t0,
// row group r0); on return (behind a barrier) acc [2 (S - 1) + 1] holds the block's column sums.  E: [S][2][64] floats.
__device__ __forceinline__ void drone_tail_block(const rato_drone_params& P, const double* __restrict__ uk,
                                                 const float* __restrict__ dW, const float* __restrict__ mass,
                                                 const float* __restrict__ Qsym, float w0f, int t0, int r0,
                                                 double* __restrict__ acc, float* __restrict__ E,
                                                 double* __restrict__ Ctab) {
  const int S = P.S;
  const long M = P.M, ld = P.ld;
  const int nw = 2 * (S - 1), nc = nw + 1;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  TailLane unused;
  TailLists lists;
  TSTAMP(1);
  const int n_tail = compact_tail(w0f, t0, r0, (long)blockIdx.x * RATO_BLOCK, M, unused, &lists);
  for (int i = threadIdx.x; i < nc; i += RATO_BLOCK) acc[i] = 0.0;
  __shared__ double s_pt[2][RATO_WAVE];   // p_{t*+1} of the chunk's samples, one axis per wave
  __syncthreads();
  TSTAMP(2);
  // The two horizontal axes are independent chains (forward and adjoint) that meet only in the arg-max row's gradient:
  // wave a runs axis a.  One wave running both was bound by the latency of its own chain (a block has one wave of work).
  const double dt = P.dt64, kp = P.kp64, kd = P.kd64, drag = P.drag64;
  const int a = wave;   // (waves 2, 3 only keep the barriers company)
  const bool cols_in_regs = S - 1 <= RATO_WAVE;
  double colsum = 0.0;   // lane j: column j of axis a, summed over the chunks (chunk order = sample order)
  for (int c0 = 0; c0 < n_tail; c0 += RATO_WAVE) {   // chunks of 64 tail samples, in sample order
    const bool on = c0 + lane < n_tail;
    const long m = on ? (long)blockIdx.x * RATO_BLOCK + lists.src[c0 + lane] : 0;
    const double w = on ? (double)lists.w[c0 + lane] : 0.0;
    const int tr = on ? lists.tr[c0 + lane] : 0;
    const int ts = tr & 0xfffff, rs = tr >> 20;
    // wave-uniform: the forward pass only has to reach the largest t* of the chunk
    const int t_hi = (int)wave_max_nonneg_dpp(on ? ts : 0);
    double inv_m = 0.0, a21 = 0.0, dtm = 0.0;
    float qf[3] = {0.0f, 0.0f, 0.0f};
    if (a < 2) {
      // (everything the chunk reads from global memory is requested here, in front of the forward pass: the arg-max row's
      //  Q used to be fetched behind it -- one more memory round trip in the chain of the block's only working waves)
      const float mass_f = mass[m];
      if (on) {
#pragma unroll
        for (int c = 0; c < 3; ++c) qf[c] = Qsym[(size_t)(rs * 3 + c) * ld + m];
      }
      inv_m = 1.0 / (double)mass_f;
      a21 = -kp * dt * inv_m;
      dtm = dt * inv_m;
      const double cn = sqrt(dt) * P.beta64 * inv_m;
      const double c1 = dt * kd * inv_m, one_c1 = 1.0 - c1, c2 = dt * drag * inv_m, c22 = 2.0 * c2;
      double p = P.x_init64[a], v = P.x_init64[3 + a];
      constexpr int TB = 8;   // noise in batches of 8 steps, the next batch in flight while one is consumed
      auto load = [&](float (&xi)[TB], int tb) {
#pragma unroll
        for (int i = 0; i < TB; ++i) {
          const int t = (tb + i < S) ? tb + i : S - 1;
          xi[i] = dW[((size_t)t * 3 + a) * ld + m];
        }
      };
      auto steps = [&](const float (&xi)[TB], int tb) {
#pragma unroll
        for (int i = 0; i < TB; ++i) {
          const int t = tb + i;
          if (t <= t_hi) {
            // the step of drone_rowmax_rollout_kernel, operation for operation (same trajectory, bit for bit: the row the
            // cut is built from is evaluated where the maximum was found); 8 fp64 operations, 3 of them in the chain
            const double av = fabs(v);
            E[(t * 2 + a) * RATO_WAVE + lane] = (float)fma(c22, av, c1);      // e22 = dt (kd + 2 drag |v|) / m
            const double tv = fma(a21, p, fma(dtm, uk[t * 3 + a], cn * (double)xi[i]));
            const double pn = fma(dt, v, p);
            v = fma(fma(-c2, av, one_c1), v, tv);
            p = pn;
            if (on && t == ts) s_pt[a][lane] = p;   // the arg-max row of this sample is evaluated at p_{t*+1}
          }
        }
      };
      float xa[TB], xb[TB];
      load(xa, 0);
#if RATO_TDIAG
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      TSTAMP(3);
#endif
      for (int tb = 0; tb <= t_hi; tb += 2 * TB) {
        load(xb, tb + TB);
        steps(xa, tb);
        load(xa, tb + 2 * TB);
        steps(xb, tb + TB);
      }
      TSTAMP(4);
    }
    __syncthreads();   // both axes' positions are in s_pt
    TSTAMP(5);
    if (a < 2) {
      // g and grad_p g of the arg-max row at p_{t*+1}; adjoint sweep of this wave's axis:
      // mu_{t*+1} = e_0', mu_k = mu_{k+1} A_k; column k-1 of the row = W . (mu_k)[1] dt/m
      double wa = 0.0, gval = 0.0;
      if (on) {
        const double q00 = (double)qf[0], qss = (double)qf[1], q11 = (double)qf[2];
        const double dx = s_pt[0][lane] - P.obs_xy64[rs][0], dy = s_pt[1][lane] - P.obs_xy64[rs][1];
        gval = 1.0 - (q00 * dx * dx + qss * dx * dy + q11 * dy * dy);
        const double wx = -(2.0 * q00 * dx + qss * dy), wy = -(qss * dx + 2.0 * q11 * dy);
        wa = w * (a == 0 ? wx : wy) * dtm;
      }
      double m0 = 0.0, m1 = 0.0;
      float e_k = (t_hi >= 1) ? E[(t_hi * 2 + a) * RATO_WAVE + lane] : 0.0f;   // (read one step ahead of its use)
      double* __restrict__ Ca = Ctab ? Ctab + (size_t)a * (S - 1) * TAIL_CSTRIDE : nullptr;
      for (int k = t_hi; k >= 1; --k) {   // wave-uniform
        const float e_next = (k > 1) ? E[((k - 1) * 2 + a) * RATO_WAVE + lane] : 0.0f;
        const bool in = on && k <= ts;
        if (in && k == ts) {
          m0 = 1.0; m1 = 0.0;
        }
        double c = 0.0;
        if (in) {
          const double aa = 1.0 - (double)e_k;
          const double n0 = m0 + m1 * a21, n1 = m0 * dt + m1 * aa;
          m0 = n0; m1 = n1;
          c = wa * m1;
        }
        e_k = e_next;
        if (Ca) {   // the lanes' terms of column k-1 go to LDS as they are; summed over the lanes after the sweep
          Ca[(k - 1) * TAIL_CSTRIDE + lane] = c;
          continue;
        }
        const double s0 = rato::wave_sum_dpp(c);
        // column k-1 of this axis: kept in lane k-1's register while the horizon fits a wave (no LDS read-modify-write
        // in the chain), in LDS otherwise
        if (cols_in_regs) colsum += (lane == k - 1) ? s0 : 0.0;
        else if (lane == 0) acc[(k - 1) * 2 + a] += s0;
      }
      if (a == 0) {
        const double sg = rato::wave_sum_dpp(w * gval);
        if (lane == 0) acc[nw] += sg;
      }
      TSTAMP(6);
    }
    __syncthreads();   // s_pt and the tables are free for the next chunk
    TSTAMP(7);
    if (Ctab) {
      // columns 0 .. t_hi - 1 of both axes: one thread per (axis, column) adds the 64 lanes' terms in lane order.  (In
      // the sweep a wave-wide fp64 sum per step was 20 of the ~35 instructions of a step of the only wave on its SIMD.)
      for (int i = threadIdx.x; i < 2 * t_hi; i += RATO_BLOCK) {
        const int ax = i / t_hi, j = i - ax * t_hi;
        const double* __restrict__ row = Ctab + ((size_t)ax * (S - 1) + j) * TAIL_CSTRIDE;
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll 4
        for (int l = 0; l < 16; ++l) {
          s0 += row[l];
          s1 += row[16 + l];
          s2 += row[32 + l];
          s3 += row[48 + l];
        }
        acc[j * 2 + ax] += (s0 + s1) + (s2 + s3);
      }
      __syncthreads();
    }
  }
  if (!Ctab && cols_in_regs && a < 2 && lane < S - 1) acc[lane * 2 + a] = colsum;
  __syncthreads();
  TSTAMP(8);
#if RATO_TDIAG
  if (threadIdx.x == 0) tdiag[9] = (double)n_tail;
#endif
}

// The cut of the rollout form: the tail samples of the block (compacted, walked by wave 0 in chunks of 64) re-run the
// rollout in fp64 up to their own t*, leaving e22 = 1 - a22 of both axes in LDS ([S][2][64] floats: exact to 1e-10 of
// a22), pick up W and g of their arg-max row on the way, and then run the adjoint sweep from t* down, exactly as
// drone_tail_rows_implicit_kernel does from its table.  Output layout of tail_rows_batch_kernel (offset sum = sum w g).
__global__ __launch_bounds__(RATO_BLOCK) void drone_tail_rows_rollout_kernel(
    rato_drone_params P, const double* __restrict__ uk, const float* __restrict__ dW, const float* __restrict__ mass,
    const float* __restrict__ Qsym, const float* __restrict__ m_base, const int* __restrict__ arg_base,
    const double* __restrict__ stats_base, long stats_stride, const int* __restrict__ slots, double alphaM,
    double* __restrict__ part, int c_tab) {
  extern __shared__ __attribute__((aligned(16))) unsigned char trr_lds[];
  const int S = P.S;
  const long M = P.M, ld = P.ld;
  const int nw = 2 * (S - 1), nc = nw + 1;
  double* acc = reinterpret_cast<double*>(trr_lds);                   // [nc] column sums of the block
  TSTAMP(0);
  double* Ctab = c_tab ? acc + nc : nullptr;                          // [2][S-1][TAIL_CSTRIDE] the sweep's terms per lane
  float* E = reinterpret_cast<float*>(acc + nc + (c_tab ? tail_ctab_doubles(S) : 0));   // [S][2][64] e22 of the current chunk
  const int K = gridDim.y, kk = blockIdx.y;
  const long slot = slots ? slots[kk] : 0;
  const float* __restrict__ mvals = m_base + slot * M;
  const int* __restrict__ arg = arg_base + slot * M;
  float tstar, lambda;
  tail_rule(stats_base + slot * stats_stride, alphaM, tstar, lambda);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float w0f = 0.0f;
  int t0 = 0, r0 = 0;
  {
    const long m0 = (long)blockIdx.x * RATO_BLOCK + threadIdx.x;
    if (m0 < M) {
      w0f = tail_weight(mvals[m0], tstar, lambda);
      const int a = arg[m0];
      r0 = a / S;
      t0 = a - r0 * S;
    }
  }
  drone_tail_block(P, uk, dW, mass, Qsym, w0f, t0, r0, acc, E, Ctab);
#if RATO_TDIAG
  __syncthreads();
  if (threadIdx.x < 12) acc[threadIdx.x] = threadIdx.x == 10 ? (double)wall_clock64() : tdiag[threadIdx.x];
  __syncthreads();
#endif
  for (int i = threadIdx.x; i < nc; i += RATO_BLOCK) part[((size_t)blockIdx.x * K + kk) * nc + i] = acc[i];
}

// K > 1 (the cuts kept from the previous subproblem, re-linearized at the new u_k): ONE pass per block of 256 samples for
// all the cuts.  The rollout at u_k is the same for every cut and their tails are nearly the same samples, so the UNION of
// the tails is compacted (in sample order) and walked in chunks of 64: wave 0 re-runs the rollout once per chunk (e22 of
// both axes into LDS; at every cut's own t* the position the row is evaluated at), then the waves take a cut each and sweep its
// adjoint over the chunk.  Launched per cut (gridDim.y = K, the kernel above) the K x ceil(M/256) one-wave workgroups
// took 104 us for 9 cuts at M = 1e5 -- five latency-bound rounds; this form costs one forward pass + ceil(K / 8) sweeps.
#ifndef RATO_TRU_NW
#define RATO_TRU_NW 8   // waves per workgroup of the union pass (A/B builds)
#endif
constexpr int TRU_NW = RATO_TRU_NW, TRU_KMAX = 16;
__host__ __device__ inline size_t tail_union_lds_bytes(int S, int K) {
  const size_t nc = 2 * (size_t)(S - 1) + 1;
  return sizeof(double) * ((size_t)K * nc + (size_t)K * 2 * RATO_WAVE) + sizeof(float) * (size_t)S * 2 * RATO_WAVE +
         (sizeof(float) + sizeof(int)) * (size_t)K * RATO_WAVE + sizeof(unsigned) * (size_t)S;
}
__global__ __launch_bounds__(TRU_NW* RATO_WAVE) void drone_tail_rows_rollout_union_kernel(
    rato_drone_params P, const double* __restrict__ uk, const float* __restrict__ dW, const float* __restrict__ mass,
    const float* __restrict__ Qsym, const float* __restrict__ m_base, const int* __restrict__ arg_base,
    const double* __restrict__ stats_base, long stats_stride, const int* __restrict__ slots, int k0, int K, int K_total,
    double alphaM, double* __restrict__ part) {
  extern __shared__ __attribute__((aligned(16))) unsigned char tru_lds[];
  constexpr int NT = TRU_NW * RATO_WAVE;
  const int S = P.S;
  const long M = P.M, ld = P.ld;
  const int nw = 2 * (S - 1), nc = nw + 1;
  double* acc = reinterpret_cast<double*>(tru_lds);                        // [K][nc] column sums of the block, per cut
  double* GW = acc + (size_t)K * nc;                                       // [K][2][64] p_{t*+1} of (cut, lane)
  float* E = reinterpret_cast<float*>(GW + (size_t)K * 2 * RATO_WAVE);     // [S][2][64] e22 of the current chunk
  float* WT = E + (size_t)S * 2 * RATO_WAVE;                               // [K][64] tail weight of (cut, lane)
  int* TR = reinterpret_cast<int*>(WT + (size_t)K * RATO_WAVE);            // [K][64] arg-max row: t | r << 20
  unsigned* TMASK = reinterpret_cast<unsigned*>(TR + (size_t)K * RATO_WAVE);   // [S] the cuts with a lane whose t* is this step
  __shared__ int s_src[RATO_BLOCK];
  __shared__ int s_cnt[RATO_BLOCK / RATO_WAVE];
  __shared__ float s_ts[TRU_KMAX], s_lam[TRU_KMAX];
  __shared__ long s_slot[TRU_KMAX];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long block_base = (long)blockIdx.x * RATO_BLOCK;
  if (tid < K) {
    const long slot = slots[k0 + tid];
    float ts, lam;
    tail_rule(stats_base + slot * stats_stride, alphaM, ts, lam);
    s_ts[tid] = ts;
    s_lam[tid] = lam;
    s_slot[tid] = slot;
  }
  for (int i = tid; i < K * nc; i += NT) acc[i] = 0.0;
  __syncthreads();
  // ---- the union of the K tails, compacted in sample order
  bool act = false;
  if (tid < RATO_BLOCK && block_base + tid < M) {
    for (int kk = 0; kk < K; ++kk)
      act = act || (tail_weight(m_base[s_slot[kk] * M + block_base + tid], s_ts[kk], s_lam[kk]) != 0.0f);
  }
  const unsigned long long bal = __ballot(act);
  if (wave < RATO_BLOCK / RATO_WAVE && lane == 0) s_cnt[wave] = __popcll(bal);
  __syncthreads();
  int base = 0, n_un = 0;
#pragma unroll
  for (int i = 0; i < RATO_BLOCK / RATO_WAVE; ++i) {
    const int c = s_cnt[i];
    if (i < wave) base += c;
    n_un += c;
  }
  if (act) s_src[base + __popcll(bal & ((1ull << lane) - 1ull))] = tid;
  const double dt = P.dt64, kp = P.kp64, kd = P.kd64, drag = P.drag64;
  for (int c0 = 0; c0 < n_un; c0 += RATO_WAVE) {   // chunks of 64 union samples
    for (int i = tid; i < S; i += NT) TMASK[i] = 0u;
    __syncthreads();   // (s_src complete; the previous chunk's tables are dead)
    for (int idx = tid; idx < K * RATO_WAVE; idx += NT) {
      const int kk = idx >> 6, l = idx & 63;
      float w = 0.0f;
      int tr = 0;
      if (c0 + l < n_un) {
        const long m = block_base + s_src[c0 + l];
        w = tail_weight(m_base[s_slot[kk] * M + m], s_ts[kk], s_lam[kk]);
        const int a = arg_base[s_slot[kk] * M + m];
        const int r = a / S, t = a - r * S;
        tr = t | (r << 20);
        if (w != 0.0f) atomicOr(&TMASK[t], 1u << kk);
      }
      WT[idx] = w;
      TR[idx] = tr;
    }
    __syncthreads();
    const bool on = c0 + lane < n_un;
    const long m = on ? block_base + s_src[c0 + lane] : block_base;
    const double inv_m = 1.0 / (double)mass[m];
    const double a21 = -kp * dt * inv_m, dtm = dt * inv_m;
    if (wave == 0) {
      // ---- forward: the rollout of the chunk at u_k, once for all cuts
      const double cn = sqrt(dt) * P.beta64 * inv_m;
      const double c1 = dt * kd * inv_m, one_c1 = 1.0 - c1, c2 = dt * drag * inv_m, c22 = 2.0 * c2;
      int t_hi = 0;   // wave-uniform: the largest t* of any (cut, lane) of the chunk
      {
        int tm = 0;
        for (int kk = 0; kk < K; ++kk)
          if (WT[kk * RATO_WAVE + lane] != 0.0f) tm = max(tm, TR[kk * RATO_WAVE + lane] & 0xfffff);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) tm = max(tm, __shfl_xor(tm, off, RATO_WAVE));
        t_hi = __builtin_amdgcn_readfirstlane(tm);
      }
      double p[2] = {P.x_init64[0], P.x_init64[1]}, v[2] = {P.x_init64[3], P.x_init64[4]};
      constexpr int TB = 8;   // noise in batches of 8 steps, the next batch in flight while one is consumed
      auto load = [&](float (&xi)[TB][2], int tb) {
#pragma unroll
        for (int i = 0; i < TB; ++i) {
          const int t = (tb + i <= t_hi) ? tb + i : t_hi;
          xi[i][0] = dW[((size_t)t * 3 + 0) * ld + m];
          xi[i][1] = dW[((size_t)t * 3 + 1) * ld + m];
        }
      };
      auto steps = [&](const float (&xi)[TB][2], int tb) {
#pragma unroll
        for (int i = 0; i < TB; ++i) {
          const int t = tb + i;
          if (t <= t_hi) {
#pragma unroll
            for (int a = 0; a < 2; ++a) {
              const double av = fabs(v[a]);   // (the step of drone_rowmax_rollout_kernel, operation for operation)
              E[(t * 2 + a) * RATO_WAVE + lane] = (float)fma(c22, av, c1);
              const double tv = fma(a21, p[a], fma(dtm, uk[t * 3 + a], cn * (double)xi[i][a]));
              const double pn = fma(dt, v[a], p[a]);
              v[a] = fma(fma(-c2, av, one_c1), v[a], tv);
              p[a] = pn;
            }
            unsigned mask = __builtin_amdgcn_readfirstlane(TMASK[t]);
            while (mask) {   // the cuts with an arg-max row at this step: the position p_{t+1} the row is evaluated at
              const int kk = __builtin_ctz(mask);
              mask &= mask - 1u;
              if (on && WT[kk * RATO_WAVE + lane] != 0.0f && (TR[kk * RATO_WAVE + lane] & 0xfffff) == t) {
                GW[(kk * 2 + 0) * RATO_WAVE + lane] = p[0];
                GW[(kk * 2 + 1) * RATO_WAVE + lane] = p[1];
              }
            }
          }
        }
      };
      float xa[TB][2], xb[TB][2];
      load(xa, 0);
      for (int tb = 0; tb <= t_hi; tb += 2 * TB) {
        load(xb, tb + TB);
        steps(xa, tb);
        load(xa, tb + 2 * TB);
        steps(xb, tb + TB);
      }
    }
    __syncthreads();
    // ---- adjoint sweeps: one cut per wave.  mu_{t*+1} = e_0', mu_k = mu_{k+1} A_k; column k-1 of the row = W . (mu_k)[1] dt/m
    for (int kk = wave; kk < K; kk += TRU_NW) {
      const float wf = WT[kk * RATO_WAVE + lane];
      const bool in_tail = on && wf != 0.0f;
      const int ts = in_tail ? (TR[kk * RATO_WAVE + lane] & 0xfffff) : 0;
      const double w = in_tail ? (double)wf : 0.0;
      double gval = 0.0, w0 = 0.0, w1 = 0.0;
      if (in_tail) {   // g and grad_p g of the arg-max row at p_{t*+1}
        const int rs = TR[kk * RATO_WAVE + lane] >> 20;
        const double q00 = (double)Qsym[(size_t)(rs * 3 + 0) * ld + m], qss = (double)Qsym[(size_t)(rs * 3 + 1) * ld + m],
                     q11 = (double)Qsym[(size_t)(rs * 3 + 2) * ld + m];
        const double ox = rs == 0 ? P.obs_xy64[0][0] : (rs == 1 ? P.obs_xy64[1][0] : P.obs_xy64[2][0]);
        const double oy = rs == 0 ? P.obs_xy64[0][1] : (rs == 1 ? P.obs_xy64[1][1] : P.obs_xy64[2][1]);
        const double dx = GW[(kk * 2 + 0) * RATO_WAVE + lane] - ox, dy = GW[(kk * 2 + 1) * RATO_WAVE + lane] - oy;
        gval = 1.0 - (q00 * dx * dx + qss * dx * dy + q11 * dy * dy);
        w0 = w * -(2.0 * q00 * dx + qss * dy) * dtm;
        w1 = w * -(qss * dx + 2.0 * q11 * dy) * dtm;
      }
      int t_hi = 0;
      {
        int tm = ts;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) tm = max(tm, __shfl_xor(tm, off, RATO_WAVE));
        t_hi = __builtin_amdgcn_readfirstlane(tm);
      }
      double* __restrict__ ak = acc + (size_t)kk * nc;
      const bool cols_in_regs = S - 1 <= RATO_WAVE;   // lane j keeps column j of both axes: no LDS update inside the chain
      double colx = 0.0, coly = 0.0;
      double m0x = 0.0, m1x = 0.0, m0y = 0.0, m1y = 0.0;
      for (int k = t_hi; k >= 1; --k) {   // wave-uniform
        const bool in = in_tail && k <= ts;
        if (in && k == ts) {
          m0x = 1.0; m1x = 0.0; m0y = 1.0; m1y = 0.0;
        }
        double cx = 0.0, cy = 0.0;
        if (in) {
          const double ax = 1.0 - (double)E[(k * 2 + 0) * RATO_WAVE + lane], ay = 1.0 - (double)E[(k * 2 + 1) * RATO_WAVE + lane];
          const double n0x = m0x + m1x * a21, n1x = m0x * dt + m1x * ax;
          const double n0y = m0y + m1y * a21, n1y = m0y * dt + m1y * ay;
          m0x = n0x; m1x = n1x; m0y = n0y; m1y = n1y;
          cx = w0 * m1x;
          cy = w1 * m1y;
        }
        const double s0 = rato::wave_sum_dpp(cx);
        const double s1 = rato::wave_sum_dpp(cy);
        if (cols_in_regs) {
          colx += (lane == k - 1) ? s0 : 0.0;
          coly += (lane == k - 1) ? s1 : 0.0;
        } else if (lane == 0) {
          ak[(k - 1) * 2 + 0] += s0;
          ak[(k - 1) * 2 + 1] += s1;
        }
      }
      if (cols_in_regs && lane < S - 1) {
        ak[lane * 2 + 0] += colx;
        ak[lane * 2 + 1] += coly;
      }
      const double sg = rato::wave_sum_dpp(w * gval);
      if (lane == 0) ak[nw] += sg;
    }
  }
  __syncthreads();
  for (int i = tid; i < K * nc; i += NT) {
    const int kk = i / nc, j = i - kk * nc;
    part[((size_t)blockIdx.x * K_total + (k0 + kk)) * nc + j] = acc[i];
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Table-free form of the DRIVING oracle (R = 1: g_t = -(|e_{t+1} - q_{t+1}| - d_min), e = ego position, q = pedestrian
// position; driving.py:145-236, :260-313).  Everything about the ego is sample independent (its state carries no noise:
// driving.py:196-203 puts the diffusion on the pedestrian's velocity only), so every workgroup first folds, in fp64,
//   the ego trajectory at u_k                       v, phi:  v_{k+1} = v_k + dt u_{k,0},  phi_{k+1} = phi_k + dt u_{k,1}
//                                                   e_{k+1} = e_k + dt v_k (cos phi_k, sin phi_k)
//   TANGENT (rowmax): its derivative along x        de_{k+1} = de_k + dt (dv_k cos - v_k sin dphi_k, dv_k sin + v_k cos dphi_k)
//   !TANGENT (tail rows): the adjoint's table       C_k = -dt^2 (cos, -v sin | sin, v cos)(phi_k)
// into LDS (thread t sums the first t terms in the order of the recursion; S + 1 threads have work), then the samples
// re-run the pedestrian (4 states) in fp64 from their own inputs.
template <bool TANGENT>
__device__ __forceinline__ void car_ego64_tables(const rato_car_params& P, const double* __restrict__ uk,
                                                 const double* __restrict__ xs, double* U /* [S][2] (+ [S][2] xs) */,
                                                 double* TERM /* [S+1][4] */, double* EGO /* [S+1][2] */,
                                                 double* AUX /* TANGENT: dEGO [S+1][2]; else C [S][4] */) {
  const int S = P.S;
  const double dt = P.dt64;
  for (int i = threadIdx.x; i < 2 * S; i += blockDim.x) {
    U[i] = uk[i];
    if (TANGENT) U[2 * S + i] = xs[i];
  }
  __syncthreads();
  for (int t = threadIdx.x; t <= S; t += blockDim.x) {
    double v = P.ego_init64[2], ph = P.ego_init64[3], dv = 0.0, dph = 0.0;
    for (int k = 0; k < t; ++k) {
      v += dt * U[2 * k + 0];
      ph += dt * U[2 * k + 1];
      if (TANGENT) {
        dv += dt * U[2 * S + 2 * k + 0];
        dph += dt * U[2 * S + 2 * k + 1];
      }
    }
    double sn, cs;
    sincos(ph, &sn, &cs);
    TERM[t * 4 + 0] = dt * v * cs;
    TERM[t * 4 + 1] = dt * v * sn;
    if (TANGENT) {
      TERM[t * 4 + 2] = dt * (dv * cs - v * sn * dph);
      TERM[t * 4 + 3] = dt * (dv * sn + v * cs * dph);
    } else if (t < S) {
      AUX[t * 4 + 0] = -dt * dt * cs;
      AUX[t * 4 + 1] = dt * dt * v * sn;
      AUX[t * 4 + 2] = -dt * dt * sn;
      AUX[t * 4 + 3] = -dt * dt * v * cs;
    }
  }
  __syncthreads();
  for (int t = threadIdx.x; t <= S; t += blockDim.x) {
    double x = P.ego_init64[0], y = P.ego_init64[1], dx = 0.0, dy = 0.0;
    for (int k = 0; k < t; ++k) {
      x += TERM[k * 4 + 0];
      y += TERM[k * 4 + 1];
      if (TANGENT) {
        dx += TERM[k * 4 + 2];
        dy += TERM[k * 4 + 3];
      }
    }
    EGO[t * 2 + 0] = x;
    EGO[t * 2 + 1] = y;
    if (TANGENT) {
      AUX[t * 2 + 0] = dx;
      AUX[t * 2 + 1] = dy;
    }
  }
  __syncthreads();
}

__host__ __device__ inline size_t car_rollout_ego_doubles(int S) {   // U (uk | xs) + TERM + EGO + AUX
  return (size_t)4 * S + (size_t)(S + 1) * 4 + (size_t)(S + 1) * 2 + (size_t)(S + 1) * 4;
}

// m_i(u) = max_t [ g_t + grad g_t . x ],  x = u - u_k: the pedestrian and its tangent along x, one sample per thread.
//   d = e_t - q_t, n = d / |d|, H = (I - n n') / |d|
//   F = -w_r n + w_s (v_des - qv_y) (1, 1)                       dF = -w_r H (de_t - dq_t) - w_s dqv_y (1, 1)
//   q_{t+1} = q_t + dt qv_t,  qv_{t+1} = qv_t + dt F + sqrt(dt) beta dW_t          (same recursion for dq, dqv with dF)
//   row t = g_t - n_{t+1} . (de_{t+1} - dq_{t+1})
template <bool BYVAL>
__global__ __launch_bounds__(RATO_BLOCK) void car_rowmax_rollout_kernel(
    rato_car_params P, const double* __restrict__ uk, const float* __restrict__ dW, const float* __restrict__ x0_ped,
    const float* __restrict__ w_speed, const float* __restrict__ w_rep, const double* __restrict__ xs_mem,
    const XArg xv, float* __restrict__ m_out, int* __restrict__ arg_out) {
  const double* xs = BYVAL ? xv.v : xs_mem;
  extern __shared__ __attribute__((aligned(16))) unsigned char crr_lds[];
  const int S = P.S;
  double* U = reinterpret_cast<double*>(crr_lds);
  double* TERM = U + 4 * S;
  double* EGO = TERM + (size_t)(S + 1) * 4;
  double* DEGO = EGO + (size_t)(S + 1) * 2;
  car_ego64_tables<true>(P, uk, xs, U, TERM, EGO, DEGO);
  const size_t M = (size_t)P.M;
  const size_t m = (size_t)blockIdx.x * RATO_BLOCK + threadIdx.x;
  if (m >= M) return;
  const double dt = P.dt64, w_s = (double)w_speed[m], w_r = (double)w_rep[m], cn = sqrt(dt) * P.beta64;
  double px = (double)x0_ped[0 * M + m], py = (double)x0_ped[1 * M + m], vx = (double)x0_ped[2 * M + m],
         vy = (double)x0_ped[3 * M + m];
  double dqx = 0.0, dqy = 0.0, dvx = 0.0, dvy = 0.0;
  double best = -INFINITY;
  int best_idx = 0;
  bool bad = false;
  // geometry at the current state (carried: the normal of row t - 1 is the normal of step t)
  double dx = EGO[0] - px, dy = EGO[1] - py;
  double rinv = 1.0 / sqrt(dx * dx + dy * dy);
  constexpr int TB = 8;   // noise: batches of 8 steps, the next batch in flight while one is consumed (see the drone kernel)
  auto load = [&](float (&xi)[TB][2], int t0) {
#pragma unroll
    for (int i = 0; i < TB; ++i) {
      const int t = (t0 + i < S) ? t0 + i : S - 1;
      xi[i][0] = dW[(size_t)(t * 2 + 0) * M + m];
      xi[i][1] = dW[(size_t)(t * 2 + 1) * M + m];
    }
  };
  auto steps = [&](const float (&xi)[TB][2], int t0) {
#pragma unroll
    for (int i = 0; i < TB; ++i) {
      const int t = t0 + i;
      if (t < S) {
        const double n0 = dx * rinv, n1 = dy * rinv;
        const double ddx = DEGO[t * 2 + 0] - dqx, ddy = DEGO[t * 2 + 1] - dqy;
        const double nd = n0 * ddx + n1 * ddy;
        const double hx = (ddx - n0 * nd) * rinv, hy = (ddy - n1 * nd) * rinv;   // H (de - dq)
        const double dcommon = -w_s * dvy;
        const double dF0 = -w_r * hx + dcommon, dF1 = -w_r * hy + dcommon;
        const double common = w_s * (P.speed_ped_des64 - vy);
        const double F0 = -w_r * n0 + common, F1 = -w_r * n1 + common;
        const double ndqx = dqx + dt * dvx, ndqy = dqy + dt * dvy;
        dvx += dt * dF0;
        dvy += dt * dF1;
        dqx = ndqx;
        dqy = ndqy;
        const double npx = px + dt * vx, npy = py + dt * vy;
        vx += dt * F0 + cn * (double)xi[i][0];
        vy += dt * F1 + cn * (double)xi[i][1];
        px = npx;
        py = npy;
        dx = EGO[(t + 1) * 2 + 0] - px;
        dy = EGO[(t + 1) * 2 + 1] - py;
        const double r2 = dx * dx + dy * dy;
        rinv = 1.0 / sqrt(r2);
        const double g = -(r2 * rinv - P.d_min64);
        const double val = g - rinv * (dx * (DEGO[(t + 1) * 2 + 0] - dqx) + dy * (DEGO[(t + 1) * 2 + 1] - dqy));
        bad = bad || (val != val);
        if (val > best) {   // ascending t: the smallest row index among equal values
          best = val;
          best_idx = t;
        }
      }
    }
  };
  float xa[TB][2], xb[TB][2];
  load(xa, 0);
  for (int t0 = 0; t0 < S; t0 += 2 * TB) {
    load(xb, t0 + TB);
    steps(xa, t0);
    load(xa, t0 + 2 * TB);
    steps(xb, t0 + TB);
  }
  // a NaN row loses every comparison: it is reported, not dropped (the facades check the statistics of m; there is
  // no stored linearization to scan in this form)
  m_out[m] = bad ? __builtin_nanf("") : (float)best;
  arg_out[m] = best_idx;
}

// The cut of the rollout form.  The tail samples of the block (compacted, walked by wave 0 in chunks of 64) re-run the
// pedestrian in fp64 up to their own t*, leaving K_k = dt w_r H_k (3 numbers; ~1e-3, a correction to the identity, kept
// as floats: 1e-10 of the step Jacobian) in LDS, pick up g and n of their arg-max row on the way, and run the 8-state
// adjoint from t* down in the reduced form of car_linearize_rows_kernel (eta_e = -eta_q; E = dt (eta_v, eta_phi) IS the
// Jacobian entry):      E += q . C_k;   qv' = qv + dt q - dt w_s (qv_x + qv_y) e_y;   q += qv K_k;   column k - 1 = E.
__global__ __launch_bounds__(RATO_BLOCK) void car_tail_rows_rollout_kernel(
    rato_car_params P, const double* __restrict__ uk, const float* __restrict__ dW, const float* __restrict__ x0_ped,
    const float* __restrict__ w_speed, const float* __restrict__ w_rep, const float* __restrict__ m_base,
    const int* __restrict__ arg_base, const double* __restrict__ stats_base, long stats_stride,
    const int* __restrict__ slots, double alphaM, double* __restrict__ part) {
  extern __shared__ __attribute__((aligned(16))) unsigned char ctr_lds[];
  const int S = P.S;
  const long M = P.M;
  const int nw = 2 * (S - 1), nc = nw + 1;
  double* U = reinterpret_cast<double*>(ctr_lds);
  double* TERM = U + 4 * S;
  double* EGO = TERM + (size_t)(S + 1) * 4;
  double* C = EGO + (size_t)(S + 1) * 2;
  double* acc = C + (size_t)(S + 1) * 4;                              // [nc] column sums of the block
  float* KT = reinterpret_cast<float*>(acc + nc);                     // [S][3][64] K of the current chunk
  car_ego64_tables<false>(P, uk, nullptr, U, TERM, EGO, C);
  const int K = gridDim.y, kk = blockIdx.y;
  const long slot = slots ? slots[kk] : 0;
  const float* __restrict__ mvals = m_base + slot * M;
  const int* __restrict__ arg = arg_base + slot * M;
  float tstar, lambda;
  tail_rule(stats_base + slot * stats_stride, alphaM, tstar, lambda);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float w0f = 0.0f;
  int t0 = 0;
  {
    const long m0 = (long)blockIdx.x * RATO_BLOCK + threadIdx.x;
    if (m0 < M) {
      w0f = tail_weight(mvals[m0], tstar, lambda);
      t0 = arg[m0];
    }
  }
  TailLane unused;
  TailLists lists;
  const int n_tail = compact_tail(w0f, t0, 0, (long)blockIdx.x * RATO_BLOCK, M, unused, &lists);
  for (int i = threadIdx.x; i < nc; i += RATO_BLOCK) acc[i] = 0.0;
  __syncthreads();
  if (wave == 0) {
    const double dt = P.dt64, cn = sqrt(dt) * P.beta64;
    for (int c0 = 0; c0 < n_tail; c0 += RATO_WAVE) {   // chunks of 64 tail samples, in sample order
      const bool on = c0 + lane < n_tail;
      const size_t m = on ? (size_t)((long)blockIdx.x * RATO_BLOCK + lists.src[c0 + lane]) : 0;
      const double w = on ? (double)lists.w[c0 + lane] : 0.0;
      const int ts = on ? (lists.tr[c0 + lane] & 0xfffff) : 0;
      const double w_s = (double)w_speed[m], w_r = (double)w_rep[m], ks = dt * w_s;
      double px = (double)x0_ped[0 * (size_t)M + m], py = (double)x0_ped[1 * (size_t)M + m],
             vx = (double)x0_ped[2 * (size_t)M + m], vy = (double)x0_ped[3 * (size_t)M + m];
      double n0s = 0.0, n1s = 0.0, gval = 0.0;
      int t_hi = 0;   // wave-uniform: the forward pass only has to reach the largest t* of the chunk
      {
        int tm = ts;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) tm = max(tm, __shfl_xor(tm, off, RATO_WAVE));
        t_hi = __builtin_amdgcn_readfirstlane(tm);
      }
      double dx = EGO[0] - px, dy = EGO[1] - py;
      double rinv = 1.0 / sqrt(dx * dx + dy * dy);
      constexpr int TB = 8;   // noise in batches of 8 steps, the next batch in flight while one is consumed
      auto load = [&](float (&xi)[TB][2], int tb) {
#pragma unroll
        for (int i = 0; i < TB; ++i) {
          const int t = (tb + i <= t_hi) ? tb + i : t_hi;
          xi[i][0] = dW[(size_t)(t * 2 + 0) * (size_t)M + m];
          xi[i][1] = dW[(size_t)(t * 2 + 1) * (size_t)M + m];
        }
      };
      auto steps = [&](const float (&xi)[TB][2], int tb) {
#pragma unroll
        for (int i = 0; i < TB; ++i) {
          const int t = tb + i;
          if (t <= t_hi) {
            const double n0 = dx * rinv, n1 = dy * rinv;
            const double kr = dt * w_r * rinv;
            KT[(t * 3 + 0) * RATO_WAVE + lane] = (float)(kr * (1.0 - n0 * n0));
            KT[(t * 3 + 1) * RATO_WAVE + lane] = (float)(-kr * n0 * n1);
            KT[(t * 3 + 2) * RATO_WAVE + lane] = (float)(kr * (1.0 - n1 * n1));
            const double common = w_s * (P.speed_ped_des64 - vy);
            const double F0 = -w_r * n0 + common, F1 = -w_r * n1 + common;
            const double npx = px + dt * vx, npy = py + dt * vy;
            vx += dt * F0 + cn * (double)xi[i][0];
            vy += dt * F1 + cn * (double)xi[i][1];
            px = npx;
            py = npy;
            dx = EGO[(t + 1) * 2 + 0] - px;
            dy = EGO[(t + 1) * 2 + 1] - py;
            const double r2 = dx * dx + dy * dy;
            rinv = 1.0 / sqrt(r2);
            if (on && t == ts) {   // the arg-max row of this sample: g and the normal at t* + 1
              gval = -(r2 * rinv - P.d_min64);
              n0s = dx * rinv;
              n1s = dy * rinv;
            }
          }
        }
      };
      {
        float xa[TB][2], xb[TB][2];
        load(xa, 0);
        for (int tb = 0; tb <= t_hi; tb += 2 * TB) {
          load(xb, tb + TB);
          steps(xa, tb);
          load(xa, tb + 2 * TB);
          steps(xb, tb + TB);
        }
      }
      double qx = 0.0, qy = 0.0, qvx = 0.0, qvy = 0.0, Ex = 0.0, Ey = 0.0;
      for (int k = t_hi; k >= 1; --k) {   // wave-uniform
        const bool in = on && k <= ts;
        if (in && k == ts) {
          qx = n0s; qy = n1s; qvx = 0.0; qvy = 0.0; Ex = 0.0; Ey = 0.0;
        }
        double cx = 0.0, cy = 0.0;
        if (in) {
          const double k00 = (double)KT[(k * 3 + 0) * RATO_WAVE + lane], k01 = (double)KT[(k * 3 + 1) * RATO_WAVE + lane],
                       k11 = (double)KT[(k * 3 + 2) * RATO_WAVE + lane];
          const double fx = qvx * k00 + qvy * k01, fy = qvx * k01 + qvy * k11;
          Ex += qx * C[k * 4 + 0] + qy * C[k * 4 + 2];   // uses eta_q BEFORE its update
          Ey += qx * C[k * 4 + 1] + qy * C[k * 4 + 3];
          const double nqvx = dt * qx + qvx, nqvy = dt * qy + qvy - ks * (qvx + qvy);
          qx += fx;
          qy += fy;
          qvx = nqvx;
          qvy = nqvy;
          cx = w * Ex;
          cy = w * Ey;
        }
        const double s0 = rato::wave_sum_dpp(cx);
        const double s1 = rato::wave_sum_dpp(cy);
        if (lane == 0) {
          acc[(k - 1) * 2 + 0] += s0;
          acc[(k - 1) * 2 + 1] += s1;
        }
      }
      const double sg = rato::wave_sum_dpp(w * gval);
      if (lane == 0) acc[nw] += sg;
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < nc; i += RATO_BLOCK) part[((size_t)blockIdx.x * K + kk) * nc + i] = acc[i];
}


// ---- sums of the matrix-free KKT certificate (rato_kkt_sums) ------------------------------------------------------
// The reduced solution (u*, slack) of a subproblem, lifted to the reference's QP (y_i = max(-slack, m_i(u*) - t)), is
// certified against that QP without forming it: every residual is a sum over the samples of quantities the oracle holds
// -- the m values at u* and, per cut k with a multiplier, its tail weighting w_k (from the m values and the statistics
// record of the ring slot it was generated in).  With v = t - slack, per block and in fp64:
//   part[blk][k]         = sum_i w_ki (m_i* - v)^+        k < K     (obstacle-row and y-row complementarity)
//   part[blk][K + k]     = sum_i w_ki                               (= alpha M: the t- and slack-stationarity)
//   part[blk][2K]        = sum_i (m_i* - v)^+                       (the CVaR row: sum_i y_i = -slack M + this)
//   part[blk][2K + 1]    = max_i sum_k lam_k w_ki                   (<= sum_k lam_k: sign of the y-row multipliers)
__global__ __launch_bounds__(RATO_BLOCK) void kkt_sums_kernel(const float* __restrict__ m_star, long M,
                                                              const float* __restrict__ m_base,
                                                              const double* __restrict__ stats_base, long stats_stride,
                                                              const int* __restrict__ slots,
                                                              const double* __restrict__ lam, int K, double alphaM,
                                                              double v, double* __restrict__ part) {
  extern __shared__ double ks_lds[];   // [4 waves][2K + 2]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long i = (long)blockIdx.x * RATO_BLOCK + threadIdx.x;
  const bool valid = i < M;
  const float ms = valid ? m_star[i] : 0.0f;
  const double ex = valid ? fmax((double)ms - v, 0.0) : 0.0;
  const int nc = 2 * K + 2;
  double* mine = ks_lds + (size_t)wave * nc;
  double lw = 0.0;
  for (int k = 0; k < K; ++k) {
    const long slot = slots[k];
    float tstar, lambda;
    tail_rule(stats_base + slot * stats_stride, alphaM, tstar, lambda);
    const double w = valid ? (double)tail_weight(m_base[slot * M + i], tstar, lambda) : 0.0;
    lw += lam[k] * w;
    const double a = rato::wave_sum_dpp(w * ex), b = rato::wave_sum_dpp(w);
    if (lane == 63) {
      mine[k] = a;
      mine[K + k] = b;
    }
  }
  const double e = rato::wave_sum_dpp(ex);
  double mx = lw;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) mx = fmax(mx, __shfl_xor(mx, off, RATO_WAVE));
  if (lane == 63) {
    mine[2 * K] = e;
    mine[2 * K + 1] = mx;
  }
  __syncthreads();
  for (int c = threadIdx.x; c < nc; c += RATO_BLOCK) {
    double r;
    if (c == nc - 1) {
      r = fmax(fmax(ks_lds[c], ks_lds[nc + c]), fmax(ks_lds[2 * nc + c], ks_lds[3 * nc + c]));
    } else {
      r = (ks_lds[c] + ks_lds[nc + c]) + (ks_lds[2 * nc + c] + ks_lds[3 * nc + c]);
    }
    part[(size_t)blockIdx.x * nc + c] = r;
  }
}

}  // namespace

extern "C" int rato_drone_rowmax_implicit(const rato_drone_params* p, const float* mass, const float* A22,
                                          int32_t a22_axes, const float* W, const float* base, double sign,
                                          const double* xs, float* m_out, int32_t* arg_out, void* stream) {
  RATO_CLEAR_ERROR();
  if (!p || p->M <= 0 || p->S < 1 || p->ld < p->M || !(p->dt > 0.0f) || !(p->dt64 > 0.0) || !mass || !A22 || !W || !base || !xs ||
      !m_out || !arg_out || (a22_axes != 2 && a22_axes != 3) || (sign != 1.0 && sign != -1.0))
    return RATO_EINVAL;
  dim3 grid((unsigned)rato::nblocks_for(p->M)), block(RATO_BLOCK);
  hipLaunchKernelGGL(drone_rowmax_implicit_kernel, grid, block, 0, rato::as_stream(stream), *p, mass, A22, a22_axes, W,
                     base, sign, xs, m_out, arg_out);
  RATO_LAUNCH_CHECK();
  return RATO_OK;
}

extern "C" int rato_saa_rowmax(const float* G, const float* W, int32_t tile, int32_t R, int32_t S, int64_t M,
                               int64_t ld, const float* base, double sign, const double* xs, int32_t n_u,
                               float* m_out, int32_t* arg_out, void* stream) {
  RATO_CLEAR_ERROR();
  if (!G || !base || !xs || !m_out || !arg_out || M <= 0 || S < 1 || ld < M || n_u < 2 || (tile != 64 && tile != 256) ||
      (R != 1 && R != 3) || (sign != 1.0 && sign != -1.0))
    return RATO_EINVAL;
  const size_t lds = (size_t)S * 16 + RM_NW * 64 * 12 + 16;
  if (lds > 64 * 1024) return RATO_EINVAL;
  dim3 grid((unsigned)((M + 63) / 64)), block(RM_NW * RATO_WAVE);
  hipStream_t st = rato::as_stream(stream);
  if (R == 3 && W)
    hipLaunchKernelGGL((rowmax_kernel<3, true>), grid, block, lds, st, G, W, tile, S, (long)M, (long)ld, base, sign, xs,
                       n_u, m_out, arg_out);
  else if (R == 3)
    hipLaunchKernelGGL((rowmax_kernel<3, false>), grid, block, lds, st, G, W, tile, S, (long)M, (long)ld, base, sign,
                       xs, n_u, m_out, arg_out);
  else if (!W)
    hipLaunchKernelGGL((rowmax_kernel<1, false>), grid, block, lds, st, G, W, tile, S, (long)M, (long)ld, base, sign,
                       xs, n_u, m_out, arg_out);
  else
    return RATO_EINVAL;
  RATO_LAUNCH_CHECK();
  return RATO_OK;
}

extern "C" int rato_drone_tail_rows_implicit(const rato_drone_params* p, const float* mass, const float* A22,
                                             int32_t a22_axes, const float* W, const float* base,
                                             const float* m_base, const int32_t* arg_base, const double* stats_base,
                                             int64_t stats_stride, const int32_t* slots, int32_t K, double alphaM,
                                             double* part, void* stream) {
  RATO_CLEAR_ERROR();
  if (!p || p->M <= 0 || p->S < 2 || p->ld < p->M || !(p->dt > 0.0f) || !(p->dt64 > 0.0) || !mass || !A22 || !W || !base || !m_base ||
      !arg_base || !stats_base || !part || K < 1 || K > 65535 || (!slots && K != 1) || stats_stride < 11 ||
      (a22_axes != 2 && a22_axes != 3))
    return RATO_EINVAL;
  const size_t lds = (size_t)(RATO_BLOCK / 64) * (2 * (p->S - 1) + 1) * sizeof(double);
  dim3 grid((unsigned)rato::nblocks_for(p->M), (unsigned)K), block(RATO_BLOCK);
  hipLaunchKernelGGL(drone_tail_rows_implicit_kernel, grid, block, lds, rato::as_stream(stream), *p, mass, A22,
                     a22_axes, W, base, m_base, arg_base, stats_base, (long)stats_stride, slots, alphaM, part);
  RATO_LAUNCH_CHECK();
  return RATO_OK;
}

extern "C" int rato_saa_tail_rows_batch(const float* G, const float* W, int64_t ld, int32_t tile, int32_t R, int32_t S,
                                        int64_t M, const float* base, const float* m_base, const int32_t* arg_base,
                                        const double* stats_base, int64_t stats_stride, const int32_t* slots,
                                        int32_t K, double alphaM, double* part, void* stream) {
  RATO_CLEAR_ERROR();
  if (!G || !base || !m_base || !arg_base || !stats_base || !part || M <= 0 || S < 2 || K < 1 || K > 65535 ||
      (!slots && K != 1) || ld < M || stats_stride < 11 || (tile != 64 && tile != 256) || (R != 1 && R != 3))
    return RATO_EINVAL;
  const size_t lds = (size_t)(RATO_BLOCK / 64) * (2 * (S - 1) + 1) * sizeof(double);
  dim3 grid((unsigned)rato::nblocks_for((int32_t)M), (unsigned)K), block(RATO_BLOCK);
  hipStream_t st = rato::as_stream(stream);
  if (R == 3)
    hipLaunchKernelGGL(tail_rows_batch_kernel<3>, grid, block, lds, st, G, W, (long)ld, tile, S, (long)M, base, m_base,
                       arg_base, stats_base, (long)stats_stride, slots, alphaM, part);
  else
    hipLaunchKernelGGL(tail_rows_batch_kernel<1>, grid, block, lds, st, G, W, (long)ld, tile, S, (long)M, base, m_base,
                       arg_base, stats_base, (long)stats_stride, slots, alphaM, part);
  RATO_LAUNCH_CHECK();
  return RATO_OK;
}

namespace {
// xs_host != NULL and S n_u <= XARG_MAX: x is passed by value (no device copy of it is read)
int drone_rowmax_rollout_launch(const rato_drone_params* p, const double* uk, const float* dW, const float* mass,
                                const float* Qsym, const double* xs, const double* xs_host, float* m_out,
                                int32_t* arg_out, void* stream) {
  RATO_CLEAR_ERROR();
  if (!p || p->M <= 0 || p->S < 1 || p->ld < p->M || !(p->dt > 0.0f) || !(p->dt64 > 0.0) || !uk || !dW || !mass || !Qsym ||
      (!xs && !xs_host) || !m_out || !arg_out)
    return RATO_EINVAL;
  dim3 grid((unsigned)rato::nblocks_for(p->M)), block(RATO_BLOCK);
  XArg xv;
  if (xs_host && p->S * 3 <= XARG_MAX) {
    for (int i = 0; i < p->S * 3; ++i) xv.v[i] = xs_host[i];
    hipLaunchKernelGGL(drone_rowmax_rollout_kernel<true>, grid, block, 0, rato::as_stream(stream), *p, uk, dW, mass, Qsym,
                       xs, xv, m_out, arg_out);
  } else {
    if (!xs) return RATO_EINVAL;
    hipLaunchKernelGGL(drone_rowmax_rollout_kernel<false>, grid, block, 0, rato::as_stream(stream), *p, uk, dW, mass, Qsym,
                       xs, xv, m_out, arg_out);
  }
  RATO_LAUNCH_CHECK();
  return RATO_OK;
}
}  // namespace

extern "C" int rato_drone_rowmax_rollout(const rato_drone_params* p, const double* uk, const float* dW,
                                         const float* mass, const float* Qsym, const double* xs, float* m_out,
                                         int32_t* arg_out, void* stream) {
  if (!xs) return RATO_EINVAL;
  return drone_rowmax_rollout_launch(p, uk, dW, mass, Qsym, xs, nullptr, m_out, arg_out, stream);
}

extern "C" int rato_drone_tail_rows_rollout(const rato_drone_params* p, const double* uk, const float* dW,
                                            const float* mass, const float* Qsym, const float* m_base,
                                            const int32_t* arg_base, const double* stats_base, int64_t stats_stride,
                                            const int32_t* slots, int32_t K, double alphaM, double* part,
                                            void* stream) {
  RATO_CLEAR_ERROR();
  if (!p || p->M <= 0 || p->S < 2 || p->S >= (1 << 20) || p->ld < p->M || !(p->dt > 0.0f) || !(p->dt64 > 0.0) || !uk || !dW || !mass ||
      !Qsym || !m_base || !arg_base || !stats_base || !part || K < 1 || K > 65535 || (!slots && K != 1) ||
      stats_stride < 11)
    return RATO_EINVAL;
  size_t lds = (size_t)(2 * (p->S - 1) + 1) * sizeof(double) + (size_t)p->S * 2 * RATO_WAVE * sizeof(float);
  if (lds + 4096 > 160 * 1024) return RATO_EINVAL;   // S <= 300 (4 KB: the static lists of the tail compaction)
  // the sweep's per-lane terms in LDS (summed after the sweep) while two workgroups still fit a CU; RATO_TAIL_CTAB=0: the
  // wave-wide sum inside every step of the sweep
  static const int ctab_env = [] { const char* e = getenv("RATO_TAIL_CTAB"); return e ? atoi(e) : 1; }();
  const size_t lds_c = lds + tail_ctab_doubles(p->S) * sizeof(double);
  const int c_tab = (ctab_env && lds_c + 4096 <= 80 * 1024) ? 1 : 0;
  if (c_tab) lds = lds_c;
  static rato::DynamicLdsLimit lds_limit;
  {
    const hipError_t e = lds_limit.ensure(lds, [](size_t bytes) {
      return hipFuncSetAttribute(reinterpret_cast<const void*>(drone_tail_rows_rollout_kernel),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    });
    if (e != hipSuccess) return RATO_EHIP - (int)e;
  }
  // several cuts (the kept cuts of a subproblem): the union form, up to TRU_KMAX cuts per launch (A/B: RATO_TAIL_UNION=0)
  static const int union_env = [] { const char* e = getenv("RATO_TAIL_UNION"); return e ? atoi(e) : 1; }();
  if (K > 1 && union_env) {
    const size_t lds_u = tail_union_lds_bytes(p->S, K < TRU_KMAX ? K : TRU_KMAX);
    if (lds_u + 4096 <= 160 * 1024) {
      static rato::DynamicLdsLimit lds_limit_u;
      const hipError_t e = lds_limit_u.ensure(lds_u, [](size_t bytes) {
        return hipFuncSetAttribute(reinterpret_cast<const void*>(drone_tail_rows_rollout_union_kernel),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
      });
      if (e != hipSuccess) return RATO_EHIP - (int)e;
      for (int k0 = 0; k0 < K; k0 += TRU_KMAX) {
        const int kn = (K - k0) < TRU_KMAX ? (K - k0) : TRU_KMAX;
        hipLaunchKernelGGL(drone_tail_rows_rollout_union_kernel, dim3((unsigned)rato::nblocks_for(p->M)),
                           dim3(TRU_NW * RATO_WAVE), tail_union_lds_bytes(p->S, kn), rato::as_stream(stream), *p, uk, dW,
                           mass, Qsym, m_base, arg_base, stats_base, (long)stats_stride, slots, k0, kn, K, alphaM, part);
        RATO_LAUNCH_CHECK();
      }
      return RATO_OK;
    }
  }
  dim3 grid((unsigned)rato::nblocks_for(p->M), (unsigned)K), block(RATO_BLOCK);
  hipLaunchKernelGGL(drone_tail_rows_rollout_kernel, grid, block, lds, rato::as_stream(stream), *p, uk, dW, mass, Qsym,
                     m_base, arg_base, stats_base, (long)stats_stride, slots, alphaM, part, c_tab);
  RATO_LAUNCH_CHECK();
  return RATO_OK;
}


namespace {
bool car_params64_ok(const rato_car_params* p) {
  return p && p->M > 0 && p->S >= 1 && p->S <= 1024 && p->dt64 > 0.0 && p->d_min64 >= 0.0;
}
}  // namespace

namespace {
int car_rowmax_rollout_launch(const rato_car_params* p, const double* uk, const float* dW, const float* x0_ped,
                              const float* w_speed, const float* w_rep, const double* xs, const double* xs_host,
                              float* m_out, int32_t* arg_out, void* stream) {
  RATO_CLEAR_ERROR();
  if (!car_params64_ok(p) || !uk || !dW || !x0_ped || !w_speed || !w_rep || (!xs && !xs_host) || !m_out || !arg_out)
    return RATO_EINVAL;
  const size_t lds = car_rollout_ego_doubles(p->S) * sizeof(double);
  if (lds > 64 * 1024) return RATO_EINVAL;
  dim3 grid((unsigned)rato::nblocks_for(p->M)), block(RATO_BLOCK);
  XArg xv;
  if (xs_host && p->S * 2 <= XARG_MAX) {
    for (int i = 0; i < p->S * 2; ++i) xv.v[i] = xs_host[i];
    hipLaunchKernelGGL(car_rowmax_rollout_kernel<true>, grid, block, lds, rato::as_stream(stream), *p, uk, dW, x0_ped,
                       w_speed, w_rep, xs, xv, m_out, arg_out);
  } else {
    if (!xs) return RATO_EINVAL;
    hipLaunchKernelGGL(car_rowmax_rollout_kernel<false>, grid, block, lds, rato::as_stream(stream), *p, uk, dW, x0_ped,
                       w_speed, w_rep, xs, xv, m_out, arg_out);
  }
  RATO_LAUNCH_CHECK();
  return RATO_OK;
}
}  // namespace

extern "C" int rato_car_rowmax_rollout(const rato_car_params* p, const double* uk, const float* dW,
                                       const float* x0_ped, const float* w_speed, const float* w_rep,
                                       const double* xs, float* m_out, int32_t* arg_out, void* stream) {
  if (!xs) return RATO_EINVAL;
  return car_rowmax_rollout_launch(p, uk, dW, x0_ped, w_speed, w_rep, xs, nullptr, m_out, arg_out, stream);
}

extern "C" int rato_car_tail_rows_rollout(const rato_car_params* p, const double* uk, const float* dW,
                                          const float* x0_ped, const float* w_speed, const float* w_rep,
                                          const float* m_base, const int32_t* arg_base, const double* stats_base,
                                          int64_t stats_stride, const int32_t* slots, int32_t K, double alphaM,
                                          double* part, void* stream) {
  RATO_CLEAR_ERROR();
  if (!car_params64_ok(p) || p->S < 2 || !uk || !dW || !x0_ped || !w_speed || !w_rep || !m_base || !arg_base ||
      !stats_base || !part || K < 1 || K > 65535 || (!slots && K != 1) || stats_stride < 11)
    return RATO_EINVAL;
  const size_t lds = (car_rollout_ego_doubles(p->S) + (size_t)(2 * (p->S - 1) + 1)) * sizeof(double) +
                     (size_t)p->S * 3 * RATO_WAVE * sizeof(float);
  static rato::DynamicLdsLimit limit;
  if (lds + 4096 > 160 * 1024) return RATO_EINVAL;
  {
    const hipError_t e = limit.ensure(lds + 4096, [](size_t bytes) {
      return hipFuncSetAttribute(reinterpret_cast<const void*>(car_tail_rows_rollout_kernel),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    });
    if (e != hipSuccess) return RATO_EHIP - (int)e;
  }
  dim3 grid((unsigned)rato::nblocks_for(p->M), (unsigned)K), block(RATO_BLOCK);
  hipLaunchKernelGGL(car_tail_rows_rollout_kernel, grid, block, lds, rato::as_stream(stream), *p, uk, dW, x0_ped,
                     w_speed, w_rep, m_base, arg_base, stats_base, (long)stats_stride, slots, alphaM, part);
  RATO_LAUNCH_CHECK();
  return RATO_OK;
}

// One oracle round trip of the cutting-plane loop in ONE call: upload x = u - u_k, m(u) by the table-free rowmax, exact
// tail selection, the cut's sums, read-back, stream synchronisation.  The same five stream-ordered steps the Python
// facade issued one by one (cvar_cuts.CvarCutSolver.evaluate: two copies, four library calls, one synchronize -- about as
// much host time per cut as the ~100 us the device works on it at M = 1e5).
extern "C" int rato_cut_oracle_rollout(int32_t system, const void* params, const double* uk, const float* s0,
                                       const float* s1, const float* s2, const float* s3, const double* x_host,
                                       double* x_dev, float* m_out, int32_t* arg_out, double alpha, float thr,
                                       double alphaM, void* workspace, size_t workspace_bytes, double* res_dev,
                                       double* part_dev, double* res_host, void* stream) {
  RATO_CLEAR_ERROR();
  if ((system != 0 && system != 1) || !params || !x_host || !x_dev || !res_dev || !res_host) return RATO_EINVAL;
  hipStream_t st = rato::as_stream(stream);
  int S, n_u;
  int64_t M;
  if (system == 0) {
    const rato_drone_params* p = static_cast<const rato_drone_params*>(params);
    S = p->S, n_u = 3, M = p->M;
  } else {
    const rato_car_params* p = static_cast<const rato_car_params*>(params);
    S = p->S, n_u = 2, M = p->M;
  }
  if (S < 1 || M < 1) return RATO_EINVAL;
  const int nc = 2 * (S - 1) + 1;
  hipError_t e = hipSuccess;
  // read-back: the last launch writes the record into res_host (pinned) and the host watches it arrive (rato_common.h)
  const int n_words = RATO_N_STATS + (S > 1 ? nc : 0);
  if (rato::readback_poll_enabled()) rato::readback_arm(res_host, n_words);
  if (S * n_u > XARG_MAX) {   // long horizons: x goes through device memory; otherwise it rides in the kernel arguments
    e = hipMemcpyAsync(x_dev, x_host, sizeof(double) * (size_t)S * n_u, hipMemcpyHostToDevice, st);
    if (e != hipSuccess) return RATO_EHIP - (int)e;
  }
  int rc = system == 0
               ? drone_rowmax_rollout_launch(static_cast<const rato_drone_params*>(params), uk, s0, s1, s2, x_dev, x_host,
                                             m_out, arg_out, stream)
               : car_rowmax_rollout_launch(static_cast<const rato_car_params*>(params), uk, s0, s1, s2, s3, x_dev, x_host,
                                           m_out, arg_out, stream);
  if (rc != RATO_OK) return rc;
  rc = rato_risk_stats(m_out, M, alpha, thr, workspace, workspace_bytes, res_dev, stream);
  if (rc != RATO_OK) return rc;
  if (S > 1) {
    if (!part_dev) return RATO_EINVAL;
    const int64_t stride = RATO_N_STATS + nc;
    rc = system == 0 ? rato_drone_tail_rows_rollout(static_cast<const rato_drone_params*>(params), uk, s0, s1, s2, m_out,
                                                    arg_out, res_dev, stride, nullptr, 1, alphaM, part_dev, stream)
                     : rato_car_tail_rows_rollout(static_cast<const rato_car_params*>(params), uk, s0, s1, s2, s3, m_out,
                                                  arg_out, res_dev, stride, nullptr, 1, alphaM, part_dev, stream);
    if (rc != RATO_OK) return rc;
  }
  // column sums of the cut + the statistics, written by ONE launch into the device record and into its pinned host copy
  // (res_host must be device-visible host memory: no copy node follows)
  rc = rato::launch_cut_finish(part_dev, (int)rato::nblocks_for(M), S > 1 ? nc : 0, res_dev, res_host, RATO_N_STATS, st);
  if (rc != RATO_OK) return rc;
  e = rato::readback_wait(res_host, n_words, st);
  if (e != hipSuccess) return RATO_EHIP - (int)e;
  return RATO_OK;
}

// rato_kkt_sums: see kkt_sums_kernel.  part: [ceil(M/256)][2K + 2] doubles; reduce columns 0 .. 2K with
// rato_sum_partials_f64 and column 2K + 1 with a maximum.
extern "C" int rato_kkt_sums(const float* m_star, int64_t M, const float* m_base, const double* stats_base,
                             int64_t stats_stride, const int32_t* slots, const double* lam, int32_t K, double alphaM,
                             double v, double* part, void* stream) {
  RATO_CLEAR_ERROR();
  if (!m_star || M < 1 || K < 0 || K > 512 || !part || (K > 0 && (!m_base || !stats_base || !slots || !lam)))
    return RATO_EINVAL;
  const int nblk = rato::nblocks_for((int32_t)M);
  const size_t lds = sizeof(double) * 4 * (size_t)(2 * K + 2);
  hipLaunchKernelGGL(kkt_sums_kernel, dim3(nblk), dim3(RATO_BLOCK), lds, rato::as_stream(stream), m_star, (long)M, m_base,
                     stats_base, (long)stats_stride, slots, lam, (int)K, alphaM, v, part);
  RATO_LAUNCH_CHECK();
  return RATO_OK;
}
