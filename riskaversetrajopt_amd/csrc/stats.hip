// Sample-mean second stage and Monte-Carlo risk statistics (gfx950).
// Replaces drone_risk.py:294-296 (mean), :661/:719 (fraction satisfied),
// drone_main_plot.py:640-652 (VaR by sorting) and drone_risk.py:663-695 (AVaR;
// the reference's dense 2M x (M+1) OSQP LP becomes an exact 3-pass radix
// select of the Rockafellar-Uryasev minimiser + the closed form of :694).
//
// Everything is deterministic: no floating-point atomics; integer histogram
// atomics commute.  Passes are separate launches on the caller's stream, so the
// kernel boundary provides inter-workgroup visibility (no in-launch hand-off).
#include "rato_common.h"

namespace {

// ------------------------------------------------------------ sum partials
// 1024 threads = 16 columns x 64 row lanes.  Row lanes stride over the blocks with 4
// independent loads in flight (the buffer is a few MB, L2-resident: latency-, not
// bandwidth-bound), then a fixed-order LDS tree gives a run-to-run identical fp64 sum.
constexpr int SP_COLS = 16, SP_ROWS = 64;

__device__ __forceinline__ void sum_partials_block(int col_block, const float* __restrict__ part, int nblocks,
                                                   int ncols, double scale, double* __restrict__ out) {
  __shared__ double red[SP_ROWS][SP_COLS + 1];
  const int cx = threadIdx.x % SP_COLS, ry = threadIdx.x / SP_COLS;
  const int c = col_block * SP_COLS + cx;
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
  if (c < ncols) {
    int b = ry;
    for (; b + 3 * SP_ROWS < nblocks; b += 4 * SP_ROWS) {
      const float v0 = part[(size_t)b * ncols + c];
      const float v1 = part[(size_t)(b + SP_ROWS) * ncols + c];
      const float v2 = part[(size_t)(b + 2 * SP_ROWS) * ncols + c];
      const float v3 = part[(size_t)(b + 3 * SP_ROWS) * ncols + c];
      a0 += (double)v0;
      a1 += (double)v1;
      a2 += (double)v2;
      a3 += (double)v3;
    }
    for (; b < nblocks; b += SP_ROWS) a0 += (double)part[(size_t)b * ncols + c];
  }
  red[ry][cx] = (a0 + a1) + (a2 + a3);
  __syncthreads();
#pragma unroll
  for (int half = SP_ROWS / 2; half > 0; half >>= 1) {  // fixed-order tree over the row lanes
    if (ry < half) red[ry][cx] += red[ry + half][cx];
    __syncthreads();
  }
  if (ry == 0 && c < ncols) out[c] = red[0][cx] * scale;
}

__global__ __launch_bounds__(SP_COLS* SP_ROWS) void sum_partials_kernel(const float* __restrict__ part, int nblocks,
                                                                       int ncols, double scale,
                                                                       double* __restrict__ out) {
  sum_partials_block(blockIdx.x, part, nblocks, ncols, scale, out);
}

// ------------------------------------------------------------ non-finite check
__global__ __launch_bounds__(RATO_BLOCK) void count_nonfinite_kernel(const float* __restrict__ x, long n,
                                                                     unsigned* __restrict__ count) {
  unsigned bad = 0;
  for (long i = (long)blockIdx.x * RATO_BLOCK + threadIdx.x; i < n; i += (long)gridDim.x * RATO_BLOCK)
    bad += !isfinite(x[i]);
  const unsigned long long m = __ballot(bad != 0);
  // per-lane counts can exceed 1 with the grid stride; add them exactly
  for (int off = 32; off > 0; off >>= 1) bad += __shfl_xor(bad, off, RATO_WAVE);
  if ((threadIdx.x & 63) == 0 && m) atomicAdd(count, bad);
}

// ------------------------------------------------------------- risk stats
constexpr int B1 = 2048, B2 = 2048, B3 = 1024;  // 11 + 11 + 10 key bits
constexpr int RS_MAX_BLOCKS = 1024;

struct Workspace {
  unsigned hist1[B1];
  unsigned hist2[B2];
  unsigned hist3[B3];
  double blockpart[RS_MAX_BLOCKS][6];  // sum Z, count(Z<=thr), max Z, tail sum, count(Z>t), count(Z==t)
  float tstar;
  unsigned nblocks;
  // single-launch path (rs_fused): self-cleaning state -- zero between calls, set up once by rato_risk_stats_init
  unsigned fhist[B1];
  unsigned fticket;
  unsigned magic;
};
constexpr unsigned RS_MAGIC = 0x52A70517u;

// order-preserving map float -> uint32 (ascending)
__device__ __forceinline__ unsigned key_of(float f) {
  const unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float value_of(unsigned k) {
  const unsigned u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
  return __uint_as_float(u);
}

// Whole block: find the bin containing ascending rank k in hist[0..NB) and the
// rank remaining inside that bin.  Result is returned to every thread.
template <int NB, int NT = RATO_BLOCK>
__device__ void find_bin(const unsigned* __restrict__ hist, unsigned k, unsigned& bin, unsigned& krem) {
  constexpr int PER = NB / NT;
  static_assert(PER >= 1 && PER * NT == NB, "bins must divide evenly over the threads");
  __shared__ unsigned wsum[NT / RATO_WAVE];
  __shared__ unsigned res[2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  unsigned local[PER], tot = 0;
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    local[i] = hist[tid * PER + i];
    tot += local[i];
  }
  unsigned incl = tot;  // inclusive scan across the wave
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const unsigned n = __shfl_up(incl, off, RATO_WAVE);
    if (lane >= off) incl += n;
  }
  if (lane == 63) wsum[wave] = incl;
  __syncthreads();
  unsigned base = 0;
  for (int w = 0; w < wave; ++w) base += wsum[w];
  unsigned excl = base + incl - tot;
  if (k >= excl && k < excl + tot) {
    unsigned run = excl;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      if (k >= run && k < run + local[i]) {
        res[0] = tid * PER + i;
        res[1] = k - run;
      }
      run += local[i];
    }
  }
  __syncthreads();
  bin = res[0];
  krem = res[1];
  __syncthreads();
}

template <int NB>
__device__ void flush_hist(unsigned* lds_hist, unsigned* __restrict__ ghist) {
  __syncthreads();
  for (int i = threadIdx.x; i < NB; i += RATO_BLOCK) {
    const unsigned c = lds_hist[i];
    if (c) atomicAdd(&ghist[i], c);
  }
}

// zeroes the three histograms (a kernel rather than hipMemsetAsync: memset nodes of this size did not
// replay correctly inside a captured hipGraph on ROCm 7.2)
__global__ __launch_bounds__(RATO_BLOCK) void rs_zero(Workspace* __restrict__ ws) {
  unsigned* h = ws->hist1;  // hist1, hist2, hist3 are contiguous
  for (int i = blockIdx.x * RATO_BLOCK + threadIdx.x; i < B1 + B2 + B3; i += gridDim.x * RATO_BLOCK) h[i] = 0;
}

__global__ __launch_bounds__(RATO_BLOCK) void rs_pass1(const float* __restrict__ Z, long M, float thr,
                                                       Workspace* __restrict__ ws) {
  __shared__ unsigned h[B1];
  __shared__ double red[3][RATO_BLOCK / RATO_WAVE];
  for (int i = threadIdx.x; i < B1; i += RATO_BLOCK) h[i] = 0;
  __syncthreads();
  double sum = 0.0, cnt = 0.0;
  float mx = -INFINITY;
  for (long i = (long)blockIdx.x * RATO_BLOCK + threadIdx.x; i < M; i += (long)gridDim.x * RATO_BLOCK) {
    const float z = Z[i];
    atomicAdd(&h[key_of(z) >> 21], 1u);
    sum += (double)z;
    cnt += (z <= thr) ? 1.0 : 0.0;
    mx = fmaxf(mx, z);
  }
  flush_hist<B1>(h, ws->hist1);
  sum = rato::wave_sum(sum);
  cnt = rato::wave_sum(cnt);
  mx = rato::wave_max(mx);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) {
    red[0][wave] = sum;
    red[1][wave] = cnt;
    red[2][wave] = (double)mx;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double s = 0, c = 0, m = -INFINITY;
    for (int w = 0; w < RATO_BLOCK / RATO_WAVE; ++w) {
      s += red[0][w];
      c += red[1][w];
      m = fmax(m, red[2][w]);
    }
    ws->blockpart[blockIdx.x][0] = s;
    ws->blockpart[blockIdx.x][1] = c;
    ws->blockpart[blockIdx.x][2] = m;
  }
}

__global__ __launch_bounds__(RATO_BLOCK) void rs_pass2(const float* __restrict__ Z, long M, unsigned k,
                                                       Workspace* __restrict__ ws) {
  __shared__ unsigned h[B2];
  for (int i = threadIdx.x; i < B2; i += RATO_BLOCK) h[i] = 0;
  unsigned b1, k1;
  find_bin<B1>(ws->hist1, k, b1, k1);
  for (long i = (long)blockIdx.x * RATO_BLOCK + threadIdx.x; i < M; i += (long)gridDim.x * RATO_BLOCK) {
    const unsigned key = key_of(Z[i]);
    if ((key >> 21) == b1) atomicAdd(&h[(key >> 10) & (B2 - 1)], 1u);
  }
  flush_hist<B2>(h, ws->hist2);
}

__global__ __launch_bounds__(RATO_BLOCK) void rs_pass3(const float* __restrict__ Z, long M, unsigned k,
                                                       Workspace* __restrict__ ws) {
  __shared__ unsigned h[B3];
  for (int i = threadIdx.x; i < B3; i += RATO_BLOCK) h[i] = 0;
  unsigned b1, k1, b2, k2;
  find_bin<B1>(ws->hist1, k, b1, k1);
  find_bin<B2>(ws->hist2, k1, b2, k2);
  const unsigned prefix = (b1 << 11) | b2;
  for (long i = (long)blockIdx.x * RATO_BLOCK + threadIdx.x; i < M; i += (long)gridDim.x * RATO_BLOCK) {
    const unsigned key = key_of(Z[i]);
    if ((key >> 10) == prefix) atomicAdd(&h[key & (B3 - 1)], 1u);
  }
  flush_hist<B3>(h, ws->hist3);
}

__global__ __launch_bounds__(RATO_BLOCK) void rs_tail(const float* __restrict__ Z, long M, unsigned k,
                                                      Workspace* __restrict__ ws) {
  __shared__ double red[3][RATO_BLOCK / RATO_WAVE];
  unsigned b1, k1, b2, k2, b3, k3;
  find_bin<B1>(ws->hist1, k, b1, k1);
  find_bin<B2>(ws->hist2, k1, b2, k2);
  find_bin<B3>(ws->hist3, k2, b3, k3);
  const float t = value_of((b1 << 21) | (b2 << 10) | b3);
  double tail = 0.0, ngt = 0.0, neq = 0.0;
  for (long i = (long)blockIdx.x * RATO_BLOCK + threadIdx.x; i < M; i += (long)gridDim.x * RATO_BLOCK) {
    const float z = Z[i];
    tail += (z > t) ? ((double)z - (double)t) : 0.0;
    ngt += (z > t) ? 1.0 : 0.0;
    neq += (z == t) ? 1.0 : 0.0;
  }
  tail = rato::wave_sum(tail);
  ngt = rato::wave_sum(ngt);
  neq = rato::wave_sum(neq);
  if ((threadIdx.x & 63) == 0) {
    red[0][threadIdx.x >> 6] = tail;
    red[1][threadIdx.x >> 6] = ngt;
    red[2][threadIdx.x >> 6] = neq;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double s = 0, g = 0, e = 0;
    for (int w = 0; w < RATO_BLOCK / RATO_WAVE; ++w) {
      s += red[0][w];
      g += red[1][w];
      e += red[2][w];
    }
    ws->blockpart[blockIdx.x][3] = s;
    ws->blockpart[blockIdx.x][4] = g;
    ws->blockpart[blockIdx.x][5] = e;
    if (blockIdx.x == 0) ws->tstar = t;
  }
}

__global__ __launch_bounds__(RATO_WAVE) void rs_final(long M, double alpha, unsigned k, int var_is_max, int nblocks,
                                                       const Workspace* __restrict__ ws, double* __restrict__ out) {
  // one wave: lane i folds blocks i, i+64, ... in order, then a fixed shuffle tree (deterministic)
  const int lane = threadIdx.x;
  double s = 0, c = 0, m = -INFINITY, tail = 0, ngt = 0, neq = 0;
  for (int b = lane; b < nblocks; b += RATO_WAVE) {
    s += ws->blockpart[b][0];
    c += ws->blockpart[b][1];
    m = fmax(m, ws->blockpart[b][2]);
    tail += ws->blockpart[b][3];
    ngt += ws->blockpart[b][4];
    neq += ws->blockpart[b][5];
  }
  s = rato::wave_sum(s);
  c = rato::wave_sum(c);
  tail = rato::wave_sum(tail);
  ngt = rato::wave_sum(ngt);
  neq = rato::wave_sum(neq);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = fmax(m, __shfl_xor(m, off, RATO_WAVE));
  if (lane != 0) return;
  const double t = (double)ws->tstar;
  // VaR = sort(Z)[M - floor(alpha M) - 1]; when floor(alpha M) == M the reference's
  // index is -1, which NumPy wraps to the LAST element (drone_main_plot.py:651).
  out[0] = var_is_max ? m : t;
  out[1] = t + (tail / (double)M) / alpha;     // CVaR (drone_risk.py:694)
  out[2] = c / (double)M;                      // fraction satisfied
  out[3] = s / (double)M;
  out[4] = m;
  out[5] = c;
  out[6] = tail;
  out[7] = (double)k;
  out[8] = ngt;                                // #{Z > t}
  out[9] = neq;                                // #{Z == t}
  out[10] = t;                                 // the Rockafellar-Uryasev minimiser itself (== out[0] unless var_is_max)
}

// ---- single-workgroup form for small M (<= RS_SINGLE_MAX): the whole selection in ONE launch (Z is a few tens of
// KB and L2-resident; five ~5 us launches become one ~10 us launch).  Same arithmetic, fixed reduction order.
constexpr int RS1_T = 1024;
// Measured per call (tools/stats_time.py): one workgroup 15 / 19 / 24 / 36 us at M = 1e3 / 4e3 / 1e4 / 1.6e4 against
// ~25 us for the six launches at any M <= 1e5 (launch-bound); and 8 us instead of 46 us of HOST issue time.  (Wave-level
// aggregation of the LDS histogram updates was tried and bought nothing: the cost is passes and barriers, not conflicts.)
constexpr long RS_SINGLE_MAX = 1 << 13;

__device__ __forceinline__ double block_sum_1024(double v, double* red) {
  v = rato::wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  double s = 0;
  for (int w = 0; w < RS1_T / RATO_WAVE; ++w) s += red[w];   // fixed order, every thread computes the same
  return s;
}

__global__ __launch_bounds__(RS1_T) void rs_single(const float* __restrict__ Z, long M, double alpha, unsigned k,
                                                   int var_is_max, float thr, double* __restrict__ out) {
  __shared__ unsigned h[B1];
  __shared__ double red[RS1_T / RATO_WAVE];
  const int tid = threadIdx.x;
  for (int i = tid; i < B1; i += RS1_T) h[i] = 0;
  __syncthreads();
  double sum = 0.0, cnt = 0.0;
  float mx = -INFINITY;
  for (long i = tid; i < M; i += RS1_T) {
    const float z = Z[i];
    atomicAdd(&h[key_of(z) >> 21], 1u);
    sum += (double)z;
    cnt += (z <= thr) ? 1.0 : 0.0;
    mx = fmaxf(mx, z);
  }
  __syncthreads();
  unsigned b1, k1, b2, k2, b3, k3;
  find_bin<B1, RS1_T>(h, k, b1, k1);
  for (int i = tid; i < B2; i += RS1_T) h[i] = 0;
  __syncthreads();
  for (long i = tid; i < M; i += RS1_T) {
    const unsigned key = key_of(Z[i]);
    if ((key >> 21) == b1) atomicAdd(&h[(key >> 10) & (B2 - 1)], 1u);
  }
  __syncthreads();
  find_bin<B2, RS1_T>(h, k1, b2, k2);
  for (int i = tid; i < B3; i += RS1_T) h[i] = 0;
  __syncthreads();
  const unsigned prefix = (b1 << 11) | b2;
  for (long i = tid; i < M; i += RS1_T) {
    const unsigned key = key_of(Z[i]);
    if ((key >> 10) == prefix) atomicAdd(&h[key & (B3 - 1)], 1u);
  }
  __syncthreads();
  find_bin<B3, RS1_T>(h, k2, b3, k3);
  const float t = value_of((b1 << 21) | (b2 << 10) | b3);
  double tail = 0.0, ngt = 0.0, neq = 0.0;
  for (long i = tid; i < M; i += RS1_T) {
    const float z = Z[i];
    tail += (z > t) ? ((double)z - (double)t) : 0.0;
    ngt += (z > t) ? 1.0 : 0.0;
    neq += (z == t) ? 1.0 : 0.0;
  }
  const double S = block_sum_1024(sum, red);
  const double C = block_sum_1024(cnt, red);
  const double T = block_sum_1024(tail, red);
  const double NG = block_sum_1024(ngt, red);
  const double NE = block_sum_1024(neq, red);
  mx = rato::wave_max(mx);
  __syncthreads();
  if ((tid & 63) == 0) red[tid >> 6] = (double)mx;
  __syncthreads();
  if (tid == 0) {
    double m = -INFINITY;
    for (int w = 0; w < RS1_T / RATO_WAVE; ++w) m = fmax(m, red[w]);
    out[0] = var_is_max ? m : (double)t;
    out[1] = (double)t + (T / (double)M) / alpha;
    out[2] = C / (double)M;
    out[3] = S / (double)M;
    out[4] = m;
    out[5] = C;
    out[6] = T;
    out[7] = (double)k;
    out[8] = NG;
    out[9] = NE;
    out[10] = (double)t;
  }
}


// ---- ONE launch for 8192 < M <= RS_FUSED_MAX (the BASELINE configs C2-C4 and the metric's M = 1e5), optionally with
// the second stage of the sample mean riding along in extra workgroups (what used to be rs_zero + 3 passes + tail +
// final + sum_partials = 7 launches, ~25-35 us of launch-bound time per step, as long as a whole linearize kernel at
// M = 1e4).
//   stage 1, g_hist workgroups: LDS histogram of the top 11 key bits of a slice of Z, flushed with RETURNING integer
//            atomics into ws->fhist; every wave waits for its atomics, the workgroup takes a ticket (one returning
//            atomic).  No spinning anywhere: a workgroup that is not last simply exits.
//   stage 2, the workgroup whose ticket is last: reads AND re-zeroes ws->fhist with atomic exchanges (the same
//            coherence point as the adds: no cache has to be trusted), finds the bin b1 of the wanted rank, then makes
//            ONE pass over Z (16-byte loads): sum, count(Z <= thr), max, the part of the tail that lies in bins above
//            b1 (accumulated against the lower edge of bin b1 + 1, all terms >= 0), and the keys of bin b1 compacted
//            into LDS (wave-aggregated append).  The remaining 21 key bits are selected inside LDS (2 histogram passes
//            over the candidates only).  The candidates' share of the tail is EXACT integer arithmetic: inside a bin
//            the exponent is fixed, so z = v0 + low21 * ulp and sum_{z > t}(z - t) = ulp * (sum low21 - n * low21(t)).
//            If bin b1 holds more than RSF_CAP keys (heavily tied / clustered data) the same steps re-read Z instead.
// Deterministic: integer atomics commute, every floating-point sum has a fixed order.
constexpr long RS_FUSED_MAX = 1 << 17;
constexpr int RSF_CAP = 10240;

__device__ __forceinline__ double block_max_1024(double v, double* red) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off, RATO_WAVE));
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  double m = -INFINITY;
  for (int w = 0; w < RS1_T / RATO_WAVE; ++w) m = fmax(m, red[w]);
  return m;
}

__global__ __launch_bounds__(RS1_T) void rs_fused(const float* __restrict__ Z, long M, double alpha, unsigned k,
                                                  int var_is_max, float thr, Workspace* __restrict__ ws,
                                                  double* __restrict__ out, int g_hist,
                                                  const float* __restrict__ part, int nblocks, int ncols,
                                                  double scale, double* __restrict__ sums_out) {
  if ((int)blockIdx.x >= g_hist) {   // the sample-mean second stage rides along (independent workgroups)
    sum_partials_block(blockIdx.x - g_hist, part, nblocks, ncols, scale, sums_out);
    return;
  }
  __shared__ unsigned h[B1];
  __shared__ unsigned cand[RSF_CAP];
  __shared__ double red[RS1_T / RATO_WAVE];
  __shared__ unsigned sh[2];
  const int tid = threadIdx.x, lane = tid & 63;
  if (ws->magic != RS_MAGIC) {       // workspace never initialised (rato_risk_stats_init): fail loudly, touch nothing
    if (blockIdx.x == 0 && tid < 11) out[tid] = __longlong_as_double(0x7ff8000000000000LL);
    return;
  }
  // ---- stage 1
  for (int i = tid; i < B1; i += RS1_T) h[i] = 0;
  __syncthreads();
  for (long i = (long)blockIdx.x * RS1_T + tid; i < M; i += (long)g_hist * RS1_T) atomicAdd(&h[key_of(Z[i]) >> 21], 1u);
  __syncthreads();
  unsigned keep = 0;
  for (int i = tid; i < B1; i += RS1_T) {
    const unsigned c = h[i];
    if (c) keep += __hip_atomic_fetch_add(&ws->fhist[i], c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::"v"(keep) : "memory");   // this wave's adds have been performed (values returned)
  __syncthreads();
  if (tid == 0) sh[0] = __hip_atomic_fetch_add(&ws->fticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  if (sh[0] != (unsigned)(g_hist - 1)) return;
  // ---- stage 2: this workgroup's ticket is the last one: every other workgroup's adds precede it
  for (int i = tid; i < B1; i += RS1_T)
    h[i] = __hip_atomic_exchange(&ws->fhist[i], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (tid == 0) {
    __hip_atomic_exchange(&ws->fticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    sh[1] = 0;                         // candidate cursor
  }
  __syncthreads();
  unsigned b1, k1, b2, k2, b3, k3;
  find_bin<B1, RS1_T>(h, k, b1, k1);
  const double r_hi = (b1 + 1 < (unsigned)B1) ? (double)value_of((b1 + 1) << 21) : 0.0;   // lower edge of bin b1 + 1
  double sum = 0.0, s_hi = 0.0;
  unsigned cnt_le = 0, n_hi = 0;
  float mx = -INFINITY;
  auto visit = [&](float z, bool live) {
    const unsigned key = key_of(z), bin = key >> 21;
    const bool is_c = live && bin == b1;
    if (live) {
      sum += (double)z;
      cnt_le += (z <= thr) ? 1u : 0u;
      mx = fmaxf(mx, z);
      if (bin > b1) {
        s_hi += (double)z - r_hi;
        ++n_hi;
      }
    }
    const unsigned long long mask = __ballot(is_c);   // wave-aggregated append of the candidates
    if (mask) {
      unsigned base = 0;
      const int leader = __ffsll((long long)mask) - 1;
      if (lane == leader) base = atomicAdd(&sh[1], (unsigned)__popcll(mask));
      base = __shfl(base, leader, RATO_WAVE);
      if (is_c) {
        const unsigned pos = base + (unsigned)__popcll(mask & ((1ull << lane) - 1ull));
        if (pos < (unsigned)RSF_CAP) cand[pos] = key;
      }
    }
  };
  {
    // 16-byte loads over the aligned body, scalar head / tail
    const uintptr_t addr = reinterpret_cast<uintptr_t>(Z);
    long head = (long)(((16 - (addr & 15)) & 15) >> 2);
    if (head > M) head = M;
    const long nvec = (M - head) >> 2;
    const float4* __restrict__ Zv = reinterpret_cast<const float4*>(Z + head);
    const long rounds = (nvec + RS1_T - 1) / RS1_T;     // every wave runs the same trip count (ballots inside)
    for (long rnd = 0; rnd < rounds; rnd += 4) {
      float4 v[4];
      bool lv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long i = (rnd + u) * RS1_T + tid;
        lv[u] = (rnd + u) < rounds && i < nvec;
        v[u] = lv[u] ? Zv[i] : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if ((rnd + u) < rounds) {
          visit(v[u].x, lv[u]);
          visit(v[u].y, lv[u]);
          visit(v[u].z, lv[u]);
          visit(v[u].w, lv[u]);
        }
      }
    }
    const long rest0 = head + (nvec << 2);
    {   // head [0, head) and tail [rest0, M): at most 3 + 3 elements, one round
      const long i = (tid < head) ? tid : (rest0 + (tid - head));
      const bool live = i < M && (tid < head || (tid - head) < (M - rest0));
      visit(live ? Z[i] : 0.f, live);
    }
  }
  __syncthreads();
  const unsigned ncand = sh[1];
  const bool in_lds = ncand <= (unsigned)RSF_CAP;
  auto for_each_cand = [&](auto&& f) {
    if (in_lds) {
      for (unsigned i = tid; i < ncand; i += RS1_T) f(cand[i]);
    } else {
      for (long i = tid; i < M; i += RS1_T) {
        const unsigned key = key_of(Z[i]);
        if ((key >> 21) == b1) f(key);
      }
    }
  };
  for (int i = tid; i < B2; i += RS1_T) h[i] = 0;
  __syncthreads();
  for_each_cand([&](unsigned key) { atomicAdd(&h[(key >> 10) & (B2 - 1)], 1u); });
  __syncthreads();
  find_bin<B2, RS1_T>(h, k1, b2, k2);
  for (int i = tid; i < B3; i += RS1_T) h[i] = 0;
  __syncthreads();
  const unsigned prefix = (b1 << 11) | b2;
  for_each_cand([&](unsigned key) {
    if ((key >> 10) == prefix) atomicAdd(&h[key & (B3 - 1)], 1u);
  });
  __syncthreads();
  find_bin<B3, RS1_T>(h, k2, b3, k3);
  const unsigned n_eq = h[b3];
  const unsigned tkey = (b1 << 21) | (b2 << 10) | b3;
  const float t = value_of(tkey);
  // candidates above t: count and exact integer sum of the low 21 key bits
  unsigned long long low_sum = 0;
  unsigned c_gt = 0;
  for_each_cand([&](unsigned key) {
    if (key > tkey) {
      low_sum += key & 0x1fffffu;
      ++c_gt;
    }
  });
  const double S = block_sum_1024(sum, red);
  const double SH = block_sum_1024(s_hi, red);
  const double C = block_sum_1024((double)cnt_le, red);
  const double NH = block_sum_1024((double)n_hi, red);
  const double CG = block_sum_1024((double)c_gt, red);
  const double LS = block_sum_1024((double)low_sum, red);     // per-thread sums < 2^53: exact in fp64, so is the total
  const double MX = block_max_1024((double)mx, red);
  if (tid == 0) {
    const double td = (double)t;
    const double v0 = (double)value_of(b1 << 21);
    const double ulp = (double)value_of((b1 << 21) | 1u) - v0;   // spacing of the floats inside bin b1 (exact)
    const double tail_c = ulp * (LS - CG * (double)(tkey & 0x1fffffu));
    const double tail_h = (NH > 0.0) ? (SH + NH * (r_hi - td)) : 0.0;
    const double T = tail_c + tail_h;
    out[0] = var_is_max ? MX : td;
    out[1] = td + (T / (double)M) / alpha;
    out[2] = C / (double)M;
    out[3] = S / (double)M;
    out[4] = MX;
    out[5] = C;
    out[6] = T;
    out[7] = (double)k;
    out[8] = NH + CG;
    out[9] = (double)n_eq;
    out[10] = td;
  }
}

// one workgroup: zero the whole workspace, then tag it
__global__ __launch_bounds__(RATO_BLOCK) void rs_init_kernel(Workspace* __restrict__ ws) {
  unsigned* w = reinterpret_cast<unsigned*>(ws);
  const int n = (int)(sizeof(Workspace) / sizeof(unsigned));
  for (int i = blockIdx.x * RATO_BLOCK + threadIdx.x; i < n; i += gridDim.x * RATO_BLOCK) w[i] = 0;
  __syncthreads();
  if (blockIdx.x == 0 && threadIdx.x == 0) ws->magic = RS_MAGIC;
}

// ------------------------------------------------------------ gathered records
// After the one all-gather of an evaluation every rank holds world records [fp64 sums (n_sums) | fp32 Z row];
// one launch turns them into the contiguous Z (rank order) and the totals summed in rank order (bitwise
// identical on every rank) -- instead of a handful of slice / copy / add launches per step.
__global__ __launch_bounds__(RATO_BLOCK) void unpack_records_kernel(const unsigned char* __restrict__ all, int world,
                                                                    int n_sums, long M_local, long rec_bytes,
                                                                    double* __restrict__ total,
                                                                    float* __restrict__ Z_all) {
  const long n = (long)world * M_local;
  for (long i = (long)blockIdx.x * RATO_BLOCK + threadIdx.x; i < n; i += (long)gridDim.x * RATO_BLOCK) {
    const long r = i / M_local, j = i - r * M_local;
    Z_all[i] = reinterpret_cast<const float*>(all + r * rec_bytes + 8L * n_sums)[j];
  }
  if (blockIdx.x == 0) {
    for (int c = threadIdx.x; c < n_sums; c += RATO_BLOCK) {
      double acc = reinterpret_cast<const double*>(all)[c];
      for (int r = 1; r < world; ++r) acc += reinterpret_cast<const double*>(all + (long)r * rec_bytes)[c];
      total[c] = acc;
    }
  }
}

}  // namespace

extern "C" int rato_unpack_records(const void* all, int32_t world, int32_t n_sums, int64_t M_local, int64_t rec_bytes,
                                   double* total, float* Z_all, void* stream) {
  RATO_CLEAR_ERROR();
  if (!all || (!total && n_sums > 0) || !Z_all || world < 1 || n_sums < 0 || M_local <= 0 || rec_bytes < 8L * n_sums + 4 * M_local ||
      rec_bytes % 8 != 0)
    return RATO_EINVAL;
  long nb = ((long)world * M_local + RATO_BLOCK * 4 - 1) / (RATO_BLOCK * 4);
  if (nb > 2048) nb = 2048;
  if (nb < 1) nb = 1;
  hipLaunchKernelGGL(unpack_records_kernel, dim3((unsigned)nb), dim3(RATO_BLOCK), 0, rato::as_stream(stream),
                     static_cast<const unsigned char*>(all), world, n_sums, (long)M_local, (long)rec_bytes, total, Z_all);
  RATO_LAUNCH_CHECK();
  return RATO_OK;
}

extern "C" int rato_sum_partials(const float* part, int32_t nblocks, int32_t ncols, double scale, double* out,
                                 void* stream) {
  RATO_CLEAR_ERROR();
  if (!part || !out || nblocks <= 0 || ncols <= 0) return RATO_EINVAL;
  dim3 grid((ncols + SP_COLS - 1) / SP_COLS), block(SP_COLS * SP_ROWS);
  hipLaunchKernelGGL(sum_partials_kernel, grid, block, 0, rato::as_stream(stream), part, nblocks, ncols, scale, out);
  RATO_LAUNCH_CHECK();
  return RATO_OK;
}

extern "C" int rato_count_nonfinite(const float* x, int64_t n, uint32_t* count, void* stream) {
  RATO_CLEAR_ERROR();
  if (!x || !count || n <= 0) return RATO_EINVAL;
  hipStream_t st = rato::as_stream(stream);
  hipError_t e = hipMemsetAsync(count, 0, sizeof(uint32_t), st);
  if (e != hipSuccess) return RATO_EHIP - (int)e;
  long nb = (n + RATO_BLOCK * 8 - 1) / (RATO_BLOCK * 8);
  if (nb > 2048) nb = 2048;
  hipLaunchKernelGGL(count_nonfinite_kernel, dim3((unsigned)nb), dim3(RATO_BLOCK), 0, st, x, (long)n, count);
  RATO_LAUNCH_CHECK();
  return RATO_OK;
}

extern "C" int rato_count_nonfinite_acc(const float* x, int64_t n, uint32_t* count, void* stream) {
  RATO_CLEAR_ERROR();
  if (!x || !count || n <= 0) return RATO_EINVAL;
  long nb = (n + RATO_BLOCK * 8 - 1) / (RATO_BLOCK * 8);
  if (nb > 2048) nb = 2048;
  hipLaunchKernelGGL(count_nonfinite_kernel, dim3((unsigned)nb), dim3(RATO_BLOCK), 0, rato::as_stream(stream), x,
                     (long)n, count);
  RATO_LAUNCH_CHECK();
  return RATO_OK;
}

extern "C" size_t rato_risk_stats_workspace_bytes(int64_t M) {
  (void)M;
  return sizeof(Workspace);
}

namespace {
int risk_stats_impl(const float* Z, int64_t M, double alpha, float thr, void* workspace, size_t workspace_bytes,
                    double* out, const float* part, int32_t nblocks, int32_t ncols, double scale, double* sums_out,
                    void* stream) {
  RATO_CLEAR_ERROR();
  if (!Z || !out || !workspace || M <= 0 || M >= (int64_t)0xffffffffLL || !(alpha > 0.0) || !(alpha <= 1.0))
    return RATO_EINVAL;
  if (workspace_bytes < sizeof(Workspace)) return RATO_EINVAL;
  if (part && (!sums_out || nblocks <= 0 || ncols <= 0)) return RATO_EINVAL;
  // ascending 0-based rank of sort(Z)[M - floor(alpha*M) - 1]  (drone_main_plot.py:649-651)
  long xth = (long)floor(alpha * (double)M);
  long kk = M - xth - 1;
  int var_is_max = 0;
  if (kk < 0) {  // alpha*M == M: the reference's index -1 wraps to the maximum
    kk = 0;
    var_is_max = 1;
  }
  const unsigned k = (unsigned)kk;
  hipStream_t st = rato::as_stream(stream);
  Workspace* ws = static_cast<Workspace*>(workspace);
  const int sp_blocks = part ? (ncols + SP_COLS - 1) / SP_COLS : 0;
  if (M <= RS_SINGLE_MAX && !part) {
    hipLaunchKernelGGL(rs_single, dim3(1), dim3(RS1_T), 0, st, Z, (long)M, alpha, k, var_is_max, thr, out);
    RATO_LAUNCH_CHECK();
    return RATO_OK;
  }
  if (M <= RS_FUSED_MAX) {   // ONE launch: histogram workgroups + (optionally) the partial-sum workgroups
    int g_hist = (int)((M + 4 * RS1_T - 1) / (4 * RS1_T));
    if (g_hist > 32) g_hist = 32;
    if (g_hist < 1) g_hist = 1;
    hipLaunchKernelGGL(rs_fused, dim3(g_hist + sp_blocks), dim3(RS1_T), 0, st, Z, (long)M, alpha, k, var_is_max, thr, ws,
                       out, g_hist, part, (int)nblocks, (int)ncols, scale, sums_out);
    RATO_LAUNCH_CHECK();
    return RATO_OK;
  }
  if (part) {
    hipLaunchKernelGGL(sum_partials_kernel, dim3(sp_blocks), dim3(SP_COLS * SP_ROWS), 0, st, part, (int)nblocks,
                       (int)ncols, scale, sums_out);
  }
  // Grid: 4 elements per thread up to 256 workgroups, then more elements per thread (every workgroup flushes its
  // LDS histogram with global atomics: at M ~ 1e6 of CLUSTERED values 1000 workgroups hammering the same few bins
  // cost 60 us per call, 256 workgroups 33 us; tools/stats_time.py), capped at RS_MAX_BLOCKS for very large M.
  long nb = (M + RATO_BLOCK * 4 - 1) / (RATO_BLOCK * 4);
  if (nb > 256) {
    nb = (M + RATO_BLOCK * 16 - 1) / (RATO_BLOCK * 16);
    if (nb < 256) nb = 256;
  }
  if (nb > RS_MAX_BLOCKS) nb = RS_MAX_BLOCKS;
  if (nb < 1) nb = 1;
  dim3 grid((unsigned)nb), block(RATO_BLOCK);
  hipLaunchKernelGGL(rs_zero, dim3(4), block, 0, st, ws);
  hipLaunchKernelGGL(rs_pass1, grid, block, 0, st, Z, (long)M, thr, ws);
  hipLaunchKernelGGL(rs_pass2, grid, block, 0, st, Z, (long)M, k, ws);
  hipLaunchKernelGGL(rs_pass3, grid, block, 0, st, Z, (long)M, k, ws);
  hipLaunchKernelGGL(rs_tail, grid, block, 0, st, Z, (long)M, k, ws);
  hipLaunchKernelGGL(rs_final, dim3(1), dim3(RATO_WAVE), 0, st, (long)M, alpha, k, var_is_max, (int)nb, ws, out);
  RATO_LAUNCH_CHECK();
  return RATO_OK;
}
}  // namespace

extern "C" int rato_risk_stats_init(void* workspace, size_t workspace_bytes, void* stream) {
  RATO_CLEAR_ERROR();
  if (!workspace || workspace_bytes < sizeof(Workspace)) return RATO_EINVAL;
  hipLaunchKernelGGL(rs_init_kernel, dim3(1), dim3(RATO_BLOCK), 0, rato::as_stream(stream),
                     static_cast<Workspace*>(workspace));
  RATO_LAUNCH_CHECK();
  return RATO_OK;
}

extern "C" int rato_risk_stats(const float* Z, int64_t M, double alpha, float thr, void* workspace,
                               size_t workspace_bytes, double* out, void* stream) {
  return risk_stats_impl(Z, M, alpha, thr, workspace, workspace_bytes, out, nullptr, 0, 0, 1.0, nullptr, stream);
}

extern "C" int rato_sums_and_risk_stats(const float* part, int32_t nblocks, int32_t ncols, double scale,
                                        double* sums_out, const float* Z, int64_t M, double alpha, float thr,
                                        void* workspace, size_t workspace_bytes, double* out, void* stream) {
  if (!part) return RATO_EINVAL;
  return risk_stats_impl(Z, M, alpha, thr, workspace, workspace_bytes, out, part, nblocks, ncols, scale, sums_out,
                         stream);
}
