// Sample-mean second stage and Monte-Carlo risk statistics (gfx950).
// Replaces drone_risk.py:294-296 (mean), :661/:719 (fraction satisfied),
// drone_main_plot.py:640-652 (VaR by sorting) and drone_risk.py:663-695 (AVaR;
// the reference's dense 2M x (M+1) OSQP LP becomes an exact 3-pass radix
// select of the Rockafellar-Uryasev minimiser + the closed form of :694).
//
// Everything is deterministic: no floating-point atomics; integer histogram
// atomics commute.  Passes are separate launches on the caller's stream, so the
// kernel boundary provides inter-workgroup visibility (no in-launch hand-off).
#include <stdlib.h>

#include <atomic>

#include "rato_common.h"
#include "rato_select.h"

namespace {
using namespace rato_sel;

// ------------------------------------------------------------ sum partials
// 1024 threads = 16 columns x 64 row lanes.  Row lanes stride over the blocks with 4
// independent loads in flight (the buffer is a few MB, L2-resident: latency-, not
// bandwidth-bound), then a fixed-order LDS tree gives a run-to-run identical fp64 sum.
constexpr int SP_COLS = 16, SP_ROWS = 64;

template <typename T>
__device__ __forceinline__ void sum_partials_block(int col_block, const T* __restrict__ part, int nblocks,
                                                   int ncols, double scale, double* __restrict__ out) {
  __shared__ double red[SP_ROWS][SP_COLS + 1];
  const int cx = threadIdx.x % SP_COLS, ry = threadIdx.x / SP_COLS;
  const int c = col_block * SP_COLS + cx;
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
  if (c < ncols) {
    int b = ry;
    for (; b + 3 * SP_ROWS < nblocks; b += 4 * SP_ROWS) {
      const T v0 = part[(size_t)b * ncols + c];
      const T v1 = part[(size_t)(b + SP_ROWS) * ncols + c];
      const T v2 = part[(size_t)(b + 2 * SP_ROWS) * ncols + c];
      const T v3 = part[(size_t)(b + 3 * SP_ROWS) * ncols + c];
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

template <typename T>
__global__ __launch_bounds__(SP_COLS* SP_ROWS) void sum_partials_kernel(const T* __restrict__ part, int nblocks,
                                                                       int ncols, double scale,
                                                                       double* __restrict__ out) {
  sum_partials_block(blockIdx.x, part, nblocks, ncols, scale, out);
}

// The last launch of an oracle round trip (cvar.hip: rato_cut_oracle_rollout): the column sums of the cut's block
// partials, written to the record on the device AND straight into the pinned host copy of it, while one more workgroup
// copies the statistics the selection left in the record -- instead of a reduction launch followed by a device-to-host
// copy (a copy node costs more than this whole kernel).  Same fixed-order sums as sum_partials_kernel<double>.
__global__ __launch_bounds__(SP_COLS* SP_ROWS) void cut_finish_kernel(const double* __restrict__ part, int nblocks, int ncols,
                                                                     double* __restrict__ rec_dev,
                                                                     double* __restrict__ rec_host, int n_stats) {
  const int col_blocks = (ncols + SP_COLS - 1) / SP_COLS;
  if ((int)blockIdx.x == col_blocks) {
    if ((int)threadIdx.x < n_stats) rec_host[threadIdx.x] = rec_dev[threadIdx.x];
    return;
  }
  sum_partials_block(blockIdx.x, part, nblocks, ncols, 1.0, rec_dev + n_stats);
  // (sum_partials_block ends with thread row 0 holding the sums: re-read what it just wrote)
  const int cx = threadIdx.x % SP_COLS, ry = threadIdx.x / SP_COLS;
  const int c = blockIdx.x * SP_COLS + cx;
  if (ry == 0 && c < ncols) rec_host[n_stats + c] = rec_dev[n_stats + c];
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

__global__ __launch_bounds__(RATO_BLOCK) void rs_final(long M, double alpha, unsigned k, int var_is_max, int nblocks,
                                                        Workspace* __restrict__ ws, double* __restrict__ out) {
  // leaves the three histograms zeroed for the next call (the workspace starts zeroed: rato_risk_stats_init), so a
  // call is 5 launches instead of 6
  // (zeroed by a kernel rather than hipMemsetAsync: memset nodes of this size did not replay correctly inside a
  // captured hipGraph on ROCm 7.2)
  const bool ok = ws->magic == RS_MAGIC;
  for (int i = threadIdx.x; i < B1 + B2 + B3; i += RATO_BLOCK) ws->hist1[i] = 0;   // hist1..3 are contiguous
  if (!ok) {   // never initialised: the histograms held garbage -> fail loudly instead of returning numbers
    if (threadIdx.x < RATO_N_STATS) out[threadIdx.x] = __longlong_as_double(0x7ff8000000000000LL);
    return;
  }
  if (threadIdx.x >= RATO_WAVE) return;
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

// ---- ONE workgroup, ONE launch for M <= RS_SMALL_MAX (BASELINE configs C2 / C3: M = 1e4): Z is read from memory
// ONCE (every load of a thread in flight together), its order-preserving keys stay in REGISTERS (<= 12 per thread)
// and the three radix passes and the tail sum never touch memory again -- a pass costs a few hundred cycles instead
// of a memory round trip, and there is no launch boundary between passes.  The second stage of the sample mean (sum_partials) rides along in extra workgroups of
// the same launch.  Same arithmetic as the multi-launch form: exact selection, fixed-order fp64 sums.
// Measured per call, host issue excluded (tools/stats_time.py, hipGraph replay): see DESIGN.md 4.5.
// (A single-launch form for larger M -- histogram workgroups + an atomic ticket, the last workgroup finishing alone --
//  was built and measured at 25-108 us for M = 1e4-1e5 against 20 us for six launches: one CU needs 11-23 us of issue
//  time for 1e5 elements, and constraint values cluster in 4 key bins per binade, so almost nothing is filtered by the
//  first pass.  Removed; git history has it.)
constexpr int RS1_T = 1024;

__global__ __launch_bounds__(RS1_T) void rs_small(const float* __restrict__ Z, long M, double alpha, unsigned k,
                                                  int var_is_max, float thr, double* __restrict__ out,
                                                  const float* __restrict__ part, int nblocks, int ncols,
                                                  double scale, double* __restrict__ sums_out) {
  if (blockIdx.x > 0) {   // the sample-mean second stage rides along (independent workgroups)
    sum_partials_block(blockIdx.x - 1, part, nblocks, ncols, scale, sums_out);
    return;
  }
  __shared__ unsigned h[B1];
  __shared__ double red[5 * (RS1_T / RATO_WAVE)];
  __shared__ float redmax[RS1_T / RATO_WAVE];
  rs_small_body<RS1_T>(Z, M, alpha, k, var_is_max, thr, out, nullptr, h, red, redmax);   // rato_select.h (no producer to wait for)
}

// K independent selections in one launch (rato_risk_stats_batch): workgroup k on row k of Z [K][ldz] -> out [K][RATO_N_STATS]
__global__ __launch_bounds__(RS1_T) void rs_small_batch(const float* __restrict__ Z, long M, long ldz, double alpha, unsigned k,
                                                        int var_is_max, float thr, double* __restrict__ out) {
  __shared__ unsigned h[B1];
  __shared__ double red[5 * (RS1_T / RATO_WAVE)];
  __shared__ float redmax[RS1_T / RATO_WAVE];
  rs_small_body<RS1_T>(Z + (size_t)blockIdx.x * ldz, M, alpha, k, var_is_max, thr, out + (size_t)blockIdx.x * RATO_N_STATS, nullptr,
                       h, red, redmax);
}

// ---- ONE launch, a FEW workgroups for RS_SMALL_MAX < M <= RS_COOP_MAX (BASELINE config C4: M = 5e4; the metric
// config: M = 1e5).  The five launches above cost ~4 us each of dependent-launch latency for a few hundred ns of work;
// here G <= 64 workgroups of 1024 threads keep their keys in REGISTERS (<= 16 per thread), accumulate the same three
// global histograms with device-scope atomics, and wait on the histograms themselves (find_bin_coop: a pass is
// complete when its counters add up); the last workgroup to finish (completion ticket) folds the G partial sums in a
// fixed order and leaves the workspace clean.  All G workgroups must be resident at once for the waits to complete:
// G <= 64 against 256 CUs x 2 workgroups of this size, launched on an in-order stream behind the producer of Z.
// Same arithmetic as the other two forms: exact selection, fixed-order fp64 sums (deterministic run to run).
constexpr long RS_COOP_MAX = (long)RS_COOP_MAX_WG * RS1_T * RS_COOP_KEYS;   // 1,048,576

__global__ __launch_bounds__(RS1_T) void rs_coop(const float* __restrict__ Z, long M, double alpha, unsigned k,
                                                 int var_is_max, float thr, int G, Workspace* __restrict__ ws,
                                                 double* __restrict__ out, const float* __restrict__ part, int nblocks,
                                                 int ncols, double scale, double* __restrict__ sums_out) {
  if ((int)blockIdx.x >= G) {   // the sample-mean second stage rides along (independent workgroups, no barriers)
    sum_partials_block(blockIdx.x - G, part, nblocks, ncols, scale, sums_out);
    return;
  }
  __shared__ unsigned h[B1];
  __shared__ double red[6 * (RS1_T / RATO_WAVE)];
  rs_coop_body<RS1_T>(Z, M, alpha, k, var_is_max, thr, G, ws, out, (int)blockIdx.x, 0, h, red);   // rato_select.h (stand-alone)
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
  hipLaunchKernelGGL(sum_partials_kernel<float>, grid, block, 0, rato::as_stream(stream), part, nblocks, ncols, scale, out);
  RATO_LAUNCH_CHECK();
  return RATO_OK;
}

extern "C" int rato_sum_partials_f64(const double* part, int32_t nblocks, int32_t ncols, double scale, double* out,
                                     void* stream) {
  RATO_CLEAR_ERROR();
  if (!part || !out || nblocks <= 0 || ncols <= 0) return RATO_EINVAL;
  dim3 grid((ncols + SP_COLS - 1) / SP_COLS), block(SP_COLS * SP_ROWS);
  hipLaunchKernelGGL(sum_partials_kernel<double>, grid, block, 0, rato::as_stream(stream), part, nblocks, ncols, scale, out);
  RATO_LAUNCH_CHECK();
  return RATO_OK;
}

namespace rato {
int launch_cut_finish(const double* part, int nblocks, int ncols, double* rec_dev, double* rec_host, int n_stats,
                      hipStream_t st) {
  const int col_blocks = ncols > 0 ? (ncols + SP_COLS - 1) / SP_COLS : 0;
  hipLaunchKernelGGL(cut_finish_kernel, dim3(col_blocks + 1), dim3(SP_COLS * SP_ROWS), 0, st, part, nblocks, ncols, rec_dev,
                     rec_host, n_stats);
  RATO_LAUNCH_CHECK();
  return RATO_OK;
}
}  // namespace rato

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
                    void* stream, bool recover = false) {
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
  // RATO_RS_PATH=multi: diagnostic override (A/B timing of the launch structure; tools/stats_time.py)
  static const int force_multi_env = [] { const char* e = getenv("RATO_RS_PATH"); return (e && e[0] == 'm') ? 1 : 0; }();
  const int force_multi = force_multi_env || recover;   // recover: the launch-per-pass form, which waits for nothing
  static const int force_coop = [] { const char* e = getenv("RATO_RS_PATH"); return (e && e[0] == 'c') ? 1 : 0; }();
  if (M <= RS_SMALL_MAX && !force_multi && !force_coop) {   // ONE launch, one workgroup (+ the partial-sum workgroups), keys in LDS
    hipLaunchKernelGGL(rs_small, dim3(1 + sp_blocks), dim3(RS1_T), 0, st, Z, (long)M, alpha, k, var_is_max, thr, out,
                       part, (int)nblocks, (int)ncols, scale, sums_out);
    RATO_LAUNCH_CHECK();
    return RATO_OK;
  }
  static const int no_coop = [] { const char* e = getenv("RATO_RS_COOP"); return (e && e[0] == '0') ? 1 : 0; }();
  if (M <= RS_COOP_MAX && !force_multi && (!no_coop || force_coop)) {   // ONE launch, G workgroups, keys in registers
    static const int kpt_env = [] { const char* e = getenv("RATO_RS_KPT"); return e ? atoi(e) : 0; }();   // A/B: keys per thread
    const long kpt = (kpt_env >= 1 && kpt_env <= RS_COOP_KEYS) ? kpt_env : 4;
    long G = (M + RS1_T * kpt - 1) / (RS1_T * kpt);
    if (G > RS_COOP_MAX_WG) G = RS_COOP_MAX_WG;
    hipLaunchKernelGGL(rs_coop, dim3((unsigned)G + sp_blocks), dim3(RS1_T), 0, st, Z, (long)M, alpha, k, var_is_max, thr,
                       (int)G, ws, out, part, (int)nblocks, (int)ncols, scale, sums_out);
    RATO_LAUNCH_CHECK();
    return RATO_OK;
  }
  if (part) {
    hipLaunchKernelGGL(sum_partials_kernel<float>, dim3(sp_blocks), dim3(SP_COLS * SP_ROWS), 0, st, part, (int)nblocks,
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
  hipLaunchKernelGGL(rs_pass1, grid, block, 0, st, Z, (long)M, thr, ws);
  hipLaunchKernelGGL(rs_pass2, grid, block, 0, st, Z, (long)M, k, ws);
  hipLaunchKernelGGL(rs_pass3, grid, block, 0, st, Z, (long)M, k, ws);
  hipLaunchKernelGGL(rs_tail, grid, block, 0, st, Z, (long)M, k, ws);
  hipLaunchKernelGGL(rs_final, dim3(1), dim3(RATO_BLOCK), 0, st, (long)M, alpha, k, var_is_max, (int)nb, ws, out);
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

// The records of K rows of Z [K][ldz] (the Z of K control sequences on one sample batch: rato_*_eval_batch) -> out [K][N]:
// one launch of K workgroups while a row fits the one-workgroup form (M <= 12,288), K stream-ordered rato_risk_stats calls
// on the one workspace otherwise.
extern "C" int rato_risk_stats_batch(const float* Z, int64_t M, int64_t ldz, int32_t K, double alpha, float thr, void* workspace,
                                     size_t workspace_bytes, double* out, void* stream) {
  RATO_CLEAR_ERROR();
  if (!Z || !out || M <= 0 || ldz < M || K < 1 || !(alpha > 0.0) || !(alpha <= 1.0)) return RATO_EINVAL;
  if (M <= RS_SMALL_MAX && K <= 65535) {
    unsigned k;
    int var_is_max;
    stats_rank(M, alpha, k, var_is_max);
    hipLaunchKernelGGL(rs_small_batch, dim3((unsigned)K), dim3(RS1_T), 0, rato::as_stream(stream), Z, (long)M, (long)ldz, alpha, k,
                       var_is_max, thr, out);
    RATO_LAUNCH_CHECK();
    return RATO_OK;
  }
  for (int32_t i = 0; i < K; ++i) {
    const int rc = rato_risk_stats(Z + (size_t)i * ldz, M, alpha, thr, workspace, workspace_bytes, out + (size_t)i * RATO_N_STATS, stream);
    if (rc != RATO_OK) return rc;
  }
  return RATO_OK;
}

// Recovery path of the one-launch forms: a call that ended in NaN statistics (its wait for the workgroups of its own launch
// gave up -- a chip some other stream owned for seconds -- or it found the workspace unclean) is repeated with the
// workspace re-initialised on the stream and the selection as five stream-ordered launches that wait for nothing.
extern "C" int rato_risk_stats_recover(const float* Z, int64_t M, double alpha, float thr, void* workspace,
                                       size_t workspace_bytes, double* out, void* stream) {
  const int rc = rato_risk_stats_init(workspace, workspace_bytes, stream);
  if (rc != RATO_OK) return rc;
  return risk_stats_impl(Z, M, alpha, thr, workspace, workspace_bytes, out, nullptr, 0, 0, 1.0, nullptr, stream, true);
}

extern "C" int rato_sums_and_risk_stats(const float* part, int32_t nblocks, int32_t ncols, double scale,
                                        double* sums_out, const float* Z, int64_t M, double alpha, float thr,
                                        void* workspace, size_t workspace_bytes, double* out, void* stream) {
  if (!part) return RATO_EINVAL;
  return risk_stats_impl(Z, M, alpha, thr, workspace, workspace_bytes, out, part, nblocks, ncols, scale, sums_out,
                         stream);
}
