// The exact selection (VaR / CVaR statistics) as DEVICE functions, shared by stats.hip (the stand-alone launches) and by
// the row-parallel linearize kernels of drone.hip / driving.hip, whose launch can carry the statistics of the Z it
// produces in a few extra workgroups (params.stats_workspace, rato_saa.h).  Moved here from stats.hip in round 4; the
// arithmetic is unchanged.
#ifndef RATO_SELECT_H
#define RATO_SELECT_H

#include <math.h>

#include "rato_common.h"

namespace rato_sel {

constexpr unsigned long long RS_COOP_WAIT_TICKS = 10ull * 100000000ull;   // 10 s of s_memrealtime (100 MHz)
constexpr int B1 = 2048, B2 = 2048, B3 = 1024;  // 11 + 11 + 10 key bits
constexpr int RS_MAX_BLOCKS = 1024;

struct Workspace {
  unsigned hist1[B1];
  unsigned hist2[B2];
  unsigned hist3[B3];
  double blockpart[RS_MAX_BLOCKS][6];  // sum Z, count(Z<=thr), max Z, tail sum, count(Z>t), count(Z==t)
  float tstar;
  unsigned nblocks;
  unsigned magic;   // set by rato_risk_stats_init: the histograms start zeroed and every call leaves them zeroed
  unsigned ticket;  // rs_coop: completion tickets (0 between calls)
  // Signal words of a launch that carries its own statistics (params.stats_workspace): the producer workgroups of a
  // row-parallel linearize kernel count the tiles whose Z has landed in sig[0] and raise sig[2] (z_ready) with the last
  // one; the statistics workgroups at the end of the grid wait for it and lower it again.  All zero between launches.
  unsigned sig[8];
};
constexpr int SIG_Z_COUNT = 0, SIG_Z_READY = 2;
constexpr unsigned RS_MAGIC = 0x52A70517u;

// order-preserving map float -> uint32 (ascending)
// order-preserving key of a float.  -0.0 takes the key of +0.0, so that comparing keys (the one-launch forms) and
// comparing values (rs_tail, the tail-row kernels of cvar.hip: m > t, m == t) give the same counts and tail weights
// when the threshold is a zero of either sign.  (NaN has no place in a sorted order: callers check finiteness --
// rato_count_nonfinite -- before the statistics mean anything.)
__device__ __forceinline__ unsigned key_of(float f) {
  unsigned u = __float_as_uint(f);
  if (u == 0x80000000u) u = 0u;
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
  const unsigned incl = rato::wave_scan_dpp(tot);  // inclusive scan across the wave (DPP, no LDS round trips)
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

template <int NB, int NT = RATO_BLOCK>
__device__ void flush_hist(unsigned* lds_hist, unsigned* __restrict__ ghist) {
  __syncthreads();
  for (int i = threadIdx.x; i < NB; i += NT) {
    const unsigned c = lds_hist[i];
    if (c) atomicAdd(&ghist[i], c);
  }
}


// Companion launches: every workgroup waits (thread 0 polls, bounded by the clock like find_bin_coop) until the
// producer has raised `flag`; -> false when the wait expired (the caller then reports NaN statistics).
__device__ bool wait_signal(unsigned* flag) {
  __shared__ unsigned s_ok;
  if (threadIdx.x == 0) {
    const unsigned long long t_start = wall_clock64();
    unsigned ok = 1;
    // RELAXED polls, ONE acquire at the end: an acquire load at agent scope invalidates this XCD's L2 every time -- 62
    // workgroups polling that way beside the producers took the C5 shard from 194 to 269 us
    for (unsigned tries = 0; __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u; ++tries) {
      if ((tries & 255u) == 255u && wall_clock64() - t_start > RS_COOP_WAIT_TICKS) {
        ok = 0;
        break;
      }
      __builtin_amdgcn_s_sleep(32);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    s_ok = ok;
  }
  __syncthreads();
  const bool ok = s_ok != 0;
  __syncthreads();
  return ok;
}


constexpr long RS_SMALL_MAX = 12 * 1024;   // crossover with the cooperative form: 14.0 vs 14.7 us at M = 1e4, 19.1 vs 15.6 us at M = 2e4
constexpr int RS_COOP_KEYS = 16;                                  // keys per thread (registers)
constexpr int RS_COOP_MAX_WG = 64;

// find_bin on a histogram that OTHER workgroups of this launch are still adding to: device-scope loads, repeated until
// the counters add up to `expected` (the number of keys this pass distributes: M, then the count of the chosen bin).
// Every counter only grows during a pass, so total == expected means every add has landed -- the histogram itself is
// the barrier, with no arrival counter and no fence (one memory round trip per try instead of three per barrier).
// Two ways out without a result, both loud (NaN statistics, workspace un-tagged), neither a hang:
//   * the counters EXCEED `expected`: they only grow, so the workspace was not clean when the launch started -- at once;
//   * the counters stay short for RS_COOP_WAIT_S seconds of the constant 100 MHz clock.  A launch whose workgroups are
//     not all resident yet (another stream holds the CUs) is NOT a failure: its waiting workgroups keep polling until
//     the rest has been scheduled and has added its keys, however long the other stream's kernel takes (the first
//     version gave up after 2^18 polls ~ 0.5 s and poisoned the statistics of a merely delayed launch).
template <int NB, int NT>
__device__ bool find_bin_coop(const unsigned* __restrict__ hist, unsigned k, unsigned expected, unsigned& bin,
                              unsigned& krem, unsigned& bincount) {
  constexpr int PER = NB / NT;
  static_assert(PER >= 1 && PER * NT == NB, "bins must divide evenly over the threads");
  __shared__ unsigned wsum[NT / RATO_WAVE];
  __shared__ unsigned res[3];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const unsigned long long t_start = wall_clock64();
  for (unsigned tries = 0;; ++tries) {
    unsigned local[PER], tot = 0;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      local[i] = __hip_atomic_load(const_cast<unsigned*>(hist) + tid * PER + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      tot += local[i];
    }
    const unsigned incl = rato::wave_scan_dpp(tot);
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    unsigned base = 0, grand = 0;
    for (int w = 0; w < NT / RATO_WAVE; ++w) {
      const unsigned v = wsum[w];
      if (w < wave) base += v;
      grand += v;
    }
    if (grand == expected) {        // uniform over the workgroup (every thread summed the same LDS words)
      const unsigned excl = base + incl - tot;
      if (k >= excl && k < excl + tot) {
        unsigned run = excl;
#pragma unroll
        for (int i = 0; i < PER; ++i) {
          if (k >= run && k < run + local[i]) {
            res[0] = tid * PER + i;
            res[1] = k - run;
            res[2] = local[i];
          }
          run += local[i];
        }
      }
      __syncthreads();
      bin = res[0];
      krem = res[1];
      bincount = res[2];
      __syncthreads();
      return true;
    }
    // uniform over the workgroup: every thread summed the same LDS words / thread 0's clock is published through LDS
    if (grand > expected) return false;                                   // unclean workspace
    if ((tries & 1023u) == 1023u) {
      if (tid == 0) res[0] = (wall_clock64() - t_start > RS_COOP_WAIT_TICKS) ? 1u : 0u;
      __syncthreads();
      const unsigned expired = res[0];
      __syncthreads();
      if (expired) return false;
    }
    __syncthreads();                // wsum is rewritten by the next try
    if (tries < 64) __builtin_amdgcn_s_sleep(4); else __builtin_amdgcn_s_sleep(32);
  }
}


// Unsigned wave minimum / maximum through DPP (the tree of wave_max_dpp; result in every lane).
__device__ __forceinline__ unsigned wave_min_u32_dpp(unsigned a) {
  const int top = (int)0xffffffffu;
  a = min(a, (unsigned)__builtin_amdgcn_update_dpp(top, (int)a, 0x111, 0xf, 0xf, false));
  a = min(a, (unsigned)__builtin_amdgcn_update_dpp(top, (int)a, 0x112, 0xf, 0xf, false));
  a = min(a, (unsigned)__builtin_amdgcn_update_dpp(top, (int)a, 0x114, 0xf, 0xf, false));
  a = min(a, (unsigned)__builtin_amdgcn_update_dpp(top, (int)a, 0x118, 0xf, 0xf, false));
  a = min(a, (unsigned)__builtin_amdgcn_update_dpp(top, (int)a, 0x142, 0xa, 0xf, false));
  a = min(a, (unsigned)__builtin_amdgcn_update_dpp(top, (int)a, 0x143, 0xc, 0xf, false));
  return (unsigned)__builtin_amdgcn_readlane((int)a, 63);
}
__device__ __forceinline__ unsigned wave_max_u32_dpp(unsigned a) {
  a = max(a, (unsigned)__builtin_amdgcn_update_dpp(0, (int)a, 0x111, 0xf, 0xf, false));
  a = max(a, (unsigned)__builtin_amdgcn_update_dpp(0, (int)a, 0x112, 0xf, 0xf, false));
  a = max(a, (unsigned)__builtin_amdgcn_update_dpp(0, (int)a, 0x114, 0xf, 0xf, false));
  a = max(a, (unsigned)__builtin_amdgcn_update_dpp(0, (int)a, 0x118, 0xf, 0xf, false));
  a = max(a, (unsigned)__builtin_amdgcn_update_dpp(0, (int)a, 0x142, 0xa, 0xf, false));
  a = max(a, (unsigned)__builtin_amdgcn_update_dpp(0, (int)a, 0x143, 0xc, 0xf, false));
  return (unsigned)__builtin_amdgcn_readlane((int)a, 63);
}

constexpr int RS_CAND_MAX = 256;   // candidates of a chosen bin that are ranked directly (pairwise) instead of binned again

// ONE workgroup of NT threads: the whole exact selection with the keys in registers (M <= RS_SMALL_MAX).  The body of
// stats.hip's rs_small (NT = 1024) and of the statistics workgroup that rides at the end of a row-parallel linearize /
// tiled eval launch (NT = 512 / 256).  h: B1 words, red: 5 * NT/64 doubles, redmax: NT/64 floats of LDS.  sig != NULL:
// wait until the producer of Z has raised z_ready, lower it again.
//
// Round 5: RANGE-NORMALISED bins.  Constraint values cluster (a binade holds 4 of the 2048 bins of the key's top 11
// bits), so the first pass of a plain radix select filtered almost nothing, its LDS atomics serialised on a handful of
// addresses, and all three passes (each: zero, histogram, a 2048-bin scan behind three barriers) were always paid:
// ~3 us each.  Here the workgroup first reduces the minimum and maximum key, and a pass bins (key - lo) >> sh over the
// range that is actually occupied: 2048 bins across [min, max] hold ~M / 2048 keys each, the atomics spread, and after
// ONE pass the bin that holds the wanted rank has a handful of members -- which are gathered into LDS and ranked
// pairwise (<= RS_CAND_MAX of them; a bin that is still crowded -- ties, extreme clustering -- is binned again on its
// next 11 bits, down to single keys).  Same result as before to the bit (the k-th smallest key is what it is).
template <int NT>
__device__ void rs_small_body(const float* __restrict__ Z, long M, double alpha, unsigned k, int var_is_max, float thr,
                              double* __restrict__ out, unsigned* sig, unsigned* h, double* red, float* redmax,
                              unsigned* magic = nullptr) {
  constexpr int KEYS = (int)(RS_SMALL_MAX / NT), NW = NT / RATO_WAVE;
  static_assert(RS_CAND_MAX + 8 <= B1, "the candidate list reuses the histogram's LDS");
  if (sig) {
    // (magic: the workspace's tag -- a workspace that was never initialised, or that an earlier launch un-tagged, holds
    //  no trustworthy flag: fail loudly instead of waiting on it)
    const bool tagged = !magic || *magic == RS_MAGIC;
    const bool ok = tagged && wait_signal(sig + SIG_Z_READY);
    if (!ok) {
      // The wait expired (or the workspace is not tagged).  The flag is NOT lowered here: the producers are still
      // running and will raise it later -- lowered now, it would stay up and the NEXT launch would select on a Z that is
      // not written yet.  The workspace is un-tagged instead: the next call on it reports NaN until it is re-initialised.
      if (threadIdx.x < RATO_N_STATS) out[threadIdx.x] = __longlong_as_double(0x7ff8000000000000LL);
      if (magic && threadIdx.x == 0) *magic = 0u;
      return;
    }
    if (threadIdx.x == 0) __hip_atomic_store(sig + SIG_Z_READY, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  const int tid = threadIdx.x;
  const int n = (int)M;
#ifdef RATO_RS_DIAG
  // diagnostic stamps (100 MHz wall clock) behind the record: out[16 + k], k = 0 start, 1 keys loaded + range known,
  // 6 rank key found, 7 tail terms accumulated, 8 record written.  Slots 2..5 (the three fixed passes of the round-4
  // form) are NOT written by the range-normalised form: readers must not expect them.
  if (tid == 0) out[16 + 0] = (double)wall_clock64();
#endif
  // every load of the thread in flight before anything else (Z comes from HBM: its producer's L2 was written back)
  unsigned key[KEYS];
  double sum = 0.0, cnt = 0.0;
  float mx = -INFINITY;
  unsigned kmin = 0xffffffffu, kmax = 0u;
  {
    float z[KEYS];
#pragma unroll
    for (int u = 0; u < KEYS; ++u) {
      const int i = tid + u * NT;
      z[u] = (i < n) ? Z[i] : 0.0f;
    }
#pragma unroll
    for (int u = 0; u < KEYS; ++u) {
      const int i = tid + u * NT;
      key[u] = key_of(z[u]);
      if (i < n) {
        sum += (double)z[u];
        cnt += (z[u] <= thr) ? 1.0 : 0.0;
        mx = fmaxf(mx, z[u]);
        kmin = min(kmin, key[u]);
        kmax = max(kmax, key[u]);
      }
    }
  }
  unsigned* mm = reinterpret_cast<unsigned*>(red);   // [2 NW]: the waves' minima | maxima (red is free until the end)
  kmin = wave_min_u32_dpp(kmin);
  kmax = wave_max_u32_dpp(kmax);
  if ((tid & 63) == 0) {
    mm[tid >> 6] = kmin;
    mm[NW + (tid >> 6)] = kmax;
  }
  __syncthreads();
#pragma unroll
  for (int w = 0; w < NW; ++w) {
    kmin = min(kmin, mm[w]);
    kmax = max(kmax, mm[NW + w]);
  }
#ifdef RATO_RS_DIAG
  if (tid == 0) out[16 + 1] = (double)wall_clock64();
#endif
  // lo: lower end of the key range still in play; sh: bits below the 11 this pass bins on; krem: wanted rank inside it
  unsigned lo = kmin, krem = k;
  const unsigned range = kmax - kmin;
  int sh = (range == 0u) ? 0 : (32 - __clz((int)range)) - 11;
  if (sh < 0) sh = 0;
  unsigned width = 0u;         // candidates are the keys with key - lo < width (0: every key, first pass)
  unsigned tkey = 0u;
  for (int level = 0;; ++level) {
    for (int i = tid; i < B1; i += NT) h[i] = 0;
    __syncthreads();
    {
      unsigned run_bin = 0xffffffffu, run_cnt = 0;   // runs of equal bins of a thread's own elements -> one LDS atomic
#pragma unroll
      for (int u = 0; u < KEYS; ++u) {
        const int i = tid + u * NT;
        const unsigned rel = key[u] - lo;
        if (i < n && (level == 0 || (key[u] >= lo && rel < width))) {   // (>= lo: a key below lo wraps, and with a key range near 2^32 -- NaNs of both signs -- could pass)
          const unsigned bin = rel >> sh;
          if (bin != run_bin) {
            if (run_cnt) atomicAdd(&h[run_bin], run_cnt);
            run_bin = bin;
            run_cnt = 0;
          }
          ++run_cnt;
        }
      }
      if (run_cnt) atomicAdd(&h[run_bin], run_cnt);
    }
    __syncthreads();
    unsigned b, kr;
    find_bin<B1, NT>(h, krem, b, kr);
    const unsigned c = h[b];     // members of the chosen bin (uniform: every thread reads the same word)
    lo += b << sh;
    krem = kr;
    width = 1u << sh;            // (sh <= 21)
    if (sh == 0) {               // a bin is one key
      tkey = lo;
      break;
    }
    __syncthreads();             // everybody has read h[b]: the histogram's LDS becomes the candidate list
    if (c <= (unsigned)RS_CAND_MAX) {
      unsigned* cand = h;        // [c] keys | h[B1 - 2]: fill counter | h[B1 - 1]: the selected key
      if (tid == 0) h[B1 - 2] = 0u;
      __syncthreads();
#pragma unroll
      for (int u = 0; u < KEYS; ++u) {
        const int i = tid + u * NT;
        if (i < n && key[u] >= lo && key[u] - lo < width) cand[atomicAdd(&h[B1 - 2], 1u)] = key[u];
      }
      __syncthreads();
      if (tid < (int)c) {        // the candidate with  #{< x} <= krem < #{<= x}  (equal keys all qualify, and agree)
        const unsigned x = cand[tid];
        unsigned lt = 0, le = 0;
        for (unsigned j = 0; j < c; ++j) {
          const unsigned y = cand[j];
          lt += (y < x) ? 1u : 0u;
          le += (y <= x) ? 1u : 0u;
        }
        if (lt <= krem && krem < le) h[B1 - 1] = x;
      }
      __syncthreads();
      tkey = h[B1 - 1];
      break;
    }
    sh = (sh > 11) ? sh - 11 : 0;   // still crowded: its next 11 bits
  }
#ifdef RATO_RS_DIAG
  if (tid == 0) out[16 + 6] = (double)wall_clock64();
#endif
  const float t = value_of(tkey);
  double tail = 0.0, ngt = 0.0, neq = 0.0;
#pragma unroll
  for (int u = 0; u < KEYS; ++u) {
    const int i = tid + u * NT;
    if (i < n) {
      const float z = value_of(key[u]);
      tail += (key[u] > tkey) ? ((double)z - (double)t) : 0.0;
      ngt += (key[u] > tkey) ? 1.0 : 0.0;
      neq += (key[u] == tkey) ? 1.0 : 0.0;
    }
  }
#ifdef RATO_RS_DIAG
  if (tid == 0) out[16 + 7] = (double)wall_clock64();
#endif
  // one barrier for all six block reductions; fixed order: DPP tree inside a wave, then the 16 wave totals through
  // one DPP row of wave 0 (a serial fold by one thread + shuffle trees cost 3.8 us of the 12-14 us of this kernel)
  sum = rato::wave_sum_dpp(sum);
  cnt = rato::wave_sum_dpp(cnt);
  tail = rato::wave_sum_dpp(tail);
  ngt = rato::wave_sum_dpp(ngt);
  neq = rato::wave_sum_dpp(neq);
  mx = rato::wave_max_dpp(mx);
  __syncthreads();               // (red held the key range of the waves until every thread had read it)
  if ((tid & 63) == 0) {
    const int w = tid >> 6;
    red[0 * NW + w] = sum; red[1 * NW + w] = cnt; red[2 * NW + w] = tail; red[3 * NW + w] = ngt; red[4 * NW + w] = neq;
    redmax[w] = mx;
  }
  __syncthreads();
  if (tid < RATO_WAVE) {
    static_assert(NT / RATO_WAVE <= 16, "one DPP row folds the wave totals");
    const bool in = tid < NT / RATO_WAVE;
    const double S = rato::row16_sum_dpp(in ? red[0 * NW + tid] : 0.0), C = rato::row16_sum_dpp(in ? red[1 * NW + tid] : 0.0);
    const double T = rato::row16_sum_dpp(in ? red[2 * NW + tid] : 0.0), NG = rato::row16_sum_dpp(in ? red[3 * NW + tid] : 0.0);
    const double NE = rato::row16_sum_dpp(in ? red[4 * NW + tid] : 0.0);
    const double m = (double)rato::row16_max_dpp(in ? redmax[tid] : -INFINITY);
    if (tid == 15) {
      out[0] = var_is_max ? m : (double)t;
      out[1] = (double)t + (T / (double)M) / alpha;
      out[2] = C / (double)M;                   // true divisions: the fraction is compared bit for bit with np.mean
      out[3] = S / (double)M;
      out[4] = m;
      out[5] = C;
      out[6] = T;
      out[7] = (double)k;
      out[8] = NG;
      out[9] = NE;
      out[10] = (double)t;
#ifdef RATO_RS_DIAG
      out[16 + 8] = (double)wall_clock64();
#endif
    }
  }
}


// G workgroups of NT threads, keys in registers, global histograms that the workgroups wait on (see stats.hip: rs_coop is
// this body with NT = 1024; the statistics workgroups at the end of a row-parallel linearize launch run it with NT = 512).
// bid: this workgroup's index among the G; companion: wait for the producer's z_ready first (lowered by the last one).
template <int NT>
__device__ void rs_coop_body(const float* __restrict__ Z, long M, double alpha, unsigned k, int var_is_max, float thr, int G,
                             Workspace* __restrict__ ws, double* __restrict__ out, int bid, int companion, unsigned* h,
                             double* red) {
  constexpr int NW = NT / RATO_WAVE;
  const int tid = threadIdx.x;
  if (ws->magic != RS_MAGIC) {  // never initialised: the barrier counter is garbage -> do not wait on it, fail loudly
    if (bid == 0 && tid < RATO_N_STATS) out[tid] = __longlong_as_double(0x7ff8000000000000LL);
    return;
  }
  // companion launch: Z is being produced beside this kernel (the flag is lowered by the workgroup that finishes last;
  // a wait that expires leaves the histogram protocol short of keys, which ends in NaN statistics below)
  const bool z_ok = companion ? wait_signal(ws->sig + SIG_Z_READY) : true;
  __shared__ unsigned last_flag;
  const int n = (int)M;
  unsigned key[RS_COOP_KEYS];
  double sum = 0.0, cnt = 0.0;
  float mx = -INFINITY;
  {
    float z[RS_COOP_KEYS];
#pragma unroll
    for (int u = 0; u < RS_COOP_KEYS; ++u) {                  // element (u, workgroup, thread): coalesced, all in flight
      const int i = (u * G + bid) * NT + tid;
      z[u] = (i < n) ? Z[i] : 0.0f;
    }
    for (int i = tid; i < B1; i += NT) h[i] = 0;           // while the loads are in flight
    __syncthreads();
    unsigned run_bin = 0xffffffffu, run_cnt = 0;             // runs of equal bins -> one LDS atomic (values cluster)
#pragma unroll
    for (int u = 0; u < RS_COOP_KEYS; ++u) {
      const int i = (u * G + bid) * NT + tid;
      key[u] = key_of(z[u]);
      if (i < n) {
        const unsigned bin = key[u] >> 21;
        if (bin != run_bin) {
          if (run_cnt) atomicAdd(&h[run_bin], run_cnt);
          run_bin = bin;
          run_cnt = 0;
        }
        ++run_cnt;
        sum += (double)z[u];
        cnt += (z[u] <= thr) ? 1.0 : 0.0;
        mx = fmaxf(mx, z[u]);
      }
    }
    if (run_cnt) atomicAdd(&h[run_bin], run_cnt);
  }
  flush_hist<B1, NT>(h, ws->hist1);
  unsigned b1 = 0, k1 = 0, c1 = 0, b2 = 0, k2 = 0, c2 = 0, b3 = 0, k3 = 0, c3 = 0;   // (a failed pass leaves them unset)
  bool ok = z_ok && find_bin_coop<B1, NT>(ws->hist1, k, (unsigned)n, b1, k1, c1);
  for (int i = tid; i < B2; i += NT) h[i] = 0;
  __syncthreads();
#pragma unroll
  for (int u = 0; u < RS_COOP_KEYS; ++u) {
    const int i = (u * G + bid) * NT + tid;
    if (i < n && (key[u] >> 21) == b1) atomicAdd(&h[(key[u] >> 10) & (B2 - 1)], 1u);
  }
  flush_hist<B2, NT>(h, ws->hist2);
  ok = ok && find_bin_coop<B2, NT>(ws->hist2, k1, c1, b2, k2, c2);
  for (int i = tid; i < B3; i += NT) h[i] = 0;
  __syncthreads();
  const unsigned prefix = (b1 << 11) | b2;
#pragma unroll
  for (int u = 0; u < RS_COOP_KEYS; ++u) {
    const int i = (u * G + bid) * NT + tid;
    if (i < n && (key[u] >> 10) == prefix) atomicAdd(&h[key[u] & (B3 - 1)], 1u);
  }
  flush_hist<B3, NT>(h, ws->hist3);
  ok = ok && find_bin_coop<B3, NT>(ws->hist3, k2, c2, b3, k3, c3);
  if (!ok) {   // the histograms never added up (unclean workspace) or a wait expired.  NaN out, un-tag the workspace.
    // (z_ready is left alone: after an expired wait the producers are still running and will raise it -- lowered here it
    //  would stay up for the next launch; an un-tagged workspace makes that launch fail loudly until it is re-initialised)
    if (tid < RATO_N_STATS) out[tid] = __longlong_as_double(0x7ff8000000000000LL);
    if (tid == 0) ws->magic = 0;
    return;
  }
  const unsigned tkey = (b1 << 21) | (b2 << 10) | b3;
  const float t = value_of(tkey);
  double tail = 0.0, ngt = 0.0, neq = 0.0;
#pragma unroll
  for (int u = 0; u < RS_COOP_KEYS; ++u) {
    const int i = (u * G + bid) * NT + tid;
    if (i < n) {
      const float z = value_of(key[u]);
      tail += (key[u] > tkey) ? ((double)z - (double)t) : 0.0;
      ngt += (key[u] > tkey) ? 1.0 : 0.0;
      neq += (key[u] == tkey) ? 1.0 : 0.0;
    }
  }
  sum = rato::wave_sum_dpp(sum);
  cnt = rato::wave_sum_dpp(cnt);
  tail = rato::wave_sum_dpp(tail);
  ngt = rato::wave_sum_dpp(ngt);
  neq = rato::wave_sum_dpp(neq);
  mx = rato::wave_max_dpp(mx);
  if ((tid & 63) == 0) {
    const int w = tid >> 6;
    red[0 * NW + w] = sum; red[1 * NW + w] = cnt; red[2 * NW + w] = (double)mx; red[3 * NW + w] = tail; red[4 * NW + w] = ngt; red[5 * NW + w] = neq;
  }
  __syncthreads();
  if (tid < RATO_WAVE) {   // the 16 wave totals through one DPP row (fixed order), result in lane 15
    const bool in = tid < NT / RATO_WAVE;
    const double S = rato::row16_sum_dpp(in ? red[0 * NW + tid] : 0.0), C = rato::row16_sum_dpp(in ? red[1 * NW + tid] : 0.0);
    const double T = rato::row16_sum_dpp(in ? red[3 * NW + tid] : 0.0), NG = rato::row16_sum_dpp(in ? red[4 * NW + tid] : 0.0);
    const double NE = rato::row16_sum_dpp(in ? red[5 * NW + tid] : 0.0);
    const double m = (double)rato::row16_max_dpp(in ? (float)red[2 * NW + tid] : -INFINITY);
    if (tid == 15) {
      double* bp = ws->blockpart[bid];
      bp[0] = S; bp[1] = C; bp[2] = m; bp[3] = T; bp[4] = NG; bp[5] = NE;
      // release the partials, take a completion ticket; the last workgroup acquires everyone's partials
      const unsigned tk = __hip_atomic_fetch_add(&ws->ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
      last_flag = (tk == (unsigned)G - 1u);
    }
  }
  __syncthreads();
  if (!last_flag) return;
  // every workgroup has read hist3 before taking its ticket: zero the histograms and the counters for the next call
  for (int i = tid; i < B1 + B2 + B3; i += NT) ws->hist1[i] = 0;   // hist1..3 are contiguous
  if (tid == 0) {
    ws->ticket = 0;
    if (companion) __hip_atomic_store(ws->sig + SIG_Z_READY, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (tid >= RATO_WAVE) return;
  double s = 0, c = 0, m = -INFINITY, tl = 0, g = 0, e = 0;
  if (tid < G) {   // G <= 64: one partial per lane, folded by the fixed shuffle tree (independent of who came last)
    const double* bp = ws->blockpart[tid];
    s = __hip_atomic_load(bp + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    c = __hip_atomic_load(bp + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    m = __hip_atomic_load(bp + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    tl = __hip_atomic_load(bp + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    g = __hip_atomic_load(bp + 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    e = __hip_atomic_load(bp + 5, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  s = rato::wave_sum_dpp(s);
  c = rato::wave_sum_dpp(c);
  tl = rato::wave_sum_dpp(tl);
  g = rato::wave_sum_dpp(g);
  e = rato::wave_sum_dpp(e);
  m = (double)rato::wave_max_dpp((float)m);      // maxima of fp32 values: exact in float
  if (tid != 0) return;
  out[0] = var_is_max ? m : (double)t;
  out[1] = (double)t + (tl / (double)M) / alpha;
  out[2] = c / (double)M;
  out[3] = s / (double)M;
  out[4] = m;
  out[5] = c;
  out[6] = tl;
  out[7] = (double)k;
  out[8] = g;
  out[9] = e;
  out[10] = (double)t;
}


// The statistics of the Z a row-parallel linearize launch produces, computed by G (or one) EXTRA workgroups at the end of
// that launch's grid (blockIdx >= n_prod): what the launcher hands the kernel.  ws == NULL: no statistics ride along.
// LAST in the grid, never first: with a fixed tile per workgroup the statistics must not hold a slot a producer is still
// waiting for.  Used for SMALL batches only (every workgroup of the launch resident at once, the memory system far from
// saturated).  Beside a store-saturated producer the selection's chain of dependent global round trips (histogram
// atomics, polls, Z re-read) is several times SLOWER than behind it: statistics workgroups resident from the start of a
// queue-mode launch took the C5 shard from 194 to 250-269 us and the metric configuration from 579 to 582
// (profiles/r04_j_stats_in_launch.txt) -- for those launches the same call issues rato_risk_stats behind the kernel.
struct StatsTail {
  Workspace* ws;
  double* out;
  double alpha;
  unsigned k;
  int var_is_max;
  float thr;
  int G;        // 0: one workgroup (rs_small_body); > 0: that many cooperating workgroups (rs_coop_body)
  int n_prod;   // workgroups of the producer proper
  __device__ bool is_stats(int b) const { return ws && b >= n_prod; }
};

// ascending 0-based rank of sort(Z)[M - floor(alpha M) - 1] (drone_main_plot.py:649-651); alpha M == M wraps to the maximum
inline void stats_rank(int64_t M, double alpha, unsigned& k, int& var_is_max) {
  long xth = (long)floor(alpha * (double)M);
  long kk = (long)M - xth - 1;
  var_is_max = 0;
  if (kk < 0) {
    kk = 0;
    var_is_max = 1;
  }
  k = (unsigned)kk;
}

// the extra workgroups of a launch with NT threads per workgroup for M samples: -1 = M is beyond the one-launch forms
template <int NT>
inline int stats_tail_workgroups(int64_t M, int& G) {
  if (M <= RS_SMALL_MAX) {
    G = 0;
    return 1;
  }
  long g = (M + (long)NT * 4 - 1) / ((long)NT * 4);           // 4 keys per thread ...
  if (g > RS_COOP_MAX_WG) g = RS_COOP_MAX_WG;                // ... up to RS_COOP_KEYS on 64 workgroups
  if ((long)g * NT * RS_COOP_KEYS < M) return -1;
  G = (int)g;
  return (int)g;
}

// the statistics workgroups themselves (dynamic LDS of the launch: >= rs_body_lds_bytes<NT>())
template <int NT>
__device__ __forceinline__ void stats_tail_run(const StatsTail& t, const float* __restrict__ Z, long M, unsigned char* lds) {
  unsigned* h = reinterpret_cast<unsigned*>(lds);
  double* red = reinterpret_cast<double*>(h + B1);
  float* redmax = reinterpret_cast<float*>(red + 6 * (NT / RATO_WAVE));
  if (t.G == 0) rs_small_body<NT>(Z, M, t.alpha, t.k, t.var_is_max, t.thr, t.out, t.ws->sig, h, red, redmax, &t.ws->magic);
  else rs_coop_body<NT>(Z, M, t.alpha, t.k, t.var_is_max, t.thr, t.G, t.ws, t.out, (int)blockIdx.x - t.n_prod, 1, h, red);
}

// LDS (bytes) the two bodies need from their caller: h (B1 words) | red (6 * NT/64 doubles) | redmax (NT/64 floats)
template <int NT>
constexpr size_t rs_body_lds_bytes() { return sizeof(unsigned) * B1 + sizeof(double) * 6 * (NT / RATO_WAVE) + sizeof(float) * (NT / RATO_WAVE); }

}  // namespace rato_sel
#endif
