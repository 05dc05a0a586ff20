// Shared device/host helpers for librato_saa.so (gfx950 only: wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <atomic>
#include <chrono>
#include <mutex>

#include "rato_saa.h"

// HARDWARE ASSUMPTIONS (gfx950 only; the device pass refuses any other target below):
//  * wave = 64 lanes, DPP row_bcast / row_shr semantics of GFX9;
//  * in-launch hand-offs between workgroups (rato_select.h: z_ready, the cooperative histograms; drone.hip / driving.hip:
//    LDS progress words) publish data with RELAXED agent-scope atomic stores + `s_waitcnt vmcnt(0)` + a relaxed counter:
//    this relies on agent-scope (sc1) stores being written through to the device's point of coherence before the wait
//    retires -- true on gfx942 / gfx950, outside the HSA memory model (a release fence per tile would write back the
//    XCD's whole L2 in the middle of the store stream: measured 56 -> 194 us).  The fenced alternative stays testable:
//    statistics behind the kernel (fused=False / large batches) use no hand-off at all;
//  * statistics workgroups sit at the END of the grid and the launcher only enables them while every workgroup of the
//    launch is resident at once, so that they cannot hold a slot a producer waits for (no reliance on dispatch order
//    beyond "all resident"); their waits are bounded by the wall clock and fail loudly (NaN record, un-tagged workspace);
//  * rato::store_streaming is inline assembly (`global_store_dword ... nt`, GFX94x/95x syntax): the compiler's waitcnt
//    insertion does not see it, so it is only used for write-once outputs that nothing in the same kernel reads back.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "librato_saa is written for gfx950 (MI355X) only: see the hardware assumptions in rato_common.h"
#endif

#define RATO_BLOCK 256           // 4 waves per workgroup (== RATO_TILE of rato_saa.h)
#define RATO_WAVE 64

// hipGetLastError() is sticky per host thread: clear anything left behind by an
// earlier, unrelated runtime call before launching, then check our own launches.
#define RATO_CLEAR_ERROR() (void)hipGetLastError()

#define RATO_LAUNCH_CHECK()                              \
  do {                                                   \
    hipError_t e__ = hipGetLastError();                  \
    if (e__ != hipSuccess) return RATO_EHIP - (int)e__;  \
  } while (0)

static_assert(RATO_BLOCK == RATO_TILE, "one workgroup writes one Jacobian tile");

namespace rato {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, RATO_WAVE);
  return v;
}

// Wave-wide sum through DPP (no LDS crossbar round trips: ~10x lower latency than the
// ds_bpermute chain __shfl_xor compiles to).  The total is valid in EVERY lane (read back
// from lane 63 with readlane).  Fixed combination order, so results are deterministic.
__device__ __forceinline__ float wave_sum_dpp(float v) {
  int x = __float_as_int(v);
  // row_shr:1,2,3 then row_shr:4 + row_shr:8 style prefix inside each row of 16 lanes
  float a = v;
  a += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a), 0x111, 0xf, 0xf, true));  // row_shr:1
  a += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a), 0x112, 0xf, 0xf, true));  // row_shr:2
  a += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a), 0x114, 0xf, 0xf, true));  // row_shr:4
  a += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a), 0x118, 0xf, 0xf, true));  // row_shr:8
  // lane 15 of each row now holds the row total; combine the four rows
  a += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a), 0x142, 0xa, 0xf, true));  // row_bcast:15
  a += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a), 0x143, 0xc, 0xf, true));  // row_bcast:31
  (void)x;
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a), 63));
}

// The same DPP tree for doubles (both halves moved, zero filled where the source lane does not exist: +0.0) and
// for unsigned integers; wave_scan_dpp leaves the INCLUSIVE prefix sum in every lane (lane 63 = total).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_move(double a) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(a), CTRL, ROW_MASK, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(a), CTRL, ROW_MASK, 0xf, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum_dpp(double a) {
  a += dpp_move<0x111, 0xf>(a);
  a += dpp_move<0x112, 0xf>(a);
  a += dpp_move<0x114, 0xf>(a);
  a += dpp_move<0x118, 0xf>(a);
  a += dpp_move<0x142, 0xa>(a);
  a += dpp_move<0x143, 0xc>(a);
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(a), 63), __builtin_amdgcn_readlane(__double2loint(a), 63));
}
__device__ __forceinline__ unsigned wave_scan_dpp(unsigned a) {
  a += (unsigned)__builtin_amdgcn_update_dpp(0, (int)a, 0x111, 0xf, 0xf, true);
  a += (unsigned)__builtin_amdgcn_update_dpp(0, (int)a, 0x112, 0xf, 0xf, true);
  a += (unsigned)__builtin_amdgcn_update_dpp(0, (int)a, 0x114, 0xf, 0xf, true);
  a += (unsigned)__builtin_amdgcn_update_dpp(0, (int)a, 0x118, 0xf, 0xf, true);
  a += (unsigned)__builtin_amdgcn_update_dpp(0, (int)a, 0x142, 0xa, 0xf, true);
  a += (unsigned)__builtin_amdgcn_update_dpp(0, (int)a, 0x143, 0xc, 0xf, true);
  return a;
}
__device__ __forceinline__ float wave_max_dpp(float a) {   // max in every lane; out-of-range sources read as -inf
  const int ninf = (int)0xff800000u;
  a = fmaxf(a, __int_as_float(__builtin_amdgcn_update_dpp(ninf, __float_as_int(a), 0x111, 0xf, 0xf, false)));
  a = fmaxf(a, __int_as_float(__builtin_amdgcn_update_dpp(ninf, __float_as_int(a), 0x112, 0xf, 0xf, false)));
  a = fmaxf(a, __int_as_float(__builtin_amdgcn_update_dpp(ninf, __float_as_int(a), 0x114, 0xf, 0xf, false)));
  a = fmaxf(a, __int_as_float(__builtin_amdgcn_update_dpp(ninf, __float_as_int(a), 0x118, 0xf, 0xf, false)));
  a = fmaxf(a, __int_as_float(__builtin_amdgcn_update_dpp(ninf, __float_as_int(a), 0x142, 0xa, 0xf, false)));
  a = fmaxf(a, __int_as_float(__builtin_amdgcn_update_dpp(ninf, __float_as_int(a), 0x143, 0xc, 0xf, false)));
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a), 63));
}
// max over the 16 lanes of a DPP row, in lane 15 of the row
__device__ __forceinline__ float row16_max_dpp(float a) {
  const int ninf = (int)0xff800000u;
  a = fmaxf(a, __int_as_float(__builtin_amdgcn_update_dpp(ninf, __float_as_int(a), 0x111, 0xf, 0xf, false)));
  a = fmaxf(a, __int_as_float(__builtin_amdgcn_update_dpp(ninf, __float_as_int(a), 0x112, 0xf, 0xf, false)));
  a = fmaxf(a, __int_as_float(__builtin_amdgcn_update_dpp(ninf, __float_as_int(a), 0x114, 0xf, 0xf, false)));
  a = fmaxf(a, __int_as_float(__builtin_amdgcn_update_dpp(ninf, __float_as_int(a), 0x118, 0xf, 0xf, false)));
  return a;
}
// sum over the 16 lanes of a DPP row, total in lane 15 of the row (fixed order)
__device__ __forceinline__ double row16_sum_dpp(double a) {
  a += dpp_move<0x111, 0xf>(a);
  a += dpp_move<0x112, 0xf>(a);
  a += dpp_move<0x114, 0xf>(a);
  a += dpp_move<0x118, 0xf>(a);
  return a;
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, RATO_WAVE);
  return v;
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, RATO_WAVE));
  return v;
}

// Distance (in floats) between consecutive tiles of a packed, tile-blocked Jacobian whose tile holds `payload_floats`
// numbers: tiles of 1 MiB or more start on 2 MiB boundaries (the buffer itself must be 2 MiB aligned), smaller tiles
// are packed back to back.  Why: the row-parallel kernels keep one store stream per resident tile (512 at a time); with
// the streams a power-of-two distance apart the memory system's address hash spreads them evenly over the channels --
// store-only, 512 streams of 1.88 MB tiles: 5.1-5.2 TB/s at the natural stride, 5.7 at 2 MiB (tools/store_pattern5.hip).
// ONE rule, computed from the tile shape alone, used by every producer and consumer of the layout and exported as
// rato_packed_tile_stride() for whoever allocates the buffer.
#ifndef RATO_PACKED_ALIGN_BYTES
#define RATO_PACKED_ALIGN_BYTES (2u << 20)   // A/B builds only (tools/): another alignment
#endif
#ifndef RATO_PACKED_MIN_BYTES
#define RATO_PACKED_MIN_BYTES (1u << 20)     // A/B builds only: a huge value = tiles always back to back
#endif
__host__ __device__ __forceinline__ size_t packed_tile_stride(size_t payload_floats) {
  constexpr size_t PAGE = (size_t)RATO_PACKED_ALIGN_BYTES / sizeof(float);
  return payload_floats * sizeof(float) >= (size_t)RATO_PACKED_MIN_BYTES ? (payload_floats + PAGE - 1) / PAGE * PAGE
                                                                          : payload_floats;
}

// pair(t, s) = t(t-1)/2 + s for 0 <= s < t  (row-major causal packing)
__host__ __device__ __forceinline__ int pair_row_offset(int t) { return (t * (t - 1)) >> 1; }

// A streaming (non-temporal) 4-byte store at a constant byte offset from p.  As an instruction of its own: behind a
// run-time flag the optimiser merges __builtin_nontemporal_store with the ordinary store of the other branch and drops the
// hint.  (Used for write-once output streams far larger than the memory-side cache: drone.hip, driving.hip.)
template <int BYTE_OFFSET>
__device__ __forceinline__ void store_streaming(float* p, float v) {
  static_assert(BYTE_OFFSET >= 0 && BYTE_OFFSET < 4096, "immediate offset of a global store");
  asm volatile("global_store_dword %0, %1, off offset:%2 nt" : : "v"(p), "v"(v), "n"(BYTE_OFFSET) : "memory");
}

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

static inline int nblocks_for(int32_t M) { return (M + RATO_BLOCK - 1) / RATO_BLOCK; }

// Work queues of the dynamic launch forms (row-parallel linearize kernels, eval): each queue is two device words
// {next tile, workgroups gone}, zero at load, and every launch leaves its queue zeroed.  One pool per kernel file and
// per DEVICE (a __device__ array has its own copy -- and its own address -- on every device of the process):
//   * an eager launch takes the queue bound to its STREAM: launches on one stream are ordered and may share it,
//     launches on different streams never share one (up to RATO_QUEUES_EAGER streams; a further stream gets none and
//     its launches use the static form);
//   * a launch recorded into a hipGraph (the stream is capturing) takes a queue of its OWN from the rest of the pool:
//     torch.cuda.graph captures every graph on one shared side stream, so a stream-keyed queue would be shared by
//     graphs that are later replayed concurrently on different streams -- two launches on one queue skip tiles.  A
//     graph exec cannot run concurrently with itself, so one queue per captured launch is enough.  The queues of
//     destroyed graphs are not reclaimed (16 KB of device words hold 1984 of them); when the pool is used up
//     (RATO_QUEUES_TOTAL - RATO_QUEUES_EAGER captured launches in one process) further captured launches get none and
//     use the static form, which is said once on stderr.
// The CU count is cached per device here too (no device query inside a capture after the first, uncaptured call).
#define RATO_QUEUES_EAGER 64
#define RATO_QUEUES_TOTAL 2048
class TileQueuePool {
 public:
  // resolve(): device address of the CURRENT device's copy of the file's queue array (RATO_QUEUES_TOTAL * 2 words)
  unsigned* take(hipStream_t stream, unsigned* (*resolve)()) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDev) return nullptr;
    std::lock_guard<std::mutex> lock(mu_);
    PerDev& d = dev_[dev];
    if (!d.base) {
      d.base = resolve();
      if (!d.base) return nullptr;
    }
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (stream && hipStreamIsCapturing(stream, &cs) == hipSuccess && cs == hipStreamCaptureStatusActive) {
      if (d.next_captured >= RATO_QUEUES_TOTAL) {
        if (!d.warned) {   // once per device and kernel file: the launch is still correct, only on the static form
          d.warned = true;
          fprintf(stderr, "librato_saa: work-queue pool used up (%d captured launches on device %d): further captured "
                          "launches use the static launch form\n", RATO_QUEUES_TOTAL - RATO_QUEUES_EAGER, dev);
        }
        return nullptr;
      }
      return d.base + 2 * d.next_captured++;
    }
    for (int i = 0; i < d.used; ++i)
      if (d.owner[i] == stream) return d.base + 2 * i;
    if (d.used == RATO_QUEUES_EAGER) return nullptr;
    d.owner[d.used] = stream;
    return d.base + 2 * d.used++;
  }
  int cus() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDev) return 256;
    std::lock_guard<std::mutex> lock(mu_);
    PerDev& d = dev_[dev];
    if (d.cus == 0) {
      int n = 256;
      (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
      d.cus = n > 0 ? n : 256;
    }
    return d.cus;
  }

 private:
  static constexpr int kMaxDev = 32;
  struct PerDev {
    unsigned* base = nullptr;
    hipStream_t owner[RATO_QUEUES_EAGER] = {};
    int used = 0, next_captured = RATO_QUEUES_EAGER, cus = 0;
    bool warned = false;
  };
  std::mutex mu_;
  PerDev dev_[kMaxDev];
};

// stats.hip: column sums of `part` [nblocks][ncols] into rec_dev[n_stats ..) and rec_host[n_stats ..), and
// rec_host[0 .. n_stats) = rec_dev[0 .. n_stats) -- one launch (rec_host: pinned, device-visible host memory)
int launch_cut_finish(const double* part, int nblocks, int ncols, double* rec_dev, double* rec_host, int n_stats,
                      hipStream_t st);

// Read-back of a few doubles that a kernel writes into PINNED, device-visible host memory: the host pre-sets every word
// to a NaN payload no arithmetic produces (arm) and then watches the words arrive (wait) instead of asking the runtime
// for the end of the stream -- hipStreamSynchronize / hipEventSynchronize enqueue a completion packet behind the last
// kernel and wait for ITS signal: ~6-10 us per wait on this stack, and an event in the middle of a stream was seen to
// resolve only behind LATER kernels.  An aligned 8-byte store arrives whole.  After 2 s without the words the runtime is
// asked after all (a failed launch never writes: the error comes out there).  RATO_CUT_POLL=0: always ask the runtime.
constexpr uint64_t READBACK_PENDING = 0x7ff8dead5a5abeefull;
inline bool readback_poll_enabled() {
  static const int v = [] { const char* e = getenv("RATO_CUT_POLL"); return e ? atoi(e) : 1; }();
  return v != 0;
}
inline void readback_arm(double* host, int n) {
  volatile uint64_t* w = reinterpret_cast<volatile uint64_t*>(host);
  for (int i = 0; i < n; ++i) w[i] = READBACK_PENDING;
}
inline bool readback_pending(const double* host, int n) {   // any word still the pre-set payload?
  const volatile uint64_t* w = reinterpret_cast<const volatile uint64_t*>(host);
  for (int i = 0; i < n; ++i)
    if (w[i] == READBACK_PENDING) return true;
  return false;
}
inline hipError_t readback_wait(const double* host, int n, hipStream_t st) {   // st: what to synchronise if the words stay away
  if (readback_poll_enabled()) {
    const volatile uint64_t* w = reinterpret_cast<const volatile uint64_t*>(host);
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spins = 0;; ++spins) {
      int pending = 0;
      for (int i = 0; i < n; ++i) pending += w[i] == READBACK_PENDING;
      if (!pending) {
        std::atomic_thread_fence(std::memory_order_acquire);
        return hipSuccess;
      }
      __builtin_ia32_pause();
      if ((spins & 0xfff) == 0xfff && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) break;
    }
  }
  // the words stayed away (or polling is off): ask the runtime for the end of the stream -- and then LOOK: a stream that
  // reports success while a word is still the pending payload means the writer never ran (an error path armed the
  // words and launched nothing); that must not be read as data
  const hipError_t e = hipStreamSynchronize(st);
  if (e != hipSuccess || !readback_poll_enabled()) return e;
  return readback_pending(host, n) ? hipErrorNotReady : hipSuccess;
}

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a per-DEVICE setting: remember, per device, the largest size a
// launch site has raised its kernels to (the first, uncaptured call of a shape makes the runtime call; later calls --
// including those recorded into a hipGraph -- make none).
class DynamicLdsLimit {
 public:
  template <class Apply>   // apply(bytes) -> hipError_t: raises every kernel of the launch site
  hipError_t ensure(size_t bytes, Apply&& apply) {
    if (bytes <= 64 * 1024) return hipSuccess;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDev) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lock(mu_);
    if (bytes <= set_[dev]) return hipSuccess;
    const hipError_t e = apply(bytes);
    if (e == hipSuccess) set_[dev] = bytes;
    return e;
  }

 private:
  static constexpr int kMaxDev = 32;
  std::mutex mu_;
  size_t set_[kMaxDev] = {};
};

}  // namespace rato
