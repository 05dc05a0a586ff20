// ABI version of librato_saa.so (include/rato_saa.h).
#include "rato_common.h"

extern "C" int rato_abi_version(void) { return RATO_ABI_VERSION; }

extern "C" size_t rato_packed_tile_stride(size_t payload_floats) { return rato::packed_tile_stride(payload_floats); }

extern "C" size_t rato_packed_buffer_floats(size_t n_tiles, size_t payload_floats) {
  return n_tiles * rato::packed_tile_stride(payload_floats);
}

// Diagnostic: the shader clock the device sustains right now.  One wave reads the shader-cycle counter (s_memtime)
// and the constant 100 MHz counter (s_memrealtime) `us` microseconds apart: sclk = cycles / elapsed.  (Calibrated
// against a chain of dependent fp32 FMAs, 7 cycles each: tools/clock_probe.py.)  bench.py reports it beside the
// roofline, because the pool's boxes run the same binary 5-10 % apart while their store-only ceilings agree to 2 %.
namespace {
__global__ void clock_probe_kernel(double* out, unsigned ticks) {
  if (threadIdx.x != 0) return;
  const unsigned long long c0 = clock64();
  const unsigned long long t0 = wall_clock64();
  unsigned long long t1 = t0;
  while (t1 - t0 < ticks) {
    __builtin_amdgcn_s_sleep(8);
    t1 = wall_clock64();
  }
  const unsigned long long c1 = clock64();
  const double us = (double)(t1 - t0) / 100.0;
  out[0] = (double)(c1 - c0) / us;   // MHz
  out[1] = us;
  out[2] = 0.0;
}
// Diagnostic: `blocks` workgroups of 1024 threads that each hold their wave slots for `us` microseconds of the constant
// 100 MHz clock (512 of them fill every wave slot of the 256 CUs).  Tests use it to put the one-launch statistics
// (rs_coop: waits inside the launch for its own workgroups) behind / beside a kernel that owns the chip.
__global__ __launch_bounds__(1024) void occupy_kernel(unsigned long long ticks) {
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
}
}  // namespace

extern "C" int rato_device_occupy(int32_t blocks, int64_t us, void* stream) {
  RATO_CLEAR_ERROR();
  if (blocks < 1 || blocks > 65535 || us < 1 || us > 5000000) return RATO_EINVAL;
  hipLaunchKernelGGL(occupy_kernel, dim3((unsigned)blocks), dim3(1024), 0, rato::as_stream(stream),
                     (unsigned long long)us * 100ull);
  RATO_LAUNCH_CHECK();
  return RATO_OK;
}

// Plumbing: one asynchronous copy between any two of (device memory, pinned host memory) on the caller's stream.  The
// facades' small read-backs / uploads inside the SCP loop go through this instead of torch.Tensor.copy_, whose dispatch
// costs ~40 us per call on this stack -- more than the copies and several of the kernels they sit between.
extern "C" int rato_copy_async(void* dst, const void* src, size_t bytes, void* stream) {
  RATO_CLEAR_ERROR();
  if (!dst || !src) return RATO_EINVAL;
  if (bytes == 0) return RATO_OK;
  const hipError_t e = hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault, rato::as_stream(stream));
  return e == hipSuccess ? RATO_OK : RATO_EHIP - (int)e;
}

extern "C" int rato_stream_synchronize(void* stream) {
  const hipError_t e = hipStreamSynchronize(rato::as_stream(stream));
  return e == hipSuccess ? RATO_OK : RATO_EHIP - (int)e;
}

extern "C" int rato_device_clock_probe(double* out3, int32_t us, void* stream) {
  RATO_CLEAR_ERROR();
  if (!out3 || us < 1 || us > 100000) return RATO_EINVAL;
  hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(RATO_WAVE), 0, rato::as_stream(stream), out3, (unsigned)us * 100u);
  RATO_LAUNCH_CHECK();
  return RATO_OK;
}
