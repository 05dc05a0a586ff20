// What does a row kernel's write stream lose when its workgroups do not store all the time?  A replay of
// car_linearize_rows_kernel's write stream at the C5 shard (M = 125,000, S = 40: 1954 tiles of 1560 rows x 256 B, tiles
// from a global queue, row tasks from an LDS queue in ascending order, two 256 B stores per step) with the kernel's OTHER
// costs put back one at a time as measured by the diagnostic builds (tools/car_phases.py):
//   idle   every workgroup stores nothing for `idle` microseconds per tile (staging 5.6 + rollout 8.6 + barrier and queue
//          1.5 = 15.7 us in the kernel): all waves sleep,
//   work   `work` dependent packed FMAs per step in front of the two stores (the sweep: ~114 SIMD cycles per wave-step,
//          14.1 us per tile with the stores compiled out),
//   lds    bytes of LDS the workgroup claims (61,440 + tables in the kernel: two workgroups per CU).
// Printed: time per launch, TB/s, and the time per tile of a workgroup.
//   stagger / offset: do workgroups that all start together stay in step (everybody idle, then everybody storing)?
//   hipcc --offload-arch=gfx950 -O3 -w tools/store_duty.hip -o /tmp/sd && /tmp/sd
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
constexpr int S = 40;
constexpr size_t TILE_FLOATS = (size_t)S * (S - 1) / 2 * 2 * 64;   // 1560 x 64
typedef float f2 __attribute__((ext_vector_type(2)));

template <int NW, bool STORE>
__global__ __launch_bounds__(NW * 64) void tiles(float* p, int n_tiles, unsigned* queue, int idle_ticks, int work, float seed,
                                                  int stagger, int offset_ticks, int xcd_mask,
                                                  const float* __restrict__ src, long src_ld, int read_mode) {
  extern __shared__ int lds[];
  if (!((xcd_mask >> (blockIdx.x & 7)) & 1)) return;     // workgroup i runs on XCD i mod 8: only the XCDs in the mask store
  int& tile_s = lds[0];
  int& head = lds[1];
  const int lane = threadIdx.x & 63;
  for (int first = 1;; first = 0) {
    __syncthreads();
    if (threadIdx.x == 0) {
      tile_s = (first && xcd_mask == 0xff) ? (int)blockIdx.x : (xcd_mask == 0xff ? (int)gridDim.x : 0) + (int)atomicAdd(queue, 1u);
      head = 1;
    }
    __syncthreads();
    const int tile = tile_s;
    if (tile >= n_tiles) break;
    if (first && offset_ticks > 0) {     // a phase offset and nothing else: workgroup i waits (i mod 4) / 4 of a period first
      const long long w = (long long)offset_ticks * ((blockIdx.x >> 3) & 3) / 4;   // (i / 8: different phases INSIDE an XCD)
      const unsigned long long t0 = wall_clock64();
      while ((long long)(wall_clock64() - t0) < w) __builtin_amdgcn_s_sleep(16);
    }
    float rsum = 0.f;
    if (read_mode) {   // the tile's noise: 80 rows of 256 B, 10 per wave -- 1: rows src_ld floats apart (the [2S][M] layout), 2: one 20 KB block
      const int wave = threadIdx.x >> 6;
      float v[10];
#pragma unroll
      for (int i = 0; i < 10; ++i) {
        const int r = wave + i * 8;
        v[i] = (read_mode == 1) ? src[(size_t)r * src_ld + (size_t)tile * 64 + lane] : src[(size_t)tile * 5120 + r * 64 + lane];
      }
#pragma unroll
      for (int i = 0; i < 10; ++i) rsum += v[i];
    }
    if (idle_ticks > 0) {
      const unsigned long long t0 = wall_clock64();
      while ((long long)(wall_clock64() - t0) < idle_ticks) __builtin_amdgcn_s_sleep(16);
    }
    if (rsum == 123.456f) p[tile] = rsum;
    float* base = p + (size_t)tile * TILE_FLOATS;
    for (;;) {
      int t = 0;
      if (lane == 0) t = atomicAdd(&head, 1);
      t = __builtin_amdgcn_readfirstlane(t);
      if (t >= S) break;
      if (first && stagger > 0 && (t % stagger) > (((int)blockIdx.x >> 3) % stagger)) continue;   // a part tile first: ((i / 8) mod k) + 1 of k rows
      float* row = base + (size_t)(t * (t - 1) / 2) * 2 * 64;
      f2 e = {seed, (float)lane};
      for (int k = t; k >= 1; --k) {
        for (int w = 0; w < work; ++w) e = e * 1.0000001f + seed;     // dependent chain of packed FMAs
        float* o = row + (size_t)(k - 1) * 2 * 64;
        if (STORE || e.x == 123.456f) {
          o[lane] = e.x;
          o[64 + lane] = e.y;
        }
      }
    }
  }
}

int main(int argc, char** argv) {
  const long M = argc > 1 ? atol(argv[1]) : 125000;
  const int n_tiles = (int)((M + 63) / 64);
  float* p; unsigned* q;
  (void)hipMalloc(&p, (size_t)n_tiles * TILE_FLOATS * 4); (void)hipMalloc(&q, 4);
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  const double bytes = (double)n_tiles * TILE_FLOATS * 4;
  int stagger = 0, offset_ticks = 0, xcd_mask = 0xff, read_mode = 0;
  const long src_ld = (long)n_tiles * 64;
  float* src; (void)hipMalloc(&src, (size_t)80 * src_ld * 4); (void)hipMemset(src, 0, (size_t)80 * src_ld * 4);
  auto run = [&](int wgs, int lds_bytes, double idle_us, int work, bool store) {
    float sum = 0;
    const int idle_ticks = (int)(idle_us * 100.0);
    for (int i = 0; i < 8; ++i) {
      (void)hipMemsetAsync(q, 0, 4, 0);
      (void)hipEventRecord(a);
      if (store) hipLaunchKernelGGL((tiles<8, true>), dim3(wgs), dim3(512), lds_bytes, 0, p, n_tiles, q, idle_ticks, work, 0.5f, stagger, offset_ticks, xcd_mask, src, src_ld, read_mode);
      else hipLaunchKernelGGL((tiles<8, false>), dim3(wgs), dim3(512), lds_bytes, 0, p, n_tiles, q, idle_ticks, work, 0.5f, stagger, offset_ticks, xcd_mask, src, src_ld, read_mode);
      (void)hipEventRecord(b); (void)hipEventSynchronize(b);
      float ms; (void)hipEventElapsedTime(&ms, a, b);
      if (i >= 2) sum += ms;
    }
    const double ms = sum / 6;
    const double rounds = (double)n_tiles / wgs;
    double b = bytes;
    if (stagger > 0)   // the rows the first units skipped
      for (int i = 0; i < wgs && i < n_tiles; ++i)
        for (int t = 1; t < S; ++t)
          if ((t % stagger) > ((i >> 3) % stagger)) b -= 2.0 * t * 256.0;
    if (xcd_mask != 0xff) printf("xcd mask 0x%02x ", xcd_mask);
    if (read_mode) printf("reads %s ", read_mode == 1 ? "80 rows x 256 B, strided" : "one 20 KB block      ");
    printf("wgs %4d lds %6d idle %5.1f us work %3d stores %d stagger %d offset %4.1f us : %.4f ms  %.2f TB/s  %.1f us per tile and workgroup\n",
           wgs, lds_bytes, idle_us, work, (int)store, stagger, offset_ticks / 100.0, ms, store ? b / ms / 1e9 : 0.0, ms * 1e3 / rounds);
    fflush(stdout);
  };
  (void)hipFuncSetAttribute((const void*)tiles<8, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  (void)hipFuncSetAttribute((const void*)tiles<8, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  const int L2 = 64 * 1024;   // two workgroups per CU
  printf("== A. how many storing workgroups does the write path need? (stores only)\n");
  for (int wgs : {32, 64, 128, 256, 512}) run(wgs, L2, 0.0, 0, true);
  printf("== B. sweep arithmetic alone (no stores): calibrating `work` against 14.1 us per tile\n");
  for (int work : {4, 8, 12, 16, 24}) run(512, L2, 0.0, work, false);
  printf("== C. stores + an idle phase per tile (nothing else)\n");
  for (double idle : {0.0, 5.0, 10.0, 15.7, 25.0}) run(512, L2, idle, 0, true);
  printf("== D. stores + sweep arithmetic (no idle phase)\n");
  for (int work : {4, 8, 12, 16, 24}) run(512, L2, 0.0, work, true);
  printf("== E. stores + sweep arithmetic + idle phase: the kernel's three parts together\n");
  for (int work : {8, 12, 16}) for (double idle : {10.0, 15.7}) run(512, L2, idle, work, true);
  printf("== F. the same with three / four workgroups per CU (smaller LDS claim)\n");
  for (int per_cu : {3, 4}) for (int work : {8, 12, 16}) run(256 * per_cu, 160 * 1024 / per_cu - 1024, 15.7, work, true);
  printf("== G. are the workgroups' phases correlated?  part tiles first ((i mod k) + 1 of k rows), or a bare phase offset\n");
  for (int work : {2, 4}) {
    stagger = 0; offset_ticks = 0;
    run(512, L2, 15.7, work, true);
    for (int k : {2, 4}) { stagger = k; run(512, L2, 15.7, work, true); }
    stagger = 0;
    for (int off : {2000, 4000}) { offset_ticks = off; run(512, L2, 15.7, work, true); }
    offset_ticks = 0;
  }
  printf("== I. is the ceiling one per XCD?  512 workgroups launched, only those on the XCDs of the mask store (all tiles from the queue)\n");
  for (int mask : {0x01, 0x03, 0x0f, 0x55, 0xff}) {
    xcd_mask = mask;
    run(512, L2, 0.0, 0, true);
  }
  xcd_mask = 0xff;
  printf("== K. the tile's noise read at the start of the idle phase: rows of the [2S][M] layout, or one block per tile\n");
  for (int mode : {0, 1, 2, 1, 2, 0}) {
    read_mode = mode;
    run(512, L2, 15.7, 2, true);
  }
  for (int mode : {0, 1, 2}) {
    read_mode = mode;
    run(512, L2, 15.7, 2, false);
  }
  read_mode = 0;
  printf("== H. sweep arithmetic with no idle phase at work 0 / 2 (stores off, then on)\n");
  for (int work : {0, 2}) { run(512, L2, 0.0, work, false); run(512, L2, 0.0, work, true); run(512, L2, 15.7, work, true); }
  return 0;
}
