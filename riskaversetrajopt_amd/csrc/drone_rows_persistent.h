// Persistent, double-buffered form of the row-parallel drone linearization (included by drone.hip).
//
// Why.  drone_linearize_rows_kernel gives every 64-sample tile its own workgroup: stage the noise tile (reads that
// queue behind the chip-wide store stream: ~11 us), roll the three axes out (50 dependent steps: 10-17 us), sweep the
// rows.  With 64.6 KB of LDS per workgroup 512 workgroups are resident, M = 1e5 is 1563 tiles = 3.05 rounds, and
// because all workgroups of a round start together and do equal work they also RESTART together: three times per
// launch the whole chip sits in staging + rollout with almost no stores in flight (large batches desynchronise by
// themselves after a few rounds, which is why the same kernel reaches 0.73 of 8 TB/s at M = 1e6 and 0.63-0.69 at 1e5).
//
// Here a workgroup is resident for the whole launch (one per CU, 16 waves) and owns TWO table sets in LDS.  The three
// axis waves ("producers") build the tables of tile j+1 while the other thirteen waves sweep the rows of tile j; when
// a producer has finished tile j+1 it helps sweeping tile j.  No workgroup ever waits for its own staging or rollout
// except on its very first tile, and there is no round structure left to synchronise on.
//
// Hand-off (all in LDS, workgroup scope; polls are ds_read + s_sleep like the progress polls of the one-tile kernel):
//   ctl[b].done   monotonic: every wave adds 1 when it has finished with the k-th tile of buffer b
//   ctl[b].epoch  k+1 once wave 0 has claimed buffer b for its (k+1)-th tile (needs done == 16 k), reset the row
//                 queue and the rollout progress and written the tile number (or -1: no more tiles)
//   ctl[b].prog   steps rolled out so far (x, y): row t may be swept once both are > t
//   ctl[b].q      row-task queue (ascending rows; task S = the Z row maximum)
// Tiles are assigned statically: workgroup w takes units w, w + G, w + 2G, ...
#pragma once

constexpr int PR_NW = 16;     // waves per workgroup
constexpr int PR_CTL = 8;     // ints of control state per buffer

__host__ __device__ inline size_t rowsp_buf_floats(int S) { return (size_t)S * ROWS_SAMPLES * 5; }
__host__ __device__ inline size_t rowsp_lds_floats(int S) {
  return 2 * rowsp_buf_floats(S) + (size_t)S * 3 + 2 * PR_CTL + 4;
}

template <bool FACT>
__global__ __launch_bounds__(PR_NW* RATO_WAVE) void drone_linearize_rows_persistent_kernel(
    rato_drone_params P, int n_units, const float* __restrict__ us, const float* __restrict__ dW,
    const float* __restrict__ mass, const float* __restrict__ Qsym, float* __restrict__ G, float* __restrict__ W,
    float* __restrict__ A22, float* __restrict__ g_up, float* __restrict__ Z, float* __restrict__ part) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const size_t M = (size_t)P.M, ld = (size_t)P.ld;
  const int S = P.S;
  const int lane = threadIdx.x & (RATO_WAVE - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / RATO_WAVE);
  float* L = reinterpret_cast<float*>(lds_raw);
  const int BUF = 5 * S * ROWS_SAMPLES;                                    // floats per table set
  float2_t* US = reinterpret_cast<float2_t*>(L + 2 * BUF);                  // [S] (ux, uy)
  float* UZ = reinterpret_cast<float*>(US + S);                             // [S]
  typedef __attribute__((address_space(3))) volatile int lds_vint;
  lds_vint* ctl = (lds_vint*)(UZ + S);                                      // [2][PR_CTL]
  enum { C_Q = 0, C_PX = 1, C_PY = 2, C_EPOCH = 3, C_UNIT = 4, C_DONE = 5 };

  for (int i = threadIdx.x; i < S; i += PR_NW * RATO_WAVE) {
    float2_t u2;
    u2.x = us[i * 3 + 0];
    u2.y = us[i * 3 + 1];
    US[i] = u2;
    UZ[i] = us[i * 3 + 2];
  }
  if (threadIdx.x < 2 * PR_CTL) ctl[threadIdx.x] = 0;
  __syncthreads();

  constexpr int RT = ROWS_SAMPLES;
  constexpr int RPP = FACT ? 2 : 2 * NOBS;
  const size_t tile_floats = rato::packed_tile_stride((size_t)rato::pair_row_offset(S) * RPP * RT);
  const int G_ = gridDim.x;
  // number of tiles this workgroup will process: units w, w + G, ...
  const int my_tiles = (n_units > (int)blockIdx.x) ? (n_units - 1 - (int)blockIdx.x) / G_ + 1 : 0;

  // every poll is bounded: a hand-off that never arrives (a bug) traps the launch (HIP reports an error) instead of
  // hanging the GPU.  2^24 polls x >= 128 cycles is ~1 s; a tile takes < 1 ms.
  auto poll_ge = [&](lds_vint* p, int need) {
    unsigned spins = 0;
    while (*p < need) {
      __builtin_amdgcn_s_sleep(2);
      if (++spins > (1u << 24)) __builtin_trap();
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  };

  // ---- one row task (or the Z task) of the tile whose tables live in buffer b
  auto run_tasks = [&](int b, int tile, float inv_m, float a21, float dtm, const float* q00, const float* qs,
                       const float* q11, bool valid, size_t m) {
    float2_t* A2 = reinterpret_cast<float2_t*>(L + b * BUF);
    float2_t* PP = A2 + (size_t)S * ROWS_SAMPLES;
    lds_vint* c = ctl + b * PR_CTL;
    int* qhead = (int*)(ctl + b * PR_CTL + C_Q);
    float* __restrict__ Gt = G + (size_t)tile * tile_floats + lane;
    auto next_task = [&]() -> int {
      int v = 0;
      if (lane == 0) v = atomicAdd(qhead, 1);
      return __builtin_amdgcn_readfirstlane(v);
    };
    auto wait_steps = [&](int need) {
      unsigned spins = 0;
      while (true) {
        const int p0 = c[C_PX], p1 = c[C_PY];
        if ((p0 < p1 ? p0 : p1) >= need) break;
        __builtin_amdgcn_s_sleep(4);
        if (++spins > (1u << 23)) __builtin_trap();
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    };
    int task = next_task();
    while (task <= S) {
      if (task == S) {
        wait_steps(S);
        if (Z) {
          float zmax = -INFINITY;
          for (int t = 0; t < S; ++t) {
            const float2_t pp = PP[t * ROWS_SAMPLES + lane];
#pragma unroll
            for (int j = 0; j < NOBS; ++j) {
              const float dx = pp.x - P.obs_xy[j][0], dy = pp.y - P.obs_xy[j][1];
              zmax = fmaxf(zmax, 1.0f - (q00[j] * dx * dx + qs[j] * dx * dy + q11[j] * dy * dy));
            }
          }
          if (valid) Z[m] = zmax - P.tol;
        }
      } else {
        const int t = task;
        wait_steps(t + 1);
        const float2_t pp = PP[t * ROWS_SAMPLES + lane];
        float gj[NOBS], wx[NOBS], wy[NOBS];
#pragma unroll
        for (int j = 0; j < NOBS; ++j) {
          const float dx = pp.x - P.obs_xy[j][0], dy = pp.y - P.obs_xy[j][1];
          gj[j] = 1.0f - (q00[j] * dx * dx + qs[j] * dx * dy + q11[j] * dy * dy);
          wx[j] = -(2.0f * q00[j] * dx + qs[j] * dy) * dtm;
          wy[j] = -(qs[j] * dx + 2.0f * q11[j] * dy) * dtm;
        }
        if (FACT && valid) {
#pragma unroll
          for (int j = 0; j < NOBS; ++j) {
            const float dx = pp.x - P.obs_xy[j][0], dy = pp.y - P.obs_xy[j][1];
            W[(((size_t)j * S + t) * 2 + 0) * ld + m] = -(2.0f * q00[j] * dx + qs[j] * dy);
            W[(((size_t)j * S + t) * 2 + 1) * ld + m] = -(qs[j] * dx + 2.0f * q11[j] * dy);
          }
          if (A22) {
            const float2_t at = A2[t * ROWS_SAMPLES + lane];
            A22[((size_t)t * 2 + 0) * ld + m] = at.x;
            A22[((size_t)t * 2 + 1) * ld + m] = at.y;
          }
        }
        float m0x = 1.0f, m0y = 1.0f, m1x = 0.0f, m1y = 0.0f;
        float accx = 0.0f, accy = 0.0f;
        float* __restrict__ Grow = Gt + (size_t)rato::pair_row_offset(t) * (RPP * RT);
        for (int k = t; k >= 1; --k) {
          const float2_t aa = A2[k * ROWS_SAMPLES + lane];
          const float2_t u2 = US[k - 1];
          const float n0x = m0x + m1x * a21, n0y = m0y + m1y * a21;
          const float n1x = m0x * P.dt + m1x * aa.x, n1y = m0y * P.dt + m1y * aa.y;
          m0x = n0x; m0y = n0y; m1x = n1x; m1y = n1y;
          accx += m1x * u2.x;
          accy += m1y * u2.y;
          if (valid) {
            float* __restrict__ o = Grow + (k - 1) * (RPP * RT);
            if (FACT) {
              o[0] = m1x * dtm;
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
          for (int j = 0; j < NOBS; ++j) g_up[((size_t)j * S + t) * ld + m] = -gj[j] + wx[j] * accx + wy[j] * accy;
        }
      }
      task = next_task();
    }
    (void)inv_m;
  };

  // per-tile sample constants of this lane
  struct TileConsts {
    size_t m;
    bool valid;
    float inv_m, a21, dtm, q00[NOBS], qs[NOBS], q11[NOBS];
  };
  auto load_tile_consts = [&](int tile) {
    TileConsts c;
    const size_t m_raw = (size_t)tile * ROWS_SAMPLES + lane;
    c.valid = m_raw < M;
    c.m = c.valid ? m_raw : M - 1;
    c.inv_m = 1.0f / mass[c.m];
    c.a21 = -P.kp * P.dt * c.inv_m;
    c.dtm = P.dt * c.inv_m;
#pragma unroll
    for (int j = 0; j < NOBS; ++j) {
      c.q00[j] = Qsym[(size_t)(j * 3 + 0) * ld + c.m];
      c.qs[j] = Qsym[(size_t)(j * 3 + 1) * ld + c.m];
      c.q11[j] = Qsym[(size_t)(j * 3 + 2) * ld + c.m];
    }
    return c;
  };
  auto signal_done = [&](int b) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) atomicAdd((int*)(ctl + b * PR_CTL + C_DONE), 1);
  };

  if (wave < 3) {
    // ================= producers: axis a = wave.  Sequence: P(0); P(1) H(0) D(0); P(2) H(1) D(1); ...
    const int a = wave;
    for (int j = 0; j <= my_tiles; ++j) {          // j == my_tiles: publish "no more tiles" on the next buffer
      const int b = j & 1, k = j >> 1;
      lds_vint* c = ctl + b * PR_CTL;
      const int unit = (j < my_tiles) ? (int)blockIdx.x + j * G_ : -1;
      if (a == 0) {
        poll_ge(c + C_DONE, PR_NW * k);              // everybody has left the previous tile of this buffer
        if (lane == 0) {
          c[C_Q] = 0;
          c[C_PX] = 0;
          c[C_PY] = 0;
          c[C_UNIT] = unit;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) c[C_EPOCH] = k + 1;
      } else {
        poll_ge(c + C_EPOCH, k + 1);
      }
      if (unit >= 0) {
        const int tile = unit;
        const TileConsts tc = load_tile_consts(tile);
        float* A2f = L + b * BUF;                      // [S][64][2]
        float* PPf = A2f + 2 * S * ROWS_SAMPLES;       // [S][64][2]
        float* AZ = PPf + 2 * S * ROWS_SAMPLES;        // [S][64]
        const int st = (a < 2) ? 2 : 1;
        float* noise = ((a < 2) ? PPf + a : AZ) + lane * st;          // slot (t): noise[t * 64 * st]
        float* a22p = ((a < 2) ? A2f + a : AZ) + lane * st;
        // stage this axis' noise row by row with many loads in flight
        constexpr int MAXR = 16;
        for (int t0 = 0; t0 < S; t0 += MAXR) {
          float tmp[MAXR];
#pragma unroll
          for (int i = 0; i < MAXR; ++i) {
            const int t = t0 + i;
            tmp[i] = (t < S) ? dW[(size_t)(t * 3 + a) * ld + tc.m] : 0.0f;
          }
#pragma unroll
          for (int i = 0; i < MAXR; ++i) {
            const int t = t0 + i;
            if (t < S) noise[t * ROWS_SAMPLES * st] = tmp[i];
          }
        }
        float p = P.x_init[a], v = P.x_init[3 + a];
        const float cn = sqrtf(P.dt) * P.beta * tc.inv_m;
        {
          float xi = noise[0];
          for (int t = 0; t < S; ++t) {
            const int tn = (t + 1 < S) ? t + 1 : t;
            const float xi_n = noise[tn * ROWS_SAMPLES * st];
            const float u = (a == 0) ? US[t].x : ((a == 1) ? US[t].y : UZ[t]);
            const float a22 = 1.0f - P.dt * (P.kd + 2.0f * P.drag * fabsf(v)) * tc.inv_m;
            const float acc = (u - (P.kp * p + P.kd * v)) * tc.inv_m - P.drag * fabsf(v) * v * tc.inv_m;
            const float pn = p + P.dt * v;
            const float vn = v + P.dt * acc + cn * xi;
            p = pn;
            v = vn;
            a22p[t * ROWS_SAMPLES * st] = a22;
            if (a < 2) {
              noise[t * ROWS_SAMPLES * st] = p;
              __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
              if (lane == 0) c[C_PX + a] = t + 1;
            }
            xi = xi_n;
          }
        }
        // final-state adjoint of this axis (rows P, V of d x_S / d u) summed over the tile's samples, and the rhs
        {
          float mP0 = 1.0f, mP1 = 0.0f, mV0 = 0.0f, mV1 = 1.0f, dP = 0.0f, dV = 0.0f;
          for (int s2 = S - 1; s2 >= 0; --s2) {
            const float ua = (a == 0) ? US[s2].x : ((a == 1) ? US[s2].y : UZ[s2]);
            const float eP = mP1 * tc.dtm, eV = mV1 * tc.dtm;
            dP += eP * ua;
            dV += eV * ua;
            const float sp = rato::wave_sum_dpp(tc.valid ? eP : 0.0f);
            const float sv = rato::wave_sum_dpp(tc.valid ? eV : 0.0f);
            if (lane == 0) {
              part[(size_t)tile * (6 * S + 6) + s2 * 6 + a] = sp;
              part[(size_t)tile * (6 * S + 6) + s2 * 6 + 3 + a] = sv;
            }
            if (s2 > 0) {
              const float a22 = a22p[s2 * ROWS_SAMPLES * st];
              const float nP0 = mP0 + mP1 * tc.a21, nP1 = mP0 * P.dt + mP1 * a22;
              const float nV0 = mV0 + mV1 * tc.a21, nV1 = mV0 * P.dt + mV1 * a22;
              mP0 = nP0; mP1 = nP1; mV0 = nV0; mV1 = nV1;
            }
          }
          const float rp = rato::wave_sum_dpp(tc.valid ? (-(p - P.x_final[a]) + dP) : 0.0f);
          const float rv = rato::wave_sum_dpp(tc.valid ? (-(v - P.x_final[3 + a]) + dV) : 0.0f);
          if (lane == 0) {
            part[(size_t)tile * (6 * S + 6) + 6 * S + a] = rp;
            part[(size_t)tile * (6 * S + 6) + 6 * S + 3 + a] = rv;
          }
        }
      }
      // help sweeping the tile produced one round earlier (or this one, if it is the only / last one), then leave it
      if (j >= 1) {
        const int jb = j - 1, hb = jb & 1;
        const int htile = (int)blockIdx.x + jb * G_;
        const TileConsts hc = load_tile_consts(htile);
        run_tasks(hb, htile, hc.inv_m, hc.a21, hc.dtm, hc.q00, hc.qs, hc.q11, hc.valid, hc.m);
        signal_done(hb);
      }
    }
  } else {
    // ================= sweepers
    for (int j = 0; j < my_tiles; ++j) {
      const int b = j & 1, k = j >> 1;
      lds_vint* c = ctl + b * PR_CTL;
      const int tile = (int)blockIdx.x + j * G_;
      const TileConsts tc = load_tile_consts(tile);      // issued before the wait: the loads fly while we poll
      poll_ge(c + C_EPOCH, k + 1);
      run_tasks(b, tile, tc.inv_m, tc.a21, tc.dtm, tc.q00, tc.qs, tc.q11, tc.valid, tc.m);
      signal_done(b);
    }
  }
}
