/*
 * rato_saa.h — C ABI of the MI355X-native SAA inner loop (librato_saa.so).
 *
 * Drop-in boundary for the hot path of StanfordASL/RiskAverseTrajOpt: the
 * per-SCP-iteration batched rollout, control-Jacobian linearization, sample
 * mean and Monte-Carlo VaR/CVaR evaluation over M samples x S steps.  The
 * reference has no FFI for this path (it is pure Python on JAX-CPU); each entry
 * point below names the reference method (file:line under /root/reference) it
 * replaces.  INTEGRATION.md shows the ctypes stub a maintainer would add.
 *
 * Conventions (all entry points):
 *   - extern "C", plain pointers and sizes, no torch types.
 *   - every data pointer is a CALLER-OWNED DEVICE pointer (fp32 unless noted),
 *     e.g. torch.Tensor.data_ptr(); nothing is allocated or freed here.
 *   - `stream` is a hipStream_t passed as void* (0 = the null stream); calls
 *     are asynchronous and stream-ordered and never synchronise the device --
 *     with the exceptions that exist to BE the synchronisation point of a host
 *     loop and say so where they are declared: rato_stream_synchronize,
 *     rato_cut_oracle_rollout (one oracle round trip of the cutting-plane loop:
 *     its results are read by the host master) and rato_cut_solve (the whole
 *     loop of one SCP subproblem).  Each waits for the work it issued on
 *     `stream` only, never for the device: by watching the words its last
 *     launch writes into the caller's PINNED host buffers arrive (every kernel
 *     in front of that launch has then completed; the HIP runtime is asked --
 *     hipStreamSynchronize -- only when RATO_CUT_POLL=0 is set or the words stay
 *     away for 2 s, e.g. after a failed launch, whose error comes out there).
 *   - no global state beyond cached device properties (CU count, LDS attribute) and, for the row-parallel
 *     linearize kernels on large batches, one self-cleaning two-word work queue per STREAM in device memory
 *     (up to 64 streams; launches on one stream are ordered and share it, a 65th stream falls back to the static
 *     launch): safe to call concurrently from one host thread per GPU, and — after one ordinary call per kernel
 *     variant — inside a hipGraph stream capture (no non-stream runtime call is made).
 *   - return value: 0 ok; RATO_EINVAL bad argument; RATO_EHIP-<hipError_t>
 *     when a launch fails.
 *   - the sample index m is ALWAYS the fastest-varying index (SoA), so every
 *     wave reads/writes 256 contiguous bytes per instruction.
 *
 * Packed causal layout of the obstacle/separation Jacobian: d g_t / d u_s is
 * identically zero unless s <= t-1 (position lags control by two steps), so
 * only the pairs (t, s), 0 <= s < t < S are stored, row-major in t:
 *     pair(t, s) = t*(t-1)/2 + s,   n_pairs = S*(S-1)/2.
 * The Jacobian buffer G is tile-blocked SoA: samples are grouped in tiles of
 * TILE consecutive samples and each tile is a contiguous [rows][TILE] block, so
 * that one wave's / workgroup's output is one contiguous region that it writes
 * as a stream:
 *     G[tile][row][lane],  tile = m / TILE, lane = m % TILE,
 *     n_tiles = ceil(M / TILE); lanes >= M of the last tile are not written.
 * Tile stride: consecutive tiles are rato_packed_tile_stride(rows * TILE) floats apart -- back to back while a tile
 * is smaller than 1 MiB, otherwise every tile starts on a 2 MiB boundary and G itself must be 2 MiB aligned
 * (buffer size: n_tiles * stride floats).  One rule from the tile shape alone, shared by every producer and consumer
 * of the layout in this library (linearize, rowmax / tail rows, CSC emission).  It is there for the write path: the
 * row-parallel kernels keep one store stream per resident tile, and 512 streams a power-of-two distance apart are
 * spread evenly over the memory channels (1.88 MB tiles, store-only: 5.1-5.2 TB/s back to back, 5.7 at 2 MiB).
 * TILE depends on the kernel variant and is reported by the *_plan() call
 * (64 for the row-parallel drone kernel, RATO_TILE = 256 otherwise).
 */
#ifndef RATO_SAA_H
#define RATO_SAA_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RATO_OK 0
#define RATO_EINVAL (-1)
#define RATO_ENONFINITE (-2) /* a checked output holds NaN/Inf: the reference only prints
                                "[solve]: Problem infeasible." (drone_risk.py:458-459) and carries on.
                                Raised by the facades (check_finite=True) from rato_count_nonfinite_acc. */
#define RATO_EINFEASIBLE (-4) /* rato_master_solve: the rows of the master QP admit no point */
#define RATO_ENOCOMM (-3)    /* librccl could not be bound at run time (rato_comm_*) */
#define RATO_ERANK (-6)      /* rato_cut_solve: the final rows are rank deficient / not fewer than the variables */
#define RATO_ESELECT (-7)    /* rato_cut_solve: an oracle round trip came back with NaN statistics (the one-launch tail
                                selection gave up, or the m values hold NaN): repeat with the recovering host loop */
#define RATO_ENNLS (-8)      /* rato_cut_solve: the NNLS of the master did not converge */
#define RATO_EHIP (-1000)    /* RATO_EHIP - hipError_t */
#define RATO_ERCCL (-2000)   /* RATO_ERCCL - ncclResult_t */

#define RATO_TILE 256        /* samples per Jacobian tile (= workgroup size) */
#define RATO_DRONE_NOBS 3   /* drone_params.py:34-43: n_obs = 3 */
#define RATO_HOPPER_NFEAT 30 /* hopper.py:69: num_mu_features = 30 */

/* ABI version, bumped on any signature/layout change (2: factored Jacobian W / A22 outputs, CVaR-cut oracle
 * entry points, record unpack; 3: generators-only linearization, Jacobian-free tail rows, a22_axes; 4: rato_comm_*
 * (RCCL behind the ABI), rato_car_separation_distances, rato_count_nonfinite_acc, Philox sampler entry points;
 * 5: rato_*_linearize_philox, rato_hopper_slip_host_inputs; 6: packed tile stride (rato_packed_tile_stride /
 * rato_packed_buffer_floats), rato_sums_and_risk_stats; 7: fp64 CVaR-cut oracle -- rato_saa_rowmax /
 * rato_drone_rowmax_implicit take (base, sign, double xs), the tail-row entry points write double partials
 * (rato_sum_partials_f64), rato_saa_tail_rows folded into rato_saa_tail_rows_batch (slots == NULL), params.rows_out,
 * rato_drone_rowmax_rollout / rato_drone_tail_rows_rollout, rato_comm_available, rato_device_occupy; 8: fp64 constants in
 * rato_car_params, rato_car_rowmax_rollout / rato_car_tail_rows_rollout, rato_cut_oracle_rollout, rato_nnls_warm, rato_master_*,
 * (9: rato_cut_solver_* / rato_cut_begin / rato_cut_solve -- the cutting-plane loop of a subproblem as one call;
 * params.stats_* -- the statistics of Z in extra workgroups of the row-parallel linearize launch itself)
 * rato_copy_async, rato_stream_synchronize, rato_risk_stats_recover; 12: rato_hopper_slip_hessian,
 * rato_hopper_jacobian_nnz / rato_hopper_emit_jacobian_values -- the hopper's jacrev / Lagrangian Hessian in the reference's
 * layout; rato_scp_run_drone / rato_scp_iter -- the reduced SCP loop as one call).
 * The Python binding refuses a library that reports another version. */
#define RATO_ABI_VERSION 12
#define RATO_STATS_IN_LAUNCH 1   /* params.stats_flags */
int rato_abi_version(void);

/* floats between consecutive tiles of a packed tile-blocked Jacobian whose tile holds payload_floats numbers */
size_t rato_packed_tile_stride(size_t payload_floats);
/* floats a packed buffer of n_tiles such tiles needs (n_tiles * stride): what to allocate for G -- the linearize entry
 * points take no buffer sizes, so a G allocated as n_tiles * payload_floats is overrun when the tiles are padded */
size_t rato_packed_buffer_floats(size_t n_tiles, size_t payload_floats);

/* Diagnostic (no reference counterpart): the shader clock the device sustains at this moment -- shader-cycle counter
 * against the constant 100 MHz counter over `us` microseconds (1..100000), one wave.  out3 (device, 3 doubles) =
 * { MHz, elapsed us, 0 }.  bench.py quotes it beside the roofline (boxes of one pool run the same binary several
 * percent apart). */
int rato_device_clock_probe(double* out3, int32_t us, void* stream);

/* Diagnostic (tests): `blocks` workgroups of 1024 threads hold their wave slots for `us` microseconds (<= 5 s); 512 of
 * them occupy every wave slot of an MI355X.  Used to show that the one-launch statistics, which wait inside the
 * launch for their own workgroups, survive a chip that another stream owns. */
int rato_device_occupy(int32_t blocks, int64_t us, void* stream);

/* Plumbing for the facades' small transfers inside the SCP loop: one asynchronous copy between device memory and
 * (pinned) host memory in either direction on `stream` (hipMemcpyDefault), and the matching synchronisation. */
int rato_copy_async(void* dst, const void* src, size_t bytes, void* stream);
int rato_stream_synchronize(void* stream);

/* ------------------------------------------------------------------ drone */

/* Constants of drone_params.py:1-45 / Model.__init__ drone_risk.py:71-93. */
typedef struct rato_drone_params {
  int32_t M;            /* samples in this shard (< 2^31: one GPU holds ~2e7 drone samples at S = 50; every offset
                           derived from it is computed in 64 bits) */
  int32_t ld;           /* row stride (floats) of every [..][M] array, ld >= M; a multiple
                           of samples_per_lane (use a multiple of 4) */
  int32_t S;            /* control intervals; dt = T/S (drone_risk.py:82) */
  float dt;
  float beta;           /* diffusion magnitude, 1e-2 */
  float drag;           /* drag_coefficient, 0.2 */
  float kp, kd;         /* -feedback_gain: 0.05, 0.25 */
  float tol;            /* OSQP_TOL subtracted from the max constraint (1e-3) */
  float x_init[6];
  float x_final[6];
  float obs_xy[RATO_DRONE_NOBS][2];  /* obs_positions[:, :2] */
  int32_t rows_out;     /* what the linearize entry points write into their g_up buffer: 0 = the reference's
                           g_up = -g + (grad g).u_k (drone_risk.py:278); 1 = the constraint value g at u_k itself
                           (the base of the oracle's delta form: rows(u) = g + G (u - u_k), see rato_saa_rowmax) */
  /* The same constants in fp64, for the entry points that compute in double precision (the generators-only
   * linearization and the CVaR-cut oracle): the reference's constants are Python floats, and 0.05f is 1.5e-8 away from
   * 0.05, -1.4f is 2.4e-8 away from -1.4 -- which an obstacle row sees as 5e-7 of d = p - o where the drone grazes the
   * obstacle.  The fp32 kernels keep reading the float fields above; fill both from the same numbers. */
  double dt64, beta64, drag64, kp64, kd64, tol64;
  double x_init64[6];
  double x_final64[6];
  double obs_xy64[RATO_DRONE_NOBS][2];
  /* Statistics with the linearization (round 4).  stats_workspace != NULL (an initialised rato_risk_stats workspace;
     row-parallel kernel only: cols_per_thread = -1; Z requested; M <= 524,288): the call also leaves the rato_risk_stats
     record of the Z it produces in stats_out (double[RATO_N_STATS]; tail level stats_alpha, threshold stats_thr).
     SMALL batches (every workgroup of the launch resident at once: fewer tiles than workgroup slots, e.g. BASELINE C2 /
     C3) compute it in a few extra workgroups of the SAME launch: Z is written right after the rollout, the statistics
     workgroups wait until every tile's Z has been counted in and select while the Jacobian is still being stored -- no
     dependent statistics launch.  Larger batches issue rato_risk_stats behind the kernel from the same call: beside a
     store-saturated producer the selection's dependent global round trips run several times slower than behind it.
     Same selection, same record (fp64 sums equal to summation order).  Ignored by every other entry point. */
  void* stats_workspace;
  double* stats_out;
  double stats_alpha;
  float stats_thr;
  int32_t stats_flags;       /* RATO_STATS_IN_LAUNCH: rato_drone_eval computes them in its own launch (default: behind it) */
} rato_drone_params;

/*
 * Replaces Model.us_to_state_trajectories (drone_risk.py:139-162) fused with
 * obstacle_avoidance_constraints (:169-213) and the Monte-Carlo closure
 * monte_carlo_no_collisions_constraint_verification (:656-662).
 *   us    [S][3]            controls
 *   dW    [S][3][ld]        velocity rows 3..5 of the reference's DWs (M,S,6)
 *   mass  [M]
 *   Qsym  [3 obs][3][M]     (Q00, Q01+Q10, Q11) of obs_Q[:, :2, :2]
 * (every "[M]" below is a row of p->ld floats of which the first M are used)
 * outputs (any may be NULL):
 *   Z     [M]               max_{j,t} g - tol
 *   xs    [S+1][6][M]       state trajectories (SoA of the reference's (M,S+1,6))
 *   g     [3 obs][S][M]     constraint values
 * Without trajectories (xs == NULL) the tiled kernel runs: one wave per tile of 64 samples, the noise of 32 steps in
 * flight before the first step, the vertical axis (which no obstacle row reads) not rolled out; Z and g to the bit as
 * with trajectories.
 * Monte-Carlo step in one call (ABI 11; drone_risk.py:643-725: rollout -> max -> fraction / VaR / AVaR): with
 * p->stats_workspace / stats_out / stats_alpha / stats_thr set (and Z requested) the call also leaves the rato_risk_stats
 * record of Z in stats_out -- by rato_risk_stats behind the kernel; with RATO_STATS_IN_LAUNCH in p->stats_flags, for
 * batches without trajectories up to rato_drone_eval_stats_in_launch(M), in an extra workgroup of the SAME launch
 * (measured slower than the two launches: DESIGN.md; kept for A/B).
 */
int rato_drone_eval_stats_in_launch(int32_t M);
int rato_drone_eval(const rato_drone_params* p, const float* us, const float* dW,
                    const float* mass, const float* Qsym,
                    float* Z, float* xs, float* g, void* stream);

/* Kernel variants of rato_drone_linearize:
 *   cols_per_thread = -1            row-parallel adjoint kernel (default; needs
 *                                   20*64*S + 16 bytes of LDS <= 160 KB, S >= 2)
 *   cols_per_thread in {4,8,16,32}  forward column kernel, samples_per_lane 1
 *   samples_per_lane 2 x cpt {4,8}, samples_per_lane 4 x cpt {2,4}
 * This call resolves 0 / 0 ("let the library choose") to a concrete pair and
 * returns the number of sample blocks the launch will use = rows of
 * part.  Returns <0 on bad arguments. */
int rato_drone_linearize_plan(int32_t M, int32_t S, int32_t ld,
                              int32_t* cols_per_thread, int32_t* samples_per_lane,
                              int32_t* tile /* out: TILE of the G layout */);

/*
 * Replaces vmap(Model.get_all_constraints_coeffs) (drone_risk.py:239-290) and
 * the per-block stage of the sample mean (:294-296).
 * outputs:
 *   G        W == NULL: [n_tiles][n_pairs][2 axes][3 obs][TILE]
 *                                         d g[j,t] / d u[s,axis] for s<t
 *                                         (the z-control column is identically 0)
 *            W != NULL (factored, row-parallel kernel only):
 *                       [n_tiles][n_pairs][2 axes][TILE]   Phi[t,s,axis] = d p_axis(t+1) / d u[s,axis]
 *   W        NULL, or  [3 obs][S][2 axes][ld]              W[j,t,axis] = d g[j,t] / d p_axis(t+1),
 *            so that d g[j,t]/d u[s,axis] = W[j,t,axis] * Phi[t,s,axis]: the three obstacles share
 *            Phi, i.e. the same Jacobian in S(S-1) + 6S instead of 3S(S-1) numbers per sample.
 *   A22      NULL, or  [S][2 axes][ld]  (row-parallel kernel only)  the state-dependent entry of the step
 *            Jacobian A_t = d x_{t+1} / d x_t = [[1, dt], [-kp dt/m, A22[t]]] per horizontal axis: with W and
 *            g_up this is the whole linearization in 11 S numbers per sample, enough to evaluate G.u for
 *            ANY u by an O(S) recursion (rato_drone_rowmax_implicit) instead of an O(S^2) read of Phi.
 *   g_up     [3 obs][S][M]                -g + grad g . u   (:278)
 *   Z        [M] or NULL                  max_{j,t} g - tol at this iterate
 *   part     [nblocks][6*S + 6]           per-block sums (nblocks from rato_drone_linearize_plan):
 *                                         [s*6 + e], e = (P_x,P_y,P_z,V_x,V_y,V_z): d x_S / d u_{s,axis};
 *                                         [6*S + r]: -v_final + v_final_du.u (:271), r = row of x
 * cols_per_thread / samples_per_lane: see rato_drone_linearize_plan (0 = choose).
 * All [..][M] arrays (dW, mass, Qsym, g_up, Z) use the row stride p->ld.
 */
int rato_drone_linearize(const rato_drone_params* p, const float* us, const float* dW,
                         const float* mass, const float* Qsym,
                         float* G, float* W, float* A22, float* g_up, float* Z, float* part,
                         int32_t cols_per_thread, int32_t samples_per_lane, void* stream);

/* Model.obstacle_avoidance_constraints on given trajectories (drone_risk.py:198-213).
 *   xs [S+1][6][M] -> g [3 obs][S][M] */
int rato_drone_obstacle_constraints(const rato_drone_params* p, const float* xs,
                                    const float* Qsym, float* g, void* stream);

/* The reference's Monte-Carlo report evaluates MANY control sequences on one validation batch (drone_risk.py:697-725:
 * 4 alpha x 30 repeats; driving.py:675-740): K sequences in ONE call -- the rollouts of all of them in one launch
 * (grid = tiles x K: a single sequence at M = 1e4 occupies 157 of the chip's 1024 SIMDs), the K exact selections in a
 * second one.    us [K][S][n_u];  Z [K][ldz] (ldz >= M);  stats_out [K][RATO_N_STATS] or NULL (then alpha, thr,
 * workspace are not read).  Row k is, to the bit, what rato_*_eval + rato_risk_stats give for sequence k. */
int rato_drone_eval_batch(const rato_drone_params* p, int32_t K, const float* us, const float* dW, const float* mass,
                          const float* Qsym, float* Z, int64_t ldz, double alpha, float thr, void* workspace,
                          size_t workspace_bytes, double* stats_out, void* stream);
/* (driving: rato_car_eval_batch, below) */
/* the records of K rows of Z [K][ldz] -> out [K][RATO_N_STATS] (one launch while M <= 12,288) */
int rato_risk_stats_batch(const float* Z, int64_t M, int64_t ldz, int32_t K, double alpha, float thr, void* workspace,
                          size_t workspace_bytes, double* out, void* stream);

/* ---------------------------------------------------------------- driving */

/* Constants of driving_params.py:1-42 / Model.__init__ driving.py:84-120. */
/* 1 when rato_drone_linearize / rato_car_linearize (row-parallel kernel) with params.stats_* would compute the statistics
 * in its own launch for this batch (small batches), 0 when it would issue rato_risk_stats behind the kernel. */
int rato_drone_stats_in_launch(int32_t M, int32_t S);
int rato_car_stats_in_launch(int32_t M, int32_t S);
/* 1 when the row-parallel linearize kernel would write this batch's Jacobian with streaming (non-temporal) stores: the
 * output is far beyond the 256 MB memory-side cache (>= 256 MB) and the batch's noise fits it (<= 128 MB), so that the
 * noise stays cached from one linearization to the next.  factored: the drone's (Phi, W) output. */
int rato_drone_rows_streaming_stores(int64_t M, int32_t S, int32_t factored);
int rato_car_rows_streaming_stores(int64_t M, int32_t S);

typedef struct rato_car_params {
  int32_t M;
  int32_t S;
  float dt;
  float beta;            /* 3e-2 */
  float speed_ped_des;   /* 1.3 */
  float d_min;           /* min_separation_distance */
  float tol;             /* OSQP_TOL (3e-4) */
  float ego_init[4];     /* state_init[0:4] (sample independent) */
  float ego_goal[4];     /* driving.py:217-220 */
  int32_t rows_out;      /* g_up buffer of rato_car_linearize*: 0 = g_up = -g + (grad g).u_k (driving.py:295),
                            1 = g itself (see rato_drone_params.rows_out) */
  /* the same constants in double precision, for the entry points that compute in fp64 (rato_car_*_rollout) */
  double dt64, beta64, speed_ped_des64, d_min64;
  double ego_init64[4];
  void* stats_workspace;  /* statistics in the same launch: as rato_drone_params.stats_* */
  double* stats_out;
  double stats_alpha;
  float stats_thr;
  int32_t stats_flags;       /* as rato_drone_params.stats_flags */
} rato_car_params;

/* Scratch floats needed by the driving entry points for the shared ego
 * trajectory and ego sensitivities (sample-independent, recomputed per call). */
size_t rato_car_ego_scratch_floats(int32_t S);

/*
 * Replaces Model.us_to_state_trajectories (driving.py:186-214) fused with
 * separation_distances_at_all_times (:223-236) and
 * monte_carlo_separation_constraints_verification (:630-638).
 *   us      [S][2]
 *   dW      [S][2][M]       rows 6..7 of the reference's DWs (M,S,8)
 *   x0_ped  [4][M]          states_init[:, 4:8]
 *   w_speed [M], w_rep [M]  omegas_speed, omegas_repulsive
 *   ego_scratch             rato_car_ego_scratch_floats(S) floats
 * outputs (any may be NULL):
 *   Z   [M]                 max_t(-distance_t) - tol
 *   xs  [S+1][8][M]
 *   g   [S][M]              -distance_t
 * Without trajectories (xs == NULL) the rollout is ONE launch: every workgroup folds the sample-independent ego trajectory
 * into its own LDS (the arithmetic of the ego prologue, to the bit) and rolls its tiles out (one wave per 64 samples, the
 * noise of 32 steps in flight).  Monte-Carlo step in one call (ABI 11; driving.py:618-740): p->stats_* / stats_flags as
 * for rato_drone_eval.
 */
int rato_car_eval_stats_in_launch(int32_t M);
int rato_car_eval(const rato_car_params* p, const float* us, const float* dW,
                  const float* x0_ped, const float* w_speed, const float* w_rep,
                  float* ego_scratch, float* Z, float* xs, float* g, void* stream);
/* K control sequences in one call: see rato_drone_eval_batch.  us [K][S][2]. */
int rato_car_eval_batch(const rato_car_params* p, int32_t K, const float* us, const float* dW, const float* x0_ped,
                        const float* w_speed, const float* w_rep, float* Z, int64_t ldz, double alpha, float thr,
                        void* workspace, size_t workspace_bytes, double* stats_out, void* stream);

/* Model.separation_distances_at_all_times on GIVEN trajectories (driving.py:223-236):
 *   xs [S+1][8][M] -> dist [S][M] = ||p_ego(t+1) - p_ped(t+1)|| - d_min   (the constraint value is -dist). */
int rato_car_separation_distances(const rato_car_params* p, const float* xs, float* dist, void* stream);

/* Kernel variants of rato_car_linearize: cols_per_thread = -1 row-parallel adjoint kernel (default;
 * needs ~20*64*S bytes of LDS <= 160 KB), 4/8/16 forward column kernel.  Resolves 0 to a concrete
 * value, reports the TILE of the G layout and returns the number of sample blocks (<0: bad arguments). */
int rato_car_linearize_plan(int32_t M, int32_t S, int32_t* cols_per_thread, int32_t* tile);

/*
 * Replaces vmap(Model.get_all_constraints_coeffs) (driving.py:260-307).
 * outputs:
 *   G        [n_tiles][n_pairs][2 controls][TILE]   d g_t / d u[s,i] for s<t
 *   g_up     [S][M]
 *   Z        [M] or NULL
 *   final_du [4][2S]   d x_S[0:4] / d u  (sample independent, so already the mean; :311)
 *   final_rhs[4]       -v_final + v_final_du.u (:288)
 */
int rato_car_linearize(const rato_car_params* p, const float* us, const float* dW,
                       const float* x0_ped, const float* w_speed, const float* w_rep,
                       float* ego_scratch, float* G, float* g_up, float* Z,
                       float* final_du, float* final_rhs,
                       int32_t cols_per_thread, void* stream);

/* ----------------------------------------------------------------- hopper */

/*
 * Replaces the sample-dependent part of Model.slip_risk_constraints
 * (hopper.py:300-367), friction_at_px (:75-81), its Jacobian/Hessian slices
 * (jacrev/hessian at :569,:577-580) and no_slip_constraints_verification
 * (:901-925).  C = number of contact steps ([0,time_jump) U [time_land,S)).
 *   px [C], fx [C], fz [C]        contact inputs gathered as at :305-311
 *   a, theta, tau [30][M]         friction-field features (SoA of (M,30))
 *   lam [C][M] or NULL            multipliers of the slip rows (for the Hessian sums)
 * outputs (any may be NULL):
 *   Z      [M]                    max_c (fx - mu fz)
 *   h      [C][M]                 fx - mu_i(px_c) fz
 *   dh_dfz [C][M]                 -mu_i(px_c)
 *   dh_dpx [C][M]                 -mu_i'(px_c) fz
 *   part_hess [nblocks][C][2]     per-block sums of lam*d2h/(dpx dfz), lam*d2h/dpx^2  (needs 8 C bytes of LDS per
 *                                 sample-wave: C <= 2032 contacts for M < 98,304 samples, 4064 above; RATO_EINVAL beyond)
 */
int rato_hopper_nblocks(int32_t M);
int rato_hopper_slip(int32_t M, int32_t C, const float* px, const float* fx, const float* fz,
                     const float* a, const float* theta, const float* tau, const float* lam,
                     float* Z, float* h, float* dh_dfz, float* dh_dpx, float* part_hess,
                     void* stream);

/* Same computation with px / fx / fz given as HOST arrays -- what the reference's IPOPT callbacks hold at every
 * iterate (hopper.py:305-311).  They are read during the call and travel in the kernel's argument block: no staging
 * buffer and no upload in front of the kernel (the upload was a third of the step at M = 5e4).  C must not exceed
 * RATO_HOPPER_MAX_HOST_CONTACTS (RATO_EINVAL beyond; use rato_hopper_slip with device arrays).  Under stream capture
 * the values are frozen into the captured launch: capture rato_hopper_slip if replays must see new inputs. */
#define RATO_HOPPER_MAX_HOST_CONTACTS 128
int rato_hopper_slip_host_inputs(int32_t M, int32_t C, const float* px_host, const float* fx_host,
                                 const float* fz_host, const float* a, const float* theta, const float* tau,
                                 const float* lam, float* Z, float* h, float* dh_dfz, float* dh_dpx,
                                 float* part_hess, void* stream);

/* rato_hopper_slip / rato_hopper_slip_host_inputs (host_inputs != 0) with THREE per-block sums per contact,
 *   part_hess3 [nblocks][C][3] = lam*d2h/(dpx dfz), lam*d2h/dpx^2, lam*dh/dpx
 * -- everything ``hessian(lambda . g)`` of the reference (hopper.py:575-580) needs from the samples: with the chain factors
 * J_c, H_c of the end-effector map p = x0 + x3 sin x2 (:166-171) the block of contact c on (x0, x2, x3) is
 * D2_c J_c J_c' + D0_c H_c and its mixed entries with fz are D1_c J_c (Model.slip_hessian). */
int rato_hopper_slip_hessian(int32_t M, int32_t C, const float* px, const float* fx, const float* fz,
                             int32_t host_inputs, const float* a, const float* theta, const float* tau,
                             const float* lam, float* Z, float* h, float* dh_dfz, float* dh_dpx, float* part_hess3,
                             void* stream);

/* ``jacrev(slip_risk_constraints)`` (hopper.py:569 on the rows of :300-367) in the reference's own layout: the CSC VALUE
 * array, written on the device from dh_dfz / dh_dpx [C][M] of the calls above.  Rows ('saa', saa != 0; :351-366):
 * 0 = (M alpha) t + sum y; 1 + i = -y_i; 1 + M + i C + c = h_ic - t - y_i - slack; one trailing zero row; ('baseline',
 * :339-348: i C + c = h_ic - slack).  Columns in Z order (:105-132).  Values, columns ascending and rows ascending
 * inside a column:  [c][x0, x2, x3][i] dh_dpx * chain[c][k];  [c][fx, fz][i] 1, dh_dfz;  saa: [i][1, -1, C x -1];
 * slack: M C x -1;  saa: t_risk: M alpha, M C x -1.  rato_hopper_jacobian_nnz values in all (8 C M + 2 M + 1 / 6 C M).
 *   chain_host [C][3] host floats (C <= RATO_HOPPER_MAX_HOST_CONTACTS: kernel arguments) and / or chain_dev (device);
 *   write_constants = 0 skips the constant part (a buffer written once keeps it).
 * Exact zeros (sin x2 = 0) are written as zeros; the reference's csc_matrix(dense) drops them (Model.slip_jacobian does). */
int64_t rato_hopper_jacobian_nnz(int32_t M, int32_t C, int32_t saa);
int rato_hopper_emit_jacobian_values(int32_t M, int32_t C, int32_t saa, double alpha, const float* dh_dfz,
                                     const float* dh_dpx, const float* chain_host, const float* chain_dev,
                                     int32_t write_constants, float* out, void* stream);

/* --------------------------------------------------------------- assembly */

/*
 * Device side of the sparse QP assembly (replaces the dense packing loop of
 * drone_risk.py:339-364 / driving.py:344-363 and the dense->csr scan of :419).
 * Emits the linearized-constraint block of the CSC value array in the
 * reference's order: for s = 0..S-2, for g = 0..n_g-1 (u-column s*n_u + g), for
 * sample i, for row-group r (obstacle), for t = s+1..S-1:
 *     out[...] = scale * d row(r,t) / d u[s,g]
 * i.e. column (s,g) occupies M*R*(S-1-s) consecutive values.  G is the packed
 * tile-blocked Jacobian of the linearize calls (tile = its TILE, n_g = 2,
 * R = 3 drone / 1 driving).  scale = the reference's MULTIPLIER (0.01 drone, 1
 * driving), times 1e-7 while scp_iter < 2 (drone_risk.py:413-415).
 * The transposition is staged through 64 (R (S-1) + 1) floats of LDS: RATO_EINVAL when that exceeds 160 KB
 * (S > 213 for the drone, S > 639 for driving) -- the facades then assemble on the host from the untiled Jacobian
 * (Model.get_constraints_coeffs_host), same pattern and values.
 */
int rato_emit_csc_values(const float* G, const float* W /* NULL, or the factor of a factored G */,
                         int64_t ld /* row stride of W */, int32_t tile, int32_t n_g, int32_t R, int32_t S,
                         int64_t M, float scale, float* out, void* stream);

/* ------------------------------------------- linearized CVaR constraint oracle */

/*
 * Eliminating the auxiliary y_i of the reference's QP (drone_risk.py:327-368:
 * y_i >= -slack, y_i >= (G_i u - g_up_i)_r - t for every row r) leaves the single
 * convex constraint  alpha*M*CVaR_alpha(m(u)) - (M*(1-alpha) - 1)*slack <= 0  with
 *     m_i(u) = max_r [ (G_i u)_r - g_up_{i,r} ].
 * These calls are the device oracle a cutting-plane solve needs per cut (selection of the tail: rato_risk_stats on
 * m; the cut itself: tail-weighted sums of the arg-max rows and of their offsets).  The rows are evaluated as
 *     (G_i x)_r + sign * base_{i,r}
 * which covers the reference's expression (x = u, base = g_up, sign = -1) and its delta form (x = u - u_k,
 * base = g, sign = +1: g_up = -g + G u_k, params.rows_out = 1 makes the linearize kernels write g).  Inputs are the
 * fp32 arrays of the linearize calls; ALL arithmetic is fp64 (x and the partial sums are double): value and gradient
 * of a cut agree to 1e-13, which is what the 1e-5 parity of SCP iterates with the full QP rests on.
 *
 * rato_saa_rowmax: one streaming read of the packed Jacobian G (layout of the
 * linearize calls; tile = its TILE; R = 3 drone / 1 driving; rows of sample i
 * indexed r*S + t):  m_out[i], arg_out[i] = max (rounded to fp32) / arg-max (smallest row index on ties).
 *   base [R][S][ld], xs [S][n_u] doubles (only controls 0 and 1 enter the rows).
 */
int rato_saa_rowmax(const float* G, const float* W /* NULL, or the factor of a factored G (R = 3) */,
                    int32_t tile, int32_t R, int32_t S, int64_t M, int64_t ld,
                    const float* base, double sign /* +1 or -1 */, const double* xs, int32_t n_u,
                    float* m_out, int32_t* arg_out, void* stream);

/*
 * The same m_out / arg_out for the drone WITHOUT reading the Jacobian: (G_i x)_{j,t} =
 * sum_a W[j,t,a] dp_a(t+1), where dp is the response of the linearized dynamics
 * d x_{k+1} = A_k d x_k + B x_k (d x_0 = 0, B = [0, dt/m]^T, A_k from A22: see rato_drone_linearize).
 * One pass over A22, W, base (11 S floats per sample instead of S(S-1) + 9 S).  xs [S][3] doubles (n_u = 3);
 * p supplies M, ld, S, dt, kp.
 */
int rato_drone_rowmax_implicit(const rato_drone_params* p, const float* mass, const float* A22,
                               int32_t a22_axes /* 2: [S][2][ld] of rato_drone_linearize (a22); 3: [S][3][ld] of
                                                   rato_drone_linearize_generators (1 - a22) */,
                               const float* W, const float* base, double sign, const double* xs,
                               float* m_out, int32_t* arg_out, void* stream);

/*
 * Generators-only linearization (drone): A22 [S][3 axes][ld] -- holding the COMPLEMENT 1 - a22 = dt (k_d + 2 c_d |v|) / m
 * (~1e-3: as an fp32 number it is exact to 1e-10 of a22; the consumers below rebuild a22 in fp64; the [S][2][ld] table
 * of rato_drone_linearize holds a22 itself, and a22_axes tells the two apart) --, W [3 obs][S][2][ld], g_up [3 obs][S][ld] (or g:
 * p->rows_out), Z [M] or NULL, part [ceil(M/256)][6S+6] (per-block sums, layout of rato_drone_linearize) -- the whole
 * linearization in 12 S numbers per sample and NO Jacobian entries: Phi[t,s,a] = e_0' A_t ... A_{s+1} B is
 * regenerated from A22 by the consumers (rato_drone_rowmax_implicit for G.x, rato_drone_tail_rows_implicit for rows
 * of G).  60 B per sample-step of HBM traffic instead of 245 B; what a reduced SCP iteration needs
 * (Model.solve_reduced).  W and g_up may both be NULL (a caller whose cut oracle re-runs the rollout --
 * rato_drone_*_rollout -- needs only the sample sums; A22 stays: it is the lane's scratch for the final-state adjoint).
 */
int rato_drone_linearize_generators(const rato_drone_params* p, const float* us, const float* dW,
                                    const float* mass, const float* Qsym,
                                    float* A22, float* W, float* g_up, float* Z, float* part, void* stream);

/*
 * Cuts under a linearization (the new cut of an oracle call, and cut recycling across SCP iterations).  A cut is a
 * tail weighting w (from the m values and the rato_risk_stats record that was computed on them: 1 above the
 * threshold out[10], lambda = clamp((alphaM - #{m > t}) / #{m == t}, 0, 1) on ties) plus the arg-max row of every
 * sample; it is a valid cut under ANY linearization,
 *     CVaR(m(x)) >= (1/(alpha M)) sum_i w_i [(G_i x)_{r_i} + sign base_{i,r_i}],
 * tight at the x it was computed for under the linearization it was computed with.
 * K cuts kept in rings  m_base [slot][M], arg_base [slot][M], stats_base [slot][stats_stride >= 11 doubles]  are
 * evaluated in one launch (slots: device array of K ring slots; NULL: K = 1, the pointers are the slot):
 *   part[blk][k][0 .. 2(S-1))  block sums of w_i G_i[r_i, (s,g)]      (doubles)
 *   part[blk][k][2(S-1)]       block sum  of w_i base_{i,r_i}
 * Reduce with rato_sum_partials_f64(part, nblk, K * (2(S-1) + 1), ...).
 */
int rato_saa_tail_rows_batch(const float* G, const float* W /* NULL or factor */, int64_t ld, int32_t tile,
                             int32_t R, int32_t S, int64_t M, const float* base,
                             const float* m_base, const int32_t* arg_base, const double* stats_base,
                             int64_t stats_stride, const int32_t* slots, int32_t K, double alphaM, double* part,
                             void* stream);

/*
 * The same for the drone without reading the Jacobian: the arg-max row of every tail sample is regenerated from A22
 * (adjoint sweep from its t*, fp64).  Same part layout.
 */
int rato_drone_tail_rows_implicit(const rato_drone_params* p, const float* mass, const float* A22, int32_t a22_axes,
                                  const float* W, const float* base,
                                  const float* m_base, const int32_t* arg_base, const double* stats_base,
                                  int64_t stats_stride, const int32_t* slots, int32_t K, double alphaM,
                                  double* part, void* stream);

/*
 * Table-free ("rollout") form of the drone oracle, delta form only:  rows(u) = g(u_k) + grad g(u_k) . (u - u_k).
 * Instead of reading what a linearize call stored of the rollout at u_k in fp32 (A22, W, g: 44 S bytes per sample, each
 * number rounded to 6e-8 of its own magnitude), these two calls RE-RUN that rollout in fp64 from the samples themselves
 * (dW rows of the two horizontal axes, mass, Qsym: 8 S + 40 bytes per sample) while they propagate the response to
 * xs = u - u_k (rowmax) / the adjoint of the arg-max rows (tail rows): 5 x less HBM traffic, and nothing of the fp32
 * device path is left in the rows but the rounding of its inputs.  uk, xs: doubles [S][3].  Outputs as
 * rato_drone_rowmax_implicit / rato_drone_tail_rows_implicit (the offset sum is sum_i w_i g_{i,r_i}: sign = +1).
 */
int rato_drone_rowmax_rollout(const rato_drone_params* p, const double* uk, const float* dW, const float* mass,
                              const float* Qsym, const double* xs, float* m_out, int32_t* arg_out, void* stream);
int rato_drone_tail_rows_rollout(const rato_drone_params* p, const double* uk, const float* dW, const float* mass,
                                 const float* Qsym, const float* m_base, const int32_t* arg_base,
                                 const double* stats_base, int64_t stats_stride, const int32_t* slots, int32_t K,
                                 double alphaM, double* part, void* stream);

/*
 * The same for the driving problem (R = 1 row per step: g_t = -(|p_ego(t+1) - p_ped(t+1)| - d_min), driving.py:223-230,
 * :260-313).  The ego trajectory at u_k, its tangent along xs and the tables of the adjoint are sample independent: every
 * workgroup folds them in fp64 from uk / xs (doubles [S][2]); the pedestrian is re-rolled per sample from dW [S][2][M],
 * x0_ped [4][M], w_speed [M], w_rep [M] (8 S + 24 bytes per sample against the 6240 of the packed Jacobian at S = 40).
 * m_out / arg_out, part layout and reduction as above (2(S-1) gradient sums: u_s enters g_t for s <= t - 1).
 */
int rato_car_rowmax_rollout(const rato_car_params* p, const double* uk, const float* dW, const float* x0_ped,
                            const float* w_speed, const float* w_rep, const double* xs, float* m_out,
                            int32_t* arg_out, void* stream);
int rato_car_tail_rows_rollout(const rato_car_params* p, const double* uk, const float* dW, const float* x0_ped,
                               const float* w_speed, const float* w_rep, const float* m_base, const int32_t* arg_base,
                               const double* stats_base, int64_t stats_stride, const int32_t* slots, int32_t K,
                               double alphaM, double* part, void* stream);

/*
 * One oracle round trip of a cutting-plane solve in ONE call (table-free forms; system 0 = drone: s0..s2 = dW, mass,
 * Qsym, s3 unused; 1 = driving: s0..s3 = dW, x0_ped, w_speed, w_rep):
 *   x_host [S][n_u] doubles (pinned for an asynchronous copy) -> x_dev;  m_out / arg_out = rato_*_rowmax_rollout;
 *   res_dev[0..11) = rato_risk_stats(m_out, alpha, thr);  S > 1: part_dev = rato_*_tail_rows_rollout (K = 1, record =
 *   res_dev, stride 11 + nc), res_dev[11..11+nc) = its column sums, nc = 2(S-1) + 1;  res_dev -> res_host (PINNED,
 *   device-visible host memory: the last launch writes it directly, no copy node follows);  the call RETURNS WHEN THE
 *   RECORD HAS ARRIVED in res_host (see the conventions at the top: res_host is pre-set and watched; RATO_CUT_POLL=0:
 *   hipStreamSynchronize).  Replaces two copies, four calls and a synchronize of the host loop
 *   (drone_risk.py:425-469 hands the whole QP to OSQP; here every cut of the reduced subproblem is one such trip).
 */
int rato_cut_oracle_rollout(int32_t system, const void* params, const double* uk, const float* s0, const float* s1,
                            const float* s2, const float* s3, const double* x_host, double* x_dev, float* m_out,
                            int32_t* arg_out, double alpha, float thr, double alphaM, void* workspace,
                            size_t workspace_bytes, double* res_dev, double* part_dev, double* res_host, void* stream);

/*
 * Sums of the matrix-free KKT certificate of a reduced solution against the reference's full QP (drone_risk.py:327-368,
 * driving.py:330-373; riskaversetrajopt_amd/certificate.py): m_star [M] = the m values at the solution (a rowmax call),
 * cut k < K = ring slot slots[k] of (m_base, stats_base) with multiplier lam[k] (device doubles), v = t_risk - slack.
 * part [ceil(M/256)][2K + 2] doubles per block:  [k] sum_i w_ki (m_i* - v)^+;  [K + k] sum_i w_ki;  [2K] sum_i (m_i* - v)^+;
 * [2K + 1] max_i sum_k lam_k w_ki  (w_k: the tail weighting of cut k, as in rato_saa_tail_rows_batch).
 */
int rato_kkt_sums(const float* m_star, int64_t M, const float* m_base, const double* stats_base, int64_t stats_stride,
                  const int32_t* slots, const double* lam, int32_t K, double alphaM, double v, double* part, void* stream);

/* ------------------------------------------------- the cutting-plane loop of one SCP subproblem (host, csrc/cutloop.hip)
 *
 * The reference hands every SCP subproblem to OSQP (drone_risk.py:425-469, driving.py:423-456).  Here the subproblem is
 * that QP reduced exactly to (u, slack) and solved by Kelley cuts on the linearized CVaR constraint: master QP on the
 * host (rato_master_*), cut oracle on the device (rato_cut_oracle_rollout, the table-free forms).  rato_cut_solve is
 * that whole loop -- master, lazily entering control bounds, "the last evaluated cut joins the master", the keep rule
 * for cuts recycled into the next subproblem, the multipliers -- as ONE call (statement for statement the loop of
 * riskaversetrajopt_amd/cvar_cuts.py::CvarCutSolver._solve, which remains for sharded batches and the table forms; both
 * produce bitwise the same iterates).
 *
 * rato_cut_config: sizes, constants and the CALLER-OWNED buffers of a solver (nothing is allocated on the device here):
 *   system 0 drone (n_u 3, params = rato_drone_params*, s0..s2 = dW, mass, Qsym), 1 driving (n_u 2, rato_car_params*,
 *   s0..s3 = dW, x0_ped, w_speed, w_rep); params is copied.  nU = n_u S, n = nU + 1 (u, slack), nc = 2(S-1) + 1,
 *   nres = RATO_N_STATS + nc, nblk = ceil(M / 256).
 *   device: uk_dev, x_dev [nU] doubles; ring_m [cap][M] floats, ring_arg [cap][M] int32, ring_res [cap][nres] doubles
 *   (slot cap-1 is scratch); workspace of rato_risk_stats (initialised); part [nblk][nc], part_b [nblk][keep_max nc]
 *   doubles; slots_dev [keep_max] int32.
 *   pinned host (device-visible: the reductions write into them directly): uk_host, x_host [nU], res_host [nres],
 *   sums_b_host [keep_max nc] doubles, slots_host [keep_max] int32.
 *   p_diag, q [n]: the objective 1/2 z' diag(p_diag) z + q' z (copied).
 *   mode_saa 1: CVaR_alpha(m(u)) - c_s slack <= rhs0 with the row -slack <= 0; 0 ('baseline'): max_i m_i(u) <= rhs0.
 */
typedef struct {
  int32_t system, S, cap, keep_max, keep_recent, keep_idle, mode_saa, recycle;
  int64_t M;
  double alpha, alphaM, c_s, rhs0, u_min, u_max;
  float thr;
  const void* params;
  const float *s0, *s1, *s2, *s3;
  double *uk_dev, *uk_host, *x_host, *x_dev;
  float* ring_m;
  int32_t* ring_arg;
  double* ring_res;
  void* workspace;
  size_t workspace_bytes;
  double *part, *part_b, *sums_b_host;
  int32_t *slots_dev, *slots_host;
  double* res_host;
  const double *p_diag, *q;
} rato_cut_config;

/* What a solve returns.  Caller-provided arrays: us [nU]; cut_slot / cut_lambda [cut_capacity] (ring slot and
 * multiplier of every cut row of the last master: the data of a KKT certificate against the full QP); bound_var /
 * bound_sign / bound_lambda [bound_capacity] (the control bounds that entered: variable, +1 upper / -1 lower,
 * multiplier).  status 0 solved, 1 maximum cuts reached, 2 solved: the last cut no longer moved the master's solution
 * (violation <= 1e-7, step <= 1e-10: what is left is the accuracy of the master's own NNLS). */
typedef struct {
  double* us;
  double slack, t_risk, phi, oracle_s, master_s, lam_slack;
  int32_t cuts, recycled, status, uncertified_cuts;
  int32_t* cut_slot;
  double* cut_lambda;
  int32_t cut_capacity, n_cut_rows;
  int32_t* bound_var;
  double* bound_sign;
  double* bound_lambda;
  int32_t bound_capacity, n_bounds;
} rato_cut_result;

typedef struct rato_cut_solver rato_cut_solver;
int rato_cut_solver_create(rato_cut_solver** out, const rato_cut_config* cfg);
void rato_cut_solver_destroy(rato_cut_solver* s);
size_t rato_cut_config_bytes(void); /* sizeof the two structs as this library was built: a binding checks its layout */
size_t rato_cut_result_bytes(void);

/* Stream-ordered prologue of a subproblem, NO synchronisation: u_lin [nU] (host) -> uk_dev; for the n_keep ring slots
 * kept from the previous subproblem, their tail-row sums under the new linearization point, reduced into sums_b_host. */
int rato_cut_begin(rato_cut_solver* s, const double* u_lin, const int32_t* keep, int32_t n_keep, void* stream);

/* The "define" half of a reduced SCP iteration of the drone (system 0) as one call: us [S][3] doubles (host) -> us_dev
 * (through the pinned us_host), rato_drone_linearize_generators at them without tables (A22 [S][3][ld]: kernel scratch;
 * Z [z_floats >= M] or NULL; part [ceil(M/256)][6S+6]), the sample sums reduced straight into sums_host (pinned, 6S+6 doubles),
 * the non-finite count of Z and part (bad_dev / bad_host pinned, or both NULL), rato_cut_begin(us, keep, n_keep).  WAITS
 * for the sample sums only (watched in sums_host; with the non-finite count: an event behind the count's copy): the kept
 * cuts' sums may still be in flight when it returns -- follow with rato_cut_solve(kept_in_flight = 1), which builds the
 * master first and then waits for them (or synchronise the stream yourself before reading sums_b_host).
 * RATO_ENONFINITE when the count is not zero. */
int rato_cut_define_drone(rato_cut_solver* s, const double* us, float* us_host, float* us_dev, float* A22, float* Z,
                          int64_t z_floats, float* part, double* sums_host, uint32_t* bad_dev, uint32_t* bad_host,
                          const int32_t* keep, int32_t n_keep, void* stream);

/* The loop.  final_du (n_c x nU, row-major), final_rhs (n_c): the equality rows;  u_lin: the linearization point (the
 * one given to rato_cut_begin);  with_cvar 0: the reference's relaxed first iterations (no CVaR rows: one master solve);
 * tol: a cut is added while CVaR - c_s slack - rhs0 > tol, at most max_cuts; a last cut with violation in
 * (final_cut_above, tol] still joins the master, which is solved once more.  keep / keep_idle_count [keep_max] and
 * *n_keep_io: in -- the slots kept from the previous subproblem (+ for how many solves each has carried no multiplier);
 * out -- the same for the next one.  kept_in_flight 1: rato_cut_begin was called for these slots on this stream.
 * SYNCHRONISES the stream (every oracle round trip is read by the host master).  Returns RATO_OK, RATO_ERANK,
 * RATO_EINFEASIBLE, RATO_ENNLS, RATO_ESELECT, RATO_ENONFINITE (check_finite: an oracle call saw non-finite m values). */
int rato_cut_solve(rato_cut_solver* s, const double* final_du, const double* final_rhs, int32_t n_c, const double* u_lin,
                   int32_t with_cvar, double tol, int32_t max_cuts, double final_cut_above, int32_t check_finite,
                   int32_t* keep, int32_t* keep_idle_count, int32_t* n_keep_io, int32_t kept_in_flight,
                   rato_cut_result* out, void* stream);

/* The reduced SCP of the drone as ONE call (drone_risk.py:519-532 with the timing protocol of drone_times.py:509-550):
 * `iters` iterations of [rato_cut_define_drone at the current controls -> equality rows from the sample sums ->
 * rato_cut_solve], starting from us0 [S][3]; iteration k < first_cvar runs without the CVaR rows (:413-417).  Every
 * iteration is timed from its first instruction to the moment its solution is on the host; the stream is synchronised
 * once, inside the last iteration's clock.  us_hist [iters][S][3] and rec [iters] (host) receive every iteration's
 * solution and record; the define's buffers are those of rato_cut_define_drone (no Z, no non-finite scan: check_finite
 * tests the sample sums and the oracle's m values); keep / keep_idle_count / n_keep_io as for rato_cut_solve.  Returns the
 * first non-OK status (RATO_ERANK / RATO_ESELECT: repeat with the per-iteration calls), *done = iterations completed. */
typedef struct rato_scp_iter {
  double define_s, solve_s, oracle_s, master_s; /* solve = oracle round trips + master; define = the rest of the iteration */
  double t_risk, slack, phi;
  int32_t cuts, status, recycled, reserved;
} rato_scp_iter;
size_t rato_scp_iter_bytes(void);
int rato_scp_run_drone(rato_cut_solver* s, const double* us0, int32_t iters, int32_t first_cvar, double tol,
                       int32_t max_cuts, double final_cut_above, int32_t check_finite, float* us_host, float* us_dev,
                       float* A22, float* part, double* sums_host, int32_t* keep, int32_t* keep_idle_count,
                       int32_t* n_keep_io, double* us_hist, rato_scp_iter* rec, int32_t* done, void* stream);

/* ------------------------------------------------------------ device sampler */

/*
 * Device-side sampler: Philox4x32-10 (Salmon et al., SC'11), counter = (sample m, step / row t, stream
 * id), key = seed; 24-bit uniforms, Box-Muller normals (csrc/philox.h).  The value at (seed, stream, t, m) is a pure
 * function of those numbers: the same batch can be MATERIALISED (rato_*_sample) or REGENERATED inside the rollout
 * kernels (rato_*_eval_philox), bit for bit.  These replace, for synthetic / Monte-Carlo batches that never leave
 * HBM, the host loops of drone_utils.py:61-93, driving.py:84-120, hopper.py:70-74 (same distributions; the host
 * samplers of the Python facades replay the reference's MT19937 stream when identical draws are wanted).
 *
 * Generic fills, layout out[T][C][ld] (C <= 4 values per Philox call), used by the tests and as utilities:
 *   rato_philox_u32      the raw 4 words            out[t][0..3][m]
 *   rato_philox_normal   scale[k] * N(0,1) + mean[k]           (scale / mean: HOST float[C], NULL = 1 / 0)
 *   rato_philox_uniform  low[k] + width[k] * U(0,1)
 */
int rato_philox_u32(uint32_t* out, int32_t T, int64_t M, int64_t ld, uint64_t seed, uint32_t stream_id, void* stream);
int rato_philox_normal(float* out, int32_t T, int32_t C, int64_t M, int64_t ld, uint64_t seed, uint32_t stream_id,
                       const float* scale, const float* mean, void* stream);
int rato_philox_uniform(float* out, int32_t T, int32_t C, int64_t M, int64_t ld, uint64_t seed, uint32_t stream_id,
                        const float* width, const float* low, void* stream);

/* drone_utils.py:61-93: dW [S][3][ld] = sqrt(sampler_dt) N(0,1) (velocity rows), mass [ld] ~ U(nom -+ delta),
 * Qsym [3][3][ld] from semi-axes obs_radii[j] + U(-+ obs_radii_delta) per dimension (HOST float[3]).  dW may be NULL
 * (noise regenerated in rato_drone_eval_philox); mass and Qsym may both be NULL. */
int rato_drone_sample(int64_t M, int64_t ld, int32_t S, float sampler_dt, uint64_t seed, float mass_nom,
                      float mass_delta, const float* obs_radii, float obs_radii_delta, float* dW, float* mass,
                      float* Qsym, void* stream);
/* rato_drone_eval with the noise of rato_drone_sample(seed, sampler_dt) regenerated in the kernel (no dW array). */
int rato_drone_eval_philox(const rato_drone_params* p, const float* us, uint64_t seed, float sampler_dt,
                           const float* mass, const float* Qsym, float* Z, float* xs, float* g, void* stream);
/* rato_drone_linearize (row-parallel kernel, cols_per_thread = -1: S <= 126) with the noise of
 * rato_drone_sample(seed, sampler_dt) regenerated while a tile is staged: bit for bit the outputs of rato_drone_linearize
 * on the materialised dW, without the array and without its reads in the middle of the store stream (the reads are
 * 2 % of the bytes of the products output but cost 3-8 % of the kernel's time: DESIGN.md 4.1). */
int rato_drone_linearize_philox(const rato_drone_params* p, const float* us, uint64_t seed, float sampler_dt,
                                const float* mass, const float* Qsym, float* G, float* W, float* A22, float* g_up,
                                float* Z, float* part, void* stream);

/* The row-parallel kernels read a tile's noise while the tiles of other workgroups are being stored, and reads beside a
 * saturated store stream cost more than their bytes (the noise regenerated in the kernel: -7 % at the metric
 * configuration).  A batch that is linearized again and again is re-tiled ONCE, so that a tile's 3S rows are one
 * contiguous block instead of 3S rows of 256 B that lie ld floats apart:
 *   rato_drone_tiled_noise_floats(M, S)              floats of the tiled copy: ceil(M / 64) * 3S * 64
 *   rato_drone_tile_noise(dW, M, ld, S, dW_tiled)    dW [S][3][ld] -> dW_tiled [ceil(M/64)] blocks of 3S * 64 floats, each the
 *                                                    image the kernel keeps in LDS: [S][64] (xi_x, xi_y) pairs, then [S][64]
 *                                                    xi_z (lanes beyond M: 0)
 *   rato_drone_linearize_tiled(...)                  rato_drone_linearize(cols_per_thread = -1) reading dW_tiled: the
 *                                                    same outputs, bit for bit (see also rato_car_linearize_tiled) */
size_t rato_drone_tiled_noise_floats(int64_t M, int32_t S);
int rato_drone_tile_noise(const float* dW, int64_t M, int64_t ld, int32_t S, float* dW_tiled, void* stream);
int rato_drone_linearize_tiled(const rato_drone_params* p, const float* us, const float* dW_tiled, const float* mass,
                               const float* Qsym, float* G, float* W, float* A22, float* g_up, float* Z, float* part,
                               void* stream);

/* driving.py:84-120: dW [S][2][M], x0_ped [4][M] = x0_mean + x0_std * N(0,1) (HOST float[4] each), w_speed, w_rep [M]
 * ~ U(nom -+ del).  dW may be NULL; the three parameter arrays may all be NULL. */
int rato_car_sample(int64_t M, int32_t S, float sampler_dt, uint64_t seed, float w_speed_nom, float w_speed_del,
                    float w_rep_nom, float w_rep_del, const float* x0_mean, const float* x0_std, float* dW,
                    float* x0_ped, float* w_speed, float* w_rep, void* stream);
int rato_car_eval_philox(const rato_car_params* p, const float* us, uint64_t seed, float sampler_dt,
                         const float* x0_ped, const float* w_speed, const float* w_rep, float* ego_scratch, float* Z,
                         float* xs, float* g, void* stream);
/* rato_car_linearize (row-parallel kernel, cols_per_thread = -1) with the noise of rato_car_sample(seed, sampler_dt)
 * regenerated while a tile is staged: bit for bit the outputs of rato_car_linearize on the materialised dW. */
int rato_car_linearize_philox(const rato_car_params* p, const float* us, uint64_t seed, float sampler_dt,
                              const float* x0_ped, const float* w_speed, const float* w_rep, float* ego_scratch,
                              float* G, float* g_up, float* Z, float* final_du, float* final_rhs, void* stream);

/* The row-parallel driving kernel reads a tile's noise -- 2S rows of 64 samples -- while the tiles of other workgroups are
 * being stored; as 2S rows of 256 B that lie M floats apart these reads cost the launch 6-11 % (5 % more bytes; reads and
 * writes share the HBM bus), as ONE contiguous block per tile a third less.  A batch that is linearized
 * again and again (every SCP iteration) is therefore re-tiled ONCE:
 *   rato_car_tiled_noise_floats(M, S)           floats of the tiled copy: ceil(M / 64) * 2S * 64
 *   rato_car_tile_noise(dW, M, S, dW_tiled)     dW [S][2][M] -> dW_tiled [ceil(M/64)] blocks of 2S * 64 floats, each the image
 *                                               the kernel keeps in LDS: [S][64] (xi_0, xi_1) pairs (lanes beyond M: 0)
 *   rato_car_linearize_tiled(...)               rato_car_linearize(cols_per_thread = -1) reading dW_tiled: the same
 *                                               outputs, bit for bit */
size_t rato_car_tiled_noise_floats(int64_t M, int32_t S);
int rato_car_tile_noise(const float* dW, int64_t M, int32_t S, float* dW_tiled, void* stream);
int rato_car_linearize_tiled(const rato_car_params* p, const float* us, const float* dW_tiled, const float* x0_ped,
                             const float* w_speed, const float* w_rep, float* ego_scratch, float* G, float* g_up, float* Z,
                             float* final_du, float* final_rhs, void* stream);

/* hopper.py:70-74: a = 0.025 sqrt(2/30) U(0,1), theta = pi U(0,1), tau = 2 pi U(0,1), each [30][M]. */
int rato_hopper_sample(int64_t M, uint64_t seed, float* a, float* theta, float* tau, void* stream);

/* ---------------------------------------------------------------- multi-GPU */

/*
 * The path shards over samples; the only exchange of an evaluation is ONE all-gather (RCCL) of each rank's record
 *   [ fp64 partial sums (n_sums) | fp32 Z row (>= M_local floats) ]      (rec_bytes each, a multiple of 8).
 * rato_unpack_records turns the gathered buffer (world records) into  Z_all [world * M_local]  (rank order) and
 * total [n_sums] = the partial sums added in rank order (bitwise identical on every rank), in one launch.
 */
int rato_unpack_records(const void* all, int32_t world, int32_t n_sums, int64_t M_local, int64_t rec_bytes,
                        double* total, float* Z_all, void* stream);

/*
 * The collective itself (saa_comm_init / exchange / destroy).  The reference is one process
 * (drone_risk.py:18); with the sample axis sharded over one process per GPU, the mean of drone_risk.py:294-296 and the
 * statistics of :663-695 / drone_main_plot.py:640-652 need every rank's record.  RCCL (librccl, bound at run time)
 * over xGMI; one communicator per (process, GPU); the caller selects the device (hipSetDevice) before rato_comm_init.
 *   rato_comm_unique_id   rank 0 only: 128 opaque bytes (ncclUniqueId) that the host ships to the other ranks by
 *                         any means it has (the Python facade: one torch.distributed broadcast)
 *   rato_comm_init        collective over all `world` ranks; *comm is the opaque handle
 *   rato_comm_allgather   stream-ordered all-gather of `bytes` bytes per rank into recv_all [world * bytes]
 *   rato_comm_exchange    the whole exchange of an evaluation: all-gather of the record + rato_unpack_records
 *   rato_comm_destroy     collective teardown
 * Errors: RATO_ENOCOMM (no librccl), RATO_ERCCL - ncclResult_t.
 */
#define RATO_COMM_ID_BYTES 128
typedef struct rato_comm rato_comm;
/* RATO_OK if librccl can be bound in this process, RATO_ENOCOMM otherwise: local and cheap, no communication.  Ranks
 * agree on it BEFORE any of them enters the collective rato_comm_init (a rank that cannot bind the library would
 * return at once and leave the others blocked inside ncclCommInitRank). */
int rato_comm_available(void);
int rato_comm_unique_id(void* id_out /* host, RATO_COMM_ID_BYTES */);
int rato_comm_init(rato_comm** comm, const void* id_bytes /* host */, int32_t rank, int32_t world);
int rato_comm_world(const rato_comm* comm);
int rato_comm_rank(const rato_comm* comm);
int rato_comm_allgather(rato_comm* comm, const void* send, void* recv_all, int64_t bytes, void* stream);
int rato_comm_exchange(rato_comm* comm, const void* record, void* all, int64_t rec_bytes, int32_t n_sums,
                       int64_t M_local, double* total, float* Z_all, void* stream);
int rato_comm_destroy(rato_comm* comm);

/* ----------------------------------------------------- host: the master QP */

/*
 * HOST function (no device work): Lawson-Hanson non-negative least squares  min |A y - b|, y >= 0  started from a
 * guess of the passive set, with an incrementally updated QR of the passive columns (csrc/nnls.hip).  It is the inner
 * solver of the cutting-plane master QP (riskaversetrajopt_amd/dense_qp.py) -- the counterpart of the C solver the
 * reference hands its subproblem to (osqp, drone_risk.py:433-457).  A: m x n COLUMN-major; passive: n bytes in/out;
 * y: n doubles out; maxiter <= 0: 3 n + 10.  Returns 1 converged (KKT test of the original algorithm), 0 not, < 0 bad
 * arguments.
 */
int rato_nnls_warm(const double* A, int32_t m, int32_t n, const double* b, uint8_t* passive, double* y,
                   int32_t maxiter);

/*
 * The master QP of the cutting-plane loop as an object that lives across the cuts of one SCP subproblem (host code,
 * csrc/master.hip):   min 1/2 z'Pz + q'z   s.t.  A_eq z = b_eq,  rows_j . z <= rhs_j  (rows appended one at a time),
 * P = diag(p_diag) > 0, A_eq (m_eq x n, row-major) of full row rank (RATO_EINVAL otherwise: the Python facade then takes
 * its general NumPy path).  Equalities are eliminated and the Hessian whitened once, in O(n m_eq^2); a row costs
 * O(n m_eq); a solve is one warm-started rato_nnls_warm on the rows so far.
 *   rato_master_solve -> 1: z (n doubles) and lam (one multiplier >= 0 per row);  0: NNLS did not converge;
 *                        RATO_EINFEASIBLE.
 */
typedef struct rato_master rato_master;
int rato_master_create(rato_master** out, int32_t n, const double* p_diag, const double* q, int32_t m_eq,
                       const double* A_eq, const double* b_eq);
int rato_master_add_rows(rato_master* m, int32_t k, const double* A /* k x n, row-major */, const double* b);
int rato_master_solve(rato_master* m, double* z, double* lam);
int32_t rato_master_rows(const rato_master* m);
void rato_master_destroy(rato_master* m);

/* ------------------------------------------------------------- statistics */

/* Deterministic second stage of the sample mean (drone_risk.py:294-296,
 * driving.py:311-313): out[c] = scale * sum_b part[b][c], accumulated in fp64
 * in a fixed order.  out is double[ncols]. */
int rato_sum_partials(const float* part, int32_t nblocks, int32_t ncols, double scale,
                      double* out, void* stream);
/* the same for double partials (the fp64 block sums of the CVaR-cut oracle) */
int rato_sum_partials_f64(const double* part, int32_t nblocks, int32_t ncols, double scale,
                          double* out, void* stream);

/* Failure detection (the reference has none: an infeasible QP only prints, drone_risk.py:458-459):
 * counts the NaN/Inf entries of a device array into *count (device uint32). */
int rato_count_nonfinite(const float* x, int64_t n, uint32_t* count, void* stream);

/* Same scan ACCUMULATING into *count (no reset): chain several arrays (g_up, Z, partial sums ...) into one counter,
 * zeroed by the caller (or by one rato_count_nonfinite), then read it once.  The facades' ``check_finite=True``
 * turns a non-zero count into RATO_ENONFINITE / RatoNonFiniteError. */
int rato_count_nonfinite_acc(const float* x, int64_t n, uint32_t* count, void* stream);

/* Workspace of rato_risk_stats for M samples: caller-owned device memory of this many bytes, set up ONCE with
 * rato_risk_stats_init (zero fill + a tag) before its first use; one workspace per stream that computes statistics
 * concurrently.  Every call leaves it ready for the next one (the histograms are re-zeroed by the last launch).  A
 * workspace that was never initialised makes the multi-launch path write NaN into all of `out` instead of numbers
 * computed from garbage. */
size_t rato_risk_stats_workspace_bytes(int64_t M);
int rato_risk_stats_init(void* workspace, size_t workspace_bytes, void* stream);

/*
 * Replaces the Monte-Carlo statistics: fraction satisfied (drone_risk.py:661,719),
 * empirical VaR (drone_main_plot.py:640-652: sort(Z)[M - floor(alpha M) - 1])
 * and AVaR/CVaR (drone_risk.py:663-695; the OSQP LP there is replaced by exact
 * selection of the Rockafellar-Uryasev minimiser followed by the closed form :694).
 *   Z [M]; thr = 1e-6 (the B_satisfied threshold)
 *   out: double[RATO_N_STATS] = { VaR, CVaR, fraction(Z<=thr), mean(Z), max(Z),
 *                       count(Z<=thr), sum(max(Z-t,0)), k (selected ascending rank),
 *                       #{Z > t}, #{Z == t}, t }
 *        t = the Rockafellar-Uryasev minimiser = sort(Z)[k]; VaR = t except when floor(alpha M) == M, where the
 *        reference's index -1 wraps to max(Z) (NumPy negative index, drone_main_plot.py:651) while CVaR, the counts
 *        and every consumer of the threshold (rato_saa_tail_rows*) keep using t = out[10].
 * Launches: 1 for M <= 12,288 (one workgroup, Z read once, keys resident in LDS: csrc/stats.hip rs_small); 1 for
 * M <= 1,048,576 (<= 64 workgroups, keys in registers, global histograms that the workgroups wait on: rs_coop -- these
 * workgroups must be resident together, which an otherwise idle or normally loaded GPU guarantees; a wait that does
 * not complete within ~0.5 s, e.g. on a workspace some aborted call left unclean, ends in NaN statistics and an
 * un-tagged workspace, never in a hang); 5 beyond (3 histogram passes, tail, final).  Exact selection, deterministic
 * sums; the three forms agree exactly on every output except the fp64 sums (equal to summation order).
 */
#define RATO_N_STATS 11
int rato_risk_stats(const float* Z, int64_t M, double alpha, float thr,
                    void* workspace, size_t workspace_bytes, double* out, void* stream);

/* Recovery path: the same statistics by the launch-per-pass form (which waits for nothing) on a workspace re-initialised
 * on the stream.  For a caller whose rato_risk_stats record came back NaN although Z is finite -- the one-launch forms
 * give up, loudly, when the workgroups of their launch could not run together for seconds or the workspace was left
 * unclean by an aborted call.  The Python facades do this by themselves (stats.risk_stats, CvarCutSolver.evaluate). */
int rato_risk_stats_recover(const float* Z, int64_t M, double alpha, float thr,
                            void* workspace, size_t workspace_bytes, double* out, void* stream);

/* rato_sum_partials(part, nblocks, ncols, scale, sums_out) and rato_risk_stats(Z, ...) in ONE launch when
 * M <= 1,048,576 (the partial-sum workgroups ride along with the selection workgroups; two stream-ordered calls
 * otherwise): the whole reduction stage of a single-GPU SAA step. */
int rato_sums_and_risk_stats(const float* part, int32_t nblocks, int32_t ncols, double scale, double* sums_out,
                             const float* Z, int64_t M, double alpha, float thr, void* workspace,
                             size_t workspace_bytes, double* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RATO_SAA_H */
