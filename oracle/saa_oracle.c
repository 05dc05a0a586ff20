/*
 * Oracle (TEST INFRASTRUCTURE, see oracle/__init__.py) — plain C fp64 restatement of the
 * drone SAA inner loop of /root/reference/drone/drone_risk.py, one sample per loop iteration,
 * OpenMP over the sample axis.  Used (i) to cross-check the NumPy oracle and (ii) as the
 * multi-core CPU baseline of bench.py ("port": the reference's own JAX/XLA-CPU path cannot be
 * installed here).  Checked against the NumPy oracle (tests/test_oracle_c.py), which is pinned by executing the
 * reference's own text (oracle/__init__.py).
 *
 * Outputs use the reference's dense shapes (C order):
 *   xs          (M, S+1, 6)     us_to_state_trajectories        drone_risk.py:139-162
 *   v_final_du  (M, 6, 3S)      d final_constraints / d u        :239-268
 *   val_final   (M, 6)          -v_final + v_final_du . u        :271
 *   g_obs_du    (M, 3, S, 3S)   d obstacle constraints / d u     :169-213, :255-268
 *   g_up        (M, 3, S)       -g + g_obs_du . u                :278
 *   Z           (M)             max g - OSQP_TOL                 :656-662
 * Any output pointer may be NULL.
 */
#define _GNU_SOURCE
#include <math.h>
#include <sched.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define NX 6
#define NU 3
#define NOBS 3

static const double KP = 0.05, KD = 0.25;           /* -feedback_gain, drone_params.py:14-19 */
static const double BETA = 1e-2, DRAG = 0.2;        /* :24-25 */
static const double OSQP_TOL = 1e-3;                /* :4 */
static const double OBS[NOBS][2] = {{-1.4, -0.1}, {-0.7, 0.3}, {-0.3, 0.25}};   /* :34-37 */
static const double X_INIT[NX] = {-1.9, 0.05, 0.2, 0.0, 0.0, 0.0};              /* :45 */

int rato_oracle_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* The CPU baseline's thread placement (bench.py): OMP_PROC_BIND=spread by hand, because the environment variable is read
 * when libgomp is loaded (long before bench.py knows the box) and would also pin the caller's own thread, which feeds
 * the GPU.  spread != 0: worker t (t > 0) of an nthreads-wide team is pinned to the (t * n_allowed / nthreads)-th cpu of
 * the process's affinity mask; spread == 0: every worker gets the whole mask back.  The calling thread is never touched.
 * libgomp keeps its workers between parallel regions, so the placement holds for the calls that follow with the same
 * nthreads.  -> number of cpus in the mask. */
int rato_oracle_place_threads(int nthreads, int spread) {
  cpu_set_t allowed;
  CPU_ZERO(&allowed);
  if (sched_getaffinity(0, sizeof(allowed), &allowed) != 0) return -1;
  int cpus[CPU_SETSIZE], n = 0;
  for (int c = 0; c < CPU_SETSIZE; ++c)
    if (CPU_ISSET(c, &allowed)) cpus[n++] = c;
  if (n == 0) return -1;
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#pragma omp parallel
  {
    const int t = omp_get_thread_num(), nt = omp_get_num_threads();
    if (t > 0) {
      cpu_set_t one;
      if (spread) {
        CPU_ZERO(&one);
        CPU_SET(cpus[(int)(((long)t * n) / nt) % n], &one);
      } else {
        one = allowed;
      }
      sched_setaffinity(0, sizeof(one), &one);
    }
  }
#endif
  return n;
}

/* One sample i.  ``slot`` is the index its dense outputs are written at: i itself (rato_oracle_drone: outputs of all M
 * samples kept, the reference's vmap) or a per-thread slot that is reused (rato_oracle_drone_stream). */
static void drone_sample(int i, size_t slot, int S, double dt, const double* us, const double* DWs,
                         const double* masses, const double* obs_Qs, double* xs, double* Phi, double* xs_out,
                         double* v_final_du, double* val_final, double* g_obs_du, double* g_up, double* Z) {
  const int nU = NU * S;
  {
    {
      const double m = masses[i];
      const double* dW = DWs + (size_t)i * S * NX;
      const double* Q = obs_Qs + (size_t)i * NOBS * 9;
      memcpy(xs, X_INIT, sizeof(X_INIT));
      memset(Phi, 0, sizeof(double) * (size_t)(S + 1) * 3 * S * 2);
      for (int t = 0; t < S; ++t) {
        const double* x = xs + (size_t)t * NX;
        double* xn = xs + (size_t)(t + 1) * NX;
        for (int a = 0; a < 3; ++a) {
          const double p = x[a], v = x[3 + a];
          const double u = us[t * NU + a];
          /* b (:122-131), sigma (:133-137), Euler-Maruyama with sqrt(dt) applied again (:151) */
          const double acc = (u - (KP * p + KD * v)) / m - DRAG * fabs(v) * v / m;
          xn[a] = p + dt * v;
          xn[3 + a] = v + dt * acc + sqrt(dt) * (BETA / m) * dW[t * NX + 3 + a];
          /* sensitivities: A_t = [[1, dt], [-KP dt/m, 1 - dt (KD + 2 DRAG |v|)/m]], B = [0, dt/m] */
          const double a21 = -KP * dt / m, a22 = 1.0 - dt * (KD + 2.0 * DRAG * fabs(v)) / m;
          const double* Ph = Phi + ((size_t)t * 3 + a) * S * 2;
          double* Pn = Phi + ((size_t)(t + 1) * 3 + a) * S * 2;
          for (int s = 0; s < t; ++s) {
            Pn[2 * s] = Ph[2 * s] + dt * Ph[2 * s + 1];
            Pn[2 * s + 1] = a21 * Ph[2 * s] + a22 * Ph[2 * s + 1];
          }
          Pn[2 * t] = 0.0;
          Pn[2 * t + 1] = dt / m;
        }
      }
      if (xs_out) memcpy(xs_out + slot * (S + 1) * NX, xs, sizeof(double) * (size_t)(S + 1) * NX);
      /* final constraints */
      const double* PhS = Phi + (size_t)S * 3 * S * 2;
      if (v_final_du) {
        double* out = v_final_du + slot * NX * nU;
        memset(out, 0, sizeof(double) * NX * nU);
        for (int a = 0; a < 3; ++a)
          for (int s = 0; s < S; ++s) {
            out[(size_t)a * nU + s * NU + a] = PhS[((size_t)a * S + s) * 2];
            out[(size_t)(3 + a) * nU + s * NU + a] = PhS[((size_t)a * S + s) * 2 + 1];
          }
      }
      if (val_final) {
        for (int a = 0; a < 3; ++a) {
          double dp = 0.0, dv = 0.0;
          for (int s = 0; s < S; ++s) {
            dp += PhS[((size_t)a * S + s) * 2] * us[s * NU + a];
            dv += PhS[((size_t)a * S + s) * 2 + 1] * us[s * NU + a];
          }
          val_final[slot * NX + a] = -xs[(size_t)S * NX + a] + dp;          /* x_final = 0 */
          val_final[slot * NX + 3 + a] = -xs[(size_t)S * NX + 3 + a] + dv;
        }
      }
      /* obstacle constraints g = 1 - d^T Q[:2,:2] d, gradient -(Q+Q^T) d */
      double zmax = -INFINITY;
      for (int j = 0; j < NOBS; ++j) {
        const double q00 = Q[j * 9 + 0], q01 = Q[j * 9 + 1], q10 = Q[j * 9 + 3], q11 = Q[j * 9 + 4];
        for (int t = 0; t < S; ++t) {
          const double dx = xs[(size_t)(t + 1) * NX + 0] - OBS[j][0], dy = xs[(size_t)(t + 1) * NX + 1] - OBS[j][1];
          const double g = 1.0 - (dx * (q00 * dx + q01 * dy) + dy * (q10 * dx + q11 * dy));
          const double wx = -((q00 + q00) * dx + (q01 + q10) * dy), wy = -((q10 + q01) * dx + (q11 + q11) * dy);
          if (g > zmax) zmax = g;
          const double* Pt = Phi + (size_t)(t + 1) * 3 * S * 2;
          double dot = 0.0;
          double* row = g_obs_du ? g_obs_du + ((slot * NOBS + j) * S + t) * nU : NULL;
          if (row) memset(row, 0, sizeof(double) * nU);
          for (int s = 0; s < t; ++s) {
            const double ex = wx * Pt[((size_t)0 * S + s) * 2], ey = wy * Pt[((size_t)1 * S + s) * 2];
            if (row) {
              row[s * NU + 0] = ex;
              row[s * NU + 1] = ey;
            }
            dot += ex * us[s * NU + 0] + ey * us[s * NU + 1];
          }
          if (g_up) g_up[(slot * NOBS + j) * S + t] = -g + dot;
        }
      }
      if (Z) Z[i] = zmax - OSQP_TOL;
    }
  }
}

void rato_oracle_drone(int M, int S, double dt, const double* us, const double* DWs, const double* masses,
                       const double* obs_Qs, double* xs_out, double* v_final_du, double* val_final,
                       double* g_obs_du, double* g_up, double* Z, int nthreads) {
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel
  {
    /* per-thread scratch: state trajectory and the forward sensitivities Phi[t][a][s][2] */
    double* xs = (double*)malloc(sizeof(double) * (size_t)(S + 1) * NX);
    double* Phi = (double*)malloc(sizeof(double) * (size_t)(S + 1) * 3 * S * 2);
#pragma omp for schedule(static)
    for (int i = 0; i < M; ++i)
      drone_sample(i, (size_t)i, S, dt, us, DWs, masses, obs_Qs, xs, Phi, xs_out, v_final_du, val_final, g_obs_du, g_up,
                   Z);
    free(xs);
    free(Phi);
  }
}

/* Streaming form for batches whose dense outputs do not fit the host (M = 1e5, S = 50: the reference's (M,3,S,3S)
 * Jacobian alone is 18 GB): every sample's dense linearization is still FORMED, in the reference's shapes, but into a
 * per-thread buffer that the next sample overwrites; what leaves the loop is what the SCP consumes downstream of it --
 *   sum_final_du (6, 3S), sum_val_final (6)   sums over the samples (the sample mean, drone_risk.py:294-296)
 *   Z (M)                                      max constraint value per sample (:656-662)
 *   checksum[0]                                sum of every g_obs_du and g_up entry (keeps the dense rows observable)
 * bench.py times this as the multi-core CPU baseline at the metric's own M. */
void rato_oracle_drone_stream(int M, int S, double dt, const double* us, const double* DWs, const double* masses,
                              const double* obs_Qs, double* sum_final_du, double* sum_val_final, double* Z,
                              double* checksum, int nthreads) {
  const int nU = NU * S;
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
  memset(sum_final_du, 0, sizeof(double) * NX * nU);
  memset(sum_val_final, 0, sizeof(double) * NX);
  double total = 0.0;
#pragma omp parallel reduction(+ : total)
  {
    double* xs = (double*)malloc(sizeof(double) * (size_t)(S + 1) * NX);
    double* Phi = (double*)malloc(sizeof(double) * (size_t)(S + 1) * 3 * S * 2);
    double* fdu = (double*)malloc(sizeof(double) * NX * nU);
    double* vf = (double*)malloc(sizeof(double) * NX);
    double* gdu = (double*)malloc(sizeof(double) * (size_t)NOBS * S * nU);
    double* gup = (double*)malloc(sizeof(double) * NOBS * S);
    double* acc_du = (double*)calloc((size_t)NX * nU, sizeof(double));
    double acc_vf[NX] = {0, 0, 0, 0, 0, 0};
#pragma omp for schedule(static)
    for (int i = 0; i < M; ++i) {
      drone_sample(i, 0, S, dt, us, DWs, masses, obs_Qs, xs, Phi, NULL, fdu, vf, gdu, gup, Z);
      for (int k = 0; k < NX * nU; ++k) acc_du[k] += fdu[k];
      for (int k = 0; k < NX; ++k) acc_vf[k] += vf[k];
      double c = 0.0;
      for (size_t k = 0; k < (size_t)NOBS * S * nU; ++k) c += gdu[k];
      for (int k = 0; k < NOBS * S; ++k) c += gup[k];
      total += c;
    }
#pragma omp critical
    {
      for (int k = 0; k < NX * nU; ++k) sum_final_du[k] += acc_du[k];
      for (int k = 0; k < NX; ++k) sum_val_final[k] += acc_vf[k];
    }
    free(xs); free(Phi); free(fdu); free(vf); free(gdu); free(gup); free(acc_du);
  }
  if (checksum) checksum[0] = total;
}

/* =====================================================================================================================
 * Streaming fp64 CUT ORACLE (round 4).  The reduced SCP subproblem (tests/_host_cuts.py) needs, per candidate u,
 *     m_i(u) = max_r [ (G_i u)_r - g_up_{i,r} ]        the rows of drone_risk.py:357-364 / driving.py:358-363 without
 *                                                       kappa, y_i, t:  g_obs_du . u - g_up,  g_up = -g + g_obs_du . u_k
 * its arg-max row, and for a tail weighting w the sums  sum_i w_i G_i[r_i, :],  sum_i w_i g_up_i[r_i].
 * The dense fp64 rows of M = 1e5 samples do not fit a host (drone S = 50: 18 GB), so nothing is stored per sample: each
 * thread forms ONE sample's linearization in the reference's dense shapes (drone_sample / car_sample), consumes it and
 * overwrites it with the next sample's.  Same arithmetic as the dense oracle (tests/test_oracle_c.py: equal to 1e-12).
 * ===================================================================================================================== */

void rato_oracle_drone_rowmax(int M, int S, double dt, const double* us_k, const double* u, const double* DWs,
                              const double* masses, const double* obs_Qs, double* m_out, int* arg_out, int nthreads) {
  const int nU = NU * S, R = NOBS * S;
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel
  {
    double* xs = (double*)malloc(sizeof(double) * (size_t)(S + 1) * NX);
    double* Phi = (double*)malloc(sizeof(double) * (size_t)(S + 1) * 3 * S * 2);
    double* gdu = (double*)malloc(sizeof(double) * (size_t)R * nU);
    double* gup = (double*)malloc(sizeof(double) * R);
#pragma omp for schedule(static)
    for (int i = 0; i < M; ++i) {
      drone_sample(i, 0, S, dt, us_k, DWs, masses, obs_Qs, xs, Phi, NULL, NULL, NULL, gdu, gup, NULL);
      double best = -INFINITY;
      int arg = 0;
      for (int r = 0; r < R; ++r) {
        const double* row = gdu + (size_t)r * nU;
        const int nz = NU * (r % S);                   /* row (j, t): d g_t / d u_s = 0 for s >= t (exact zeros) */
        double dot = 0.0;
        for (int c = 0; c < nz; ++c) dot += row[c] * u[c];
        const double v = dot - gup[r];
        if (v > best) { best = v; arg = r; }          /* first maximum, as numpy.argmax */
      }
      m_out[i] = best;
      arg_out[i] = arg;
    }
    free(xs); free(Phi); free(gdu); free(gup);
  }
}

/* K tail weightings at once (w, arg: (K, M) C order): grad_out (K, nU) = sum_i w_ki G_i[arg_ki, :],
 * gup_out (K) = sum_i w_ki g_up_i[arg_ki].  Per-thread partial sums are added in thread order: the result does not
 * depend on the schedule of a run. */
void rato_oracle_drone_tail_rows(int M, int S, double dt, const double* us_k, const double* DWs, const double* masses,
                                 const double* obs_Qs, int K, const double* w, const int* arg, double* grad_out,
                                 double* gup_out, int nthreads) {
  const int nU = NU * S, R = NOBS * S, W = nU + 1;
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
  const int nt_max = omp_get_max_threads();
#else
  const int nt_max = 1;
#endif
  double* acc_all = (double*)calloc((size_t)nt_max * K * W, sizeof(double));
#pragma omp parallel
  {
#ifdef _OPENMP
    double* acc = acc_all + (size_t)omp_get_thread_num() * K * W;
#else
    double* acc = acc_all;
#endif
    double* xs = (double*)malloc(sizeof(double) * (size_t)(S + 1) * NX);
    double* Phi = (double*)malloc(sizeof(double) * (size_t)(S + 1) * 3 * S * 2);
    double* gdu = (double*)malloc(sizeof(double) * (size_t)R * nU);
    double* gup = (double*)malloc(sizeof(double) * R);
#pragma omp for schedule(static)
    for (int i = 0; i < M; ++i) {
      int any = 0;
      for (int k = 0; k < K; ++k) any |= (w[(size_t)k * M + i] != 0.0);
      if (!any) continue;
      drone_sample(i, 0, S, dt, us_k, DWs, masses, obs_Qs, xs, Phi, NULL, NULL, NULL, gdu, gup, NULL);
      for (int k = 0; k < K; ++k) {
        const double wk = w[(size_t)k * M + i];
        if (wk == 0.0) continue;
        const int r = arg[(size_t)k * M + i];
        const double* row = gdu + (size_t)r * nU;
        double* a = acc + (size_t)k * W;
        for (int c = 0; c < nU; ++c) a[c] += wk * row[c];
        a[nU] += wk * gup[r];
      }
    }
    free(xs); free(Phi); free(gdu); free(gup);
  }
  for (int k = 0; k < K; ++k) {
    for (int c = 0; c < nU; ++c) grad_out[(size_t)k * nU + c] = 0.0;
    gup_out[k] = 0.0;
  }
  for (int t = 0; t < nt_max; ++t)
    for (int k = 0; k < K; ++k) {
      const double* a = acc_all + ((size_t)t * K + k) * W;
      for (int c = 0; c < nU; ++c) grad_out[(size_t)k * nU + c] += a[c];
      gup_out[k] += a[nU];
    }
  free(acc_all);
}

/* ---------------------------------------------------------------------------------------------------------------------
 * Driving (car + pedestrian), /root/reference/car/driving.py, one sample: rollout (:186-214), forward sensitivities
 * X_{t+1} = J_t X_t + B_t of the Euler-Maruyama step (what jacfwd of :267-276 computes), separation rows (:232-236,
 * :269), g_up (:295).  Constants: driving_params.py:13-40.  Outputs in the reference's dense shapes:
 *   g_obs_du (S, 2S),  g_up (S),  optionally xs (S+1, 8).
 * ------------------------------------------------------------------------------------------------------------------- */
#define CNX 8
#define CNU 2
static const double C_T = 10.0, C_BETA = 3e-2, C_SPEED_DES = 1.3;
static const double C_EGO_W = 2.695, C_EGO_H = 1.663, C_PED_R = 0.5;

static void car_sample(int i, int S, const double* us, const double* states_init, const double* omegas_speed,
                       const double* omegas_rep, const double* DWs, double* xs, double* X, double* Xn, double* g_obs_du,
                       double* g_up, double* zmax_out) {
  const int nU = CNU * S;
  const double dt = C_T / S, sq = sqrt(dt);
  const double min_sep = C_PED_R + sqrt(C_EGO_W * C_EGO_W + C_EGO_H * C_EGO_H);
  const double ws = omegas_speed[i], wr = omegas_rep[i];
  const double* dW = DWs + (size_t)i * S * CNX;
  memcpy(xs, states_init + (size_t)i * CNX, sizeof(double) * CNX);
  memset(X, 0, sizeof(double) * CNX * nU);
  double zmax = -INFINITY;
  for (int t = 0; t < S; ++t) {
    const double* x = xs + (size_t)t * CNX;
    double* xn = xs + (size_t)(t + 1) * CNX;
    /* force on the pedestrian (:145-158): -w_r delta/|delta| + w_s (v_des - v_y) added to BOTH components */
    const double dx = x[0] - x[4], dy = x[1] - x[5];
    const double r = sqrt(dx * dx + dy * dy), nx = dx / r, ny = dy / r;
    const double fs = ws * (C_SPEED_DES - x[7]);
    const double c = cos(x[3]), s = sin(x[3]);
    xn[0] = x[0] + dt * x[2] * c;
    xn[1] = x[1] + dt * x[2] * s;
    xn[2] = x[2] + dt * us[t * CNU + 0];
    xn[3] = x[3] + dt * us[t * CNU + 1];
    xn[4] = x[4] + dt * x[6];
    xn[5] = x[5] + dt * x[7];
    xn[6] = x[6] + dt * (-wr * nx + fs) + sq * C_BETA * dW[t * CNX + 6];      /* sqrt(dt) again, :200 */
    xn[7] = x[7] + dt * (-wr * ny + fs) + sq * C_BETA * dW[t * CNX + 7];
    /* sensitivities: only columns < 2t are nonzero in X_t */
    const double h00 = (1.0 - nx * nx) / r, h01 = -nx * ny / r, h11 = (1.0 - ny * ny) / r;
    const int nc = 2 * t;
    for (int col = 0; col < nc; ++col) {
      const double x0 = X[0 * nU + col], x1 = X[1 * nU + col], x2 = X[2 * nU + col], x3 = X[3 * nU + col];
      const double x4 = X[4 * nU + col], x5 = X[5 * nU + col], x6 = X[6 * nU + col], x7 = X[7 * nU + col];
      const double d0 = x0 - x4, d1 = x1 - x5;
      Xn[0 * nU + col] = x0 + dt * c * x2 - dt * x[2] * s * x3;
      Xn[1 * nU + col] = x1 + dt * s * x2 + dt * x[2] * c * x3;
      Xn[2 * nU + col] = x2;
      Xn[3 * nU + col] = x3;
      Xn[4 * nU + col] = x4 + dt * x6;
      Xn[5 * nU + col] = x5 + dt * x7;
      Xn[6 * nU + col] = x6 - dt * wr * (h00 * d0 + h01 * d1) - dt * ws * x7;
      Xn[7 * nU + col] = x7 - dt * wr * (h01 * d0 + h11 * d1) - dt * ws * x7;
    }
    for (int k = 0; k < CNX; ++k) {
      Xn[k * nU + nc] = 0.0;
      Xn[k * nU + nc + 1] = 0.0;
    }
    Xn[2 * nU + nc] = dt;
    Xn[3 * nU + nc + 1] = dt;
    for (int k = 0; k < CNX; ++k) memcpy(X + (size_t)k * nU, Xn + (size_t)k * nU, sizeof(double) * (nc + 2));
    /* row t: g = -(|delta_{t+1}| - min_sep), gradient -n . (X[0:2] - X[4:6]) */
    const double ex = xn[0] - xn[4], ey = xn[1] - xn[5];
    const double rr = sqrt(ex * ex + ey * ey), mx = ex / rr, my = ey / rr;
    const double g = -(rr - min_sep);
    if (g > zmax) zmax = g;
    double dot = 0.0;
    double* row = g_obs_du + (size_t)t * nU;
    for (int col = 0; col < nU; ++col) {
      double v = 0.0;
      if (col < nc + 2) v = -(mx * (X[0 * nU + col] - X[4 * nU + col]) + my * (X[1 * nU + col] - X[5 * nU + col]));
      row[col] = v;
      dot += v * us[col];
    }
    g_up[t] = -g + dot;
  }
  if (zmax_out) *zmax_out = zmax;
}

/* dense form (small M; checked against the NumPy oracle): g_obs_du (M,S,2S), g_up (M,S), xs (M,S+1,8), Z (M) */
void rato_oracle_car(int M, int S, const double* us, const double* states_init, const double* omegas_speed,
                     const double* omegas_rep, const double* DWs, double* xs_out, double* g_obs_du, double* g_up,
                     double* Z, int nthreads) {
  const int nU = CNU * S;
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel
  {
    double* xs = (double*)malloc(sizeof(double) * (size_t)(S + 1) * CNX);
    double* X = (double*)malloc(sizeof(double) * CNX * nU);
    double* Xn = (double*)malloc(sizeof(double) * CNX * nU);
#pragma omp for schedule(static)
    for (int i = 0; i < M; ++i) {
      double z;
      car_sample(i, S, us, states_init, omegas_speed, omegas_rep, DWs, xs, X, Xn, g_obs_du + (size_t)i * S * nU,
                 g_up + (size_t)i * S, &z);
      if (xs_out) memcpy(xs_out + (size_t)i * (S + 1) * CNX, xs, sizeof(double) * (size_t)(S + 1) * CNX);
      if (Z) Z[i] = z - 3e-4;                                      /* OSQP_TOL, driving_params.py:4; driving.py:630-638 */
    }
    free(xs); free(X); free(Xn);
  }
}

void rato_oracle_car_rowmax(int M, int S, const double* us_k, const double* u, const double* states_init,
                            const double* omegas_speed, const double* omegas_rep, const double* DWs, double* m_out,
                            int* arg_out, int nthreads) {
  const int nU = CNU * S;
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel
  {
    double* xs = (double*)malloc(sizeof(double) * (size_t)(S + 1) * CNX);
    double* X = (double*)malloc(sizeof(double) * CNX * nU);
    double* Xn = (double*)malloc(sizeof(double) * CNX * nU);
    double* gdu = (double*)malloc(sizeof(double) * (size_t)S * nU);
    double* gup = (double*)malloc(sizeof(double) * S);
#pragma omp for schedule(static)
    for (int i = 0; i < M; ++i) {
      car_sample(i, S, us_k, states_init, omegas_speed, omegas_rep, DWs, xs, X, Xn, gdu, gup, NULL);
      double best = -INFINITY;
      int arg = 0;
      for (int r = 0; r < S; ++r) {
        const double* row = gdu + (size_t)r * nU;
        double dot = 0.0;
        for (int c = 0; c < nU; ++c) dot += row[c] * u[c];
        const double v = dot - gup[r];
        if (v > best) { best = v; arg = r; }
      }
      m_out[i] = best;
      arg_out[i] = arg;
    }
    free(xs); free(X); free(Xn); free(gdu); free(gup);
  }
}

void rato_oracle_car_tail_rows(int M, int S, const double* us_k, const double* states_init, const double* omegas_speed,
                               const double* omegas_rep, const double* DWs, int K, const double* w, const int* arg,
                               double* grad_out, double* gup_out, int nthreads) {
  const int nU = CNU * S, W = nU + 1;
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
  const int nt_max = omp_get_max_threads();
#else
  const int nt_max = 1;
#endif
  double* acc_all = (double*)calloc((size_t)nt_max * K * W, sizeof(double));
#pragma omp parallel
  {
#ifdef _OPENMP
    double* acc = acc_all + (size_t)omp_get_thread_num() * K * W;
#else
    double* acc = acc_all;
#endif
    double* xs = (double*)malloc(sizeof(double) * (size_t)(S + 1) * CNX);
    double* X = (double*)malloc(sizeof(double) * CNX * nU);
    double* Xn = (double*)malloc(sizeof(double) * CNX * nU);
    double* gdu = (double*)malloc(sizeof(double) * (size_t)S * nU);
    double* gup = (double*)malloc(sizeof(double) * S);
#pragma omp for schedule(static)
    for (int i = 0; i < M; ++i) {
      int any = 0;
      for (int k = 0; k < K; ++k) any |= (w[(size_t)k * M + i] != 0.0);
      if (!any) continue;
      car_sample(i, S, us_k, states_init, omegas_speed, omegas_rep, DWs, xs, X, Xn, gdu, gup, NULL);
      for (int k = 0; k < K; ++k) {
        const double wk = w[(size_t)k * M + i];
        if (wk == 0.0) continue;
        const int r = arg[(size_t)k * M + i];
        const double* row = gdu + (size_t)r * nU;
        double* a = acc + (size_t)k * W;
        for (int c = 0; c < nU; ++c) a[c] += wk * row[c];
        a[nU] += wk * gup[r];
      }
    }
    free(xs); free(X); free(Xn); free(gdu); free(gup);
  }
  for (int k = 0; k < K; ++k) {
    for (int c = 0; c < nU; ++c) grad_out[(size_t)k * nU + c] = 0.0;
    gup_out[k] = 0.0;
  }
  for (int t = 0; t < nt_max; ++t)
    for (int k = 0; k < K; ++k) {
      const double* a = acc_all + ((size_t)t * K + k) * W;
      for (int c = 0; c < nU; ++c) grad_out[(size_t)k * nU + c] += a[c];
      gup_out[k] += a[nU];
    }
  free(acc_all);
}
