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
#include <math.h>
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
