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

void rato_oracle_drone(int M, int S, double dt, const double* us, const double* DWs, const double* masses,
                       const double* obs_Qs, double* xs_out, double* v_final_du, double* val_final,
                       double* g_obs_du, double* g_up, double* Z, int nthreads) {
  const int nU = NU * S;
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel
  {
    /* per-thread scratch: state trajectory and the forward sensitivities Phi[t][a][s][2] */
    double* xs = (double*)malloc(sizeof(double) * (size_t)(S + 1) * NX);
    double* Phi = (double*)malloc(sizeof(double) * (size_t)(S + 1) * 3 * S * 2);
#pragma omp for schedule(static)
    for (int i = 0; i < M; ++i) {
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
      if (xs_out) memcpy(xs_out + (size_t)i * (S + 1) * NX, xs, sizeof(double) * (size_t)(S + 1) * NX);
      /* final constraints */
      const double* PhS = Phi + (size_t)S * 3 * S * 2;
      if (v_final_du) {
        double* out = v_final_du + (size_t)i * NX * nU;
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
          val_final[(size_t)i * NX + a] = -xs[(size_t)S * NX + a] + dp;          /* x_final = 0 */
          val_final[(size_t)i * NX + 3 + a] = -xs[(size_t)S * NX + 3 + a] + dv;
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
          double* row = g_obs_du ? g_obs_du + (((size_t)i * NOBS + j) * S + t) * nU : NULL;
          if (row) memset(row, 0, sizeof(double) * nU);
          for (int s = 0; s < t; ++s) {
            const double ex = wx * Pt[((size_t)0 * S + s) * 2], ey = wy * Pt[((size_t)1 * S + s) * 2];
            if (row) {
              row[s * NU + 0] = ex;
              row[s * NU + 1] = ey;
            }
            dot += ex * us[s * NU + 0] + ey * us[s * NU + 1];
          }
          if (g_up) g_up[((size_t)i * NOBS + j) * S + t] = -g + dot;
        }
      }
      if (Z) Z[i] = zmax - OSQP_TOL;
    }
    free(xs);
    free(Phi);
  }
}
