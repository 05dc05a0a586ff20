// Host side of the boundary, continued (see nnls.hip): the master QP of the cutting-plane loop as an object that lives
// across the cuts of one SCP subproblem -- HOST code only.
//
//     min 1/2 z'Pz + q'z   s.t.  A_eq z = b_eq,   rows_j . z <= rhs_j  (appended one at a time)
//
// with P DIAGONAL and positive (the SCP master: 2 dt R blocks and the slack penalty) -- the case
// riskaversetrajopt_amd/dense_qp.py::Master special-cases; this is the same algorithm:
//   * whiten first, z = D^-1/2 zt (D = P): the null-space basis of the scaled equalities A_eq D^-1/2 is then orthonormal
//     in the whitened metric and the reduced Hessian is the identity.  The basis is never formed: it is the last n - m
//     columns of the Householder Q of (A_eq D^-1/2)', kept as m reflectors, and every product with it costs O(n m)
//     (NumPy's complete QR + the dense N cost 0.27 ms per SCP iteration, and a row of the master a 151 x 145 product);
//   * the problem in v (z = x0 + N (v - c)) is a least-distance problem  min |v|  s.t.  G v >= h, solved through NNLS
//     (Lawson & Hanson ch. 23) on [G' ; h'/sigma] with the passive set of the previous solve as the start AND its thin QR
//     factor kept (rato_nnls.h; round 6); sigma ~ |v| (within a factor 4) keeps the KKT tolerance of the NNLS meaningful
//     for optima far out.
// The reference hands the whole subproblem to OSQP (drone_risk.py:433-457).
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#include <vector>

#include "rato_nnls.h"
#include "rato_saa.h"

struct rato_master {
  int n = 0, m = 0, nk = 0;          // variables, equalities, null-space dimension
  std::vector<double> d;             // 1 / sqrt(P_ii)
  std::vector<double> V;             // m reflectors, v_j at V[j*n .. j*n+n) (zeros above j, v_j[j] = 1)
  std::vector<double> beta;          // H_j = I - beta_j v_j v_j'
  std::vector<double> x0;            // a particular solution of A_eq x = b_eq
  std::vector<double> c;             // N'(P x0 + q)
  std::vector<double> rows;          // master rows in the v variables: [G_j (nk) | h_j], unit |G_j|
  std::vector<double> sc;            // the scale each row was divided by
  std::vector<uint8_t> passive;      // passive set of the last solve (warm start of the next one)
  double vnorm = 0.0;
  int nrows = 0;
  std::vector<double> work, An, y;
  // Round 6: the NNLS problem [G' ; h'/sigma] y ~ [0; 1] GROWS by the columns of the rows added between two solves, and the
  // thin QR factor of the previous solve's passive columns stays valid as long as sigma does -- it is kept (rebuilding it
  // column by column was ~2/3 of a solve).  sigma is re-chosen only when |v| has left [sigma/4, 4 sigma]; the factor is
  // rebuilt then, after FACTOR_REFRESH updates / downdates (rounding of the Givens downdates), and after a failed solve.
  rato_nnls::ThinQR qr;
  bool factor_ready = false;
  int an_cols = 0, changes = 0;       // columns of An in place; updates + downdates since the factor was last built
  double sigma = 0.0, a1 = 0.0;      // the scale An was built with; max column 1-norm of An
  std::vector<double> nn_s, nn_resid, bn;
  std::vector<uint8_t> nn_banned;

  void apply_Qt(double* t) const {   // t <- Q' t = H_m ... H_1 t
    for (int j = 0; j < m; ++j) {
      const double* v = &V[(size_t)j * n];
      double s = 0.0;
      for (int i = j; i < n; ++i) s += v[i] * t[i];
      s *= beta[j];
      for (int i = j; i < n; ++i) t[i] -= s * v[i];
    }
  }
  void apply_Q(double* t) const {    // t <- Q t = H_1 ... H_m t
    for (int j = m - 1; j >= 0; --j) {
      const double* v = &V[(size_t)j * n];
      double s = 0.0;
      for (int i = j; i < n; ++i) s += v[i] * t[i];
      s *= beta[j];
      for (int i = j; i < n; ++i) t[i] -= s * v[i];
    }
  }
};

extern "C" int rato_master_create(rato_master** out, int32_t n, const double* p_diag, const double* q, int32_t m_eq,
                                  const double* A_eq, const double* b_eq) {
  if (!out || n < 1 || !p_diag || !q || m_eq < 0 || m_eq >= n || (m_eq > 0 && (!A_eq || !b_eq))) return RATO_EINVAL;
  for (int i = 0; i < n; ++i)
    if (!(p_diag[i] > 0.0)) return RATO_EINVAL;
  rato_master* M = new rato_master;
  M->n = n;
  M->m = m_eq;
  M->nk = n - m_eq;
  M->d.resize(n);
  for (int i = 0; i < n; ++i) M->d[i] = 1.0 / sqrt(p_diag[i]);
  // Householder QR of B = (A_eq D^-1/2)' (n x m), column by column
  const int m = m_eq;
  std::vector<double> B((size_t)n * m), R((size_t)m * m, 0.0);
  for (int j = 0; j < m; ++j)
    for (int i = 0; i < n; ++i) B[(size_t)j * n + i] = A_eq[(size_t)j * n + i] * M->d[i];
  M->V.assign((size_t)m * n, 0.0);
  M->beta.assign(m, 0.0);
  double rmax = 0.0, rmin = INFINITY;
  for (int j = 0; j < m; ++j) {
    double* x = &B[(size_t)j * n];
    for (int k = 0; k < j; ++k) {   // the earlier reflectors on this column
      const double* v = &M->V[(size_t)k * n];
      double s = 0.0;
      for (int i = k; i < n; ++i) s += v[i] * x[i];
      s *= M->beta[k];
      for (int i = k; i < n; ++i) x[i] -= s * v[i];
    }
    double nrm = 0.0;
    for (int i = j; i < n; ++i) nrm += x[i] * x[i];
    nrm = sqrt(nrm);
    const double alpha = (x[j] > 0.0) ? -nrm : nrm;
    double* v = &M->V[(size_t)j * n];
    const double v0 = x[j] - alpha;
    if (nrm == 0.0 || v0 == 0.0) {
      delete M;
      return RATO_EINVAL;
    }
    double vv = 1.0;
    v[j] = 1.0;
    for (int i = j + 1; i < n; ++i) {
      v[i] = x[i] / v0;
      vv += v[i] * v[i];
    }
    M->beta[j] = 2.0 / vv;
    for (int k = 0; k < j; ++k) R[(size_t)j * m + k] = x[k];   // column j of R, rows 0..j
    R[(size_t)j * m + j] = alpha;
    rmax = fmax(rmax, fabs(alpha));
    rmin = fmin(rmin, fabs(alpha));
  }
  if (m > 0 && !(rmin > 1e-10 * rmax)) {   // rank-deficient equalities: the caller falls back to the SVD path
    delete M;
    return RATO_EINVAL;
  }
  // x0 = D^-1/2 y,  y = the minimum-norm solution of (A_eq D^-1/2) y = b:  R'w = b,  y = Q [w; 0]
  std::vector<double> w(n, 0.0);
  for (int i = 0; i < m; ++i) {
    double acc = b_eq[i];
    for (int k = 0; k < i; ++k) acc -= R[(size_t)i * m + k] * w[k];   // R'[i][k] = R[k][i] = column i, row k
    w[i] = acc / R[(size_t)i * m + i];
  }
  M->apply_Q(w.data());
  M->x0.resize(n);
  for (int i = 0; i < n; ++i) M->x0[i] = M->d[i] * w[i];
  // c = N'(P x0 + q) = (Q'[D^-1/2 (P x0 + q)])[m:]
  std::vector<double> t(n);
  for (int i = 0; i < n; ++i) t[i] = M->d[i] * (p_diag[i] * M->x0[i] + q[i]);
  M->apply_Qt(t.data());
  M->c.assign(t.begin() + m, t.end());
  M->work.resize(n);
  *out = M;
  return RATO_OK;
}

extern "C" void rato_master_destroy(rato_master* M) { delete M; }

extern "C" int32_t rato_master_rows(const rato_master* M) { return M ? M->nrows : RATO_EINVAL; }

// k rows  A[j] . z <= b[j]  (A row-major k x n)
extern "C" int rato_master_add_rows(rato_master* M, int32_t k, const double* A, const double* b) {
  if (!M || k < 0 || (k > 0 && (!A || !b))) return RATO_EINVAL;
  const int n = M->n, m = M->m, nk = M->nk;
  for (int j = 0; j < k; ++j) {
    const double* a = A + (size_t)j * n;
    double* t = M->work.data();
    double ax0 = 0.0;
    for (int i = 0; i < n; ++i) {
      t[i] = a[i] * M->d[i];
      ax0 += a[i] * M->x0[i];
    }
    M->apply_Qt(t);                       // E = a N = t[m:]
    double ec = 0.0, nrm = 0.0;
    for (int i = 0; i < nk; ++i) {
      ec += t[m + i] * M->c[i];
      nrm += t[m + i] * t[m + i];
    }
    const double f = b[j] - ax0 + ec;
    const double s = fmax(sqrt(nrm), 1e-300);
    const size_t at = M->rows.size();
    M->rows.resize(at + nk + 1);
    for (int i = 0; i < nk; ++i) M->rows[at + i] = -(t[m + i] / s);
    M->rows[at + nk] = -(f / s);
    M->sc.push_back(s);
    M->passive.push_back(0);
    ++M->nrows;
  }
  return RATO_OK;
}

// -> 1: z (n) and the multipliers lam (one per row, >= 0) of the optimum;  0: the NNLS did not converge (warm and cold);
// RATO_EINFEASIBLE: the rows admit no point.
extern "C" int rato_master_solve(rato_master* M, double* z, double* lam) {
  if (!M || !z || (M->nrows > 0 && !lam)) return RATO_EINVAL;
  const int n = M->n, m = M->m, nk = M->nk, r = M->nrows;
  std::vector<double> v(nk, 0.0);
  if (r > 0) {
    constexpr int FACTOR_REFRESH = 96;
    // sigma sticks while |v| stays within a factor SIGMA_STICK of it.  Not wider: near convergence the cuts are nearly
    // parallel and differ mostly in h; a sigma 3x too large (left over from an earlier, larger |v|) shrank those differences
    // until the NNLS cycled between near-dependent columns (found on tests/test_gpu_scp.py's M = 40 case with a factor 4).
    // RATO_MASTER_STICK overrides (1: rescale whenever |v| moves, the behaviour up to round 5).
    static const double SIGMA_STICK = [] { const char* e = getenv("RATO_MASTER_STICK"); return e ? atof(e) : 1.5; }();
    double hmax = 0.0;
    for (int j = 0; j < r; ++j) hmax = fmax(hmax, M->rows[(size_t)j * (nk + 1) + nk]);
    const double want = fmax(1.0, fmax(M->vnorm, hmax));
    const int mm = nk + 1;
    auto rescale = [&](double sg) {   // every column changes: An is rebuilt, the factor with it
      M->sigma = sg;
      M->an_cols = 0;
      M->a1 = 0.0;
      M->factor_ready = false;
    };
    auto sync_columns = [&]() {       // the columns of the rows added since the last solve (all of them after a rescale)
      M->An.resize((size_t)mm * r);
      for (int j = M->an_cols; j < r; ++j) {
        const double* row = &M->rows[(size_t)j * mm];
        double* col = &M->An[(size_t)j * mm];
        double sj = 0.0;
        for (int i = 0; i < nk; ++i) {
          col[i] = row[i];
          sj += fabs(row[i]);
        }
        col[nk] = row[nk] / M->sigma;
        sj += fabs(col[nk]);
        if (sj > M->a1) M->a1 = sj;
      }
      M->an_cols = r;
    };
    if (!(M->sigma > 0.0) || want > SIGMA_STICK * M->sigma || want * SIGMA_STICK < M->sigma) rescale(want);
    sync_columns();
    if (M->changes > FACTOR_REFRESH) M->factor_ready = false;
    if (!M->factor_ready) M->changes = 0;
    M->bn.assign(mm, 0.0);
    M->bn[nk] = 1.0;
    M->y.assign(r, 0.0);
    int ok = rato_nnls::nnls_core(M->An.data(), mm, r, M->bn.data(), M->passive.data(), M->y.data(), 0, M->a1, M->qr,
                                  M->factor_ready, M->nn_s, M->nn_resid, M->nn_banned, &M->changes);
    if (ok != 1 && M->sigma != want) {   // a kept sigma was not good enough: the scale the previous rounds always used
      rescale(want);
      sync_columns();
      M->changes = 0;
      ok = rato_nnls::nnls_core(M->An.data(), mm, r, M->bn.data(), M->passive.data(), M->y.data(), 0, M->a1, M->qr, false,
                                M->nn_s, M->nn_resid, M->nn_banned, &M->changes);
    }
    if (ok != 1) {   // cold restart with a long leash (the NumPy version falls back to scipy's Lawson-Hanson here)
      for (int j = 0; j < r; ++j) M->passive[j] = 0;
      M->changes = 0;
      ok = rato_nnls::nnls_core(M->An.data(), mm, r, M->bn.data(), M->passive.data(), M->y.data(), 20 * r + 20, M->a1, M->qr,
                                false, M->nn_s, M->nn_resid, M->nn_banned, &M->changes);
      if (ok != 1) {
        M->factor_ready = false;
        return ok < 0 ? ok : 0;
      }
    }
    M->factor_ready = true;   // (qr = the factor of the passive columns of this solve: the next solve starts from it)
    std::vector<double>& res = M->nn_resid;
    res.assign(mm, 0.0);
    for (int j = 0; j < r; ++j) {
      const double yj = M->y[j];
      if (yj == 0.0) continue;
      const double* col = &M->An[(size_t)j * mm];
      for (int i = 0; i < mm; ++i) res[i] += col[i] * yj;
    }
    res[nk] -= 1.0;
    if (fabs(res[nk]) < 1e-14) return RATO_EINFEASIBLE;
    const double sigma = M->sigma;
    double vn = 0.0;
    for (int i = 0; i < nk; ++i) {
      v[i] = -sigma * res[i] / res[nk];
      vn += v[i] * v[i];
    }
    M->vnorm = sqrt(vn);
    for (int j = 0; j < r; ++j) lam[j] = sigma * (M->y[j] / (-res[nk])) / M->sc[j];
  }
  // z = x0 + N (v - c) = x0 + D^-1/2 Q [0; v - c]
  double* t = M->work.data();
  for (int i = 0; i < m; ++i) t[i] = 0.0;
  for (int i = 0; i < nk; ++i) t[m + i] = v[i] - M->c[i];
  M->apply_Q(t);
  for (int i = 0; i < n; ++i) z[i] = M->x0[i] + M->d[i] * t[i];
  return 1;
}
