// Host side of the boundary: Lawson-Hanson NNLS on a thin QR factorisation that is updated when a column enters and
// downdated when one leaves -- shared by rato_nnls_warm (nnls.hip: one problem per call, the factor built from the guess
// of the passive set) and by the master QP of the cutting-plane loop (master.hip: a sequence of problems that differ by
// the columns appended between two solves -- the factor of the previous solve's passive set is KEPT, round 6: rebuilding
// it column by column was most of a master solve).  HOST code only.
#ifndef RATO_NNLS_H
#define RATO_NNLS_H

#include <math.h>
#include <stdint.h>
#include <string.h>

#include <vector>

namespace rato_nnls {

// Thin QR of the passive columns: Q (m x k, orthonormal columns), R (k x k upper triangular, leading dimension m), Q'b.
// Grows by columns (no fixed capacity beyond k <= m).
struct ThinQR {
  int m = 0, k = 0;
  std::vector<double> Q;    // column j at Q[j*m .. j*m+m)
  std::vector<double> R;    // column j at R[j*m .. j*m+m): entries 0..j used
  std::vector<double> qtb;  // Q' b
  std::vector<int> col;     // original column index of each factor column
  std::vector<double> v;    // work vector

  void reset(int m_) {
    m = m_;
    k = 0;
    v.resize((size_t)m_);
  }
  void reserve_cols(int cols) {
    if ((int)col.size() >= cols) return;
    Q.resize((size_t)m * cols);
    R.resize((size_t)m * cols);
    qtb.resize(cols);
    col.resize(cols);
  }

  // append column a (length m); false if it is numerically dependent on the columns already there
  bool add(const double* a, const double* b, int index) {
    if (k == m) return false;
    if ((int)col.size() < k + 1) reserve_cols(k < 16 ? 32 : 2 * k);
    memcpy(v.data(), a, sizeof(double) * (size_t)m);
    double* r = &R[(size_t)k * m];
    for (int i = 0; i < k; ++i) r[i] = 0.0;
    double norm0 = 0.0;
    for (int i = 0; i < m; ++i) norm0 += a[i] * a[i];
    norm0 = sqrt(norm0);
    for (int pass = 0; pass < 2; ++pass) {   // Gram-Schmidt, twice is enough
      for (int j = 0; j < k; ++j) {
        const double* q = &Q[(size_t)j * m];
        double d = 0.0;
        for (int i = 0; i < m; ++i) d += q[i] * v[i];
        for (int i = 0; i < m; ++i) v[i] -= d * q[i];
        r[j] += d;
      }
    }
    double nrm = 0.0;
    for (int i = 0; i < m; ++i) nrm += v[i] * v[i];
    nrm = sqrt(nrm);
    if (!(nrm > 1e-12 * (norm0 > 0.0 ? norm0 : 1.0))) return false;
    double* q = &Q[(size_t)k * m];
    double d = 0.0;
    for (int i = 0; i < m; ++i) {
      q[i] = v[i] / nrm;
      d += q[i] * b[i];
    }
    r[k] = nrm;
    qtb[k] = d;
    col[k] = index;
    ++k;
    return true;
  }

  // delete factor column `pos`: R loses a column (upper Hessenberg from there), Givens rotations restore the triangle
  void remove(int pos) {
    for (int j = pos; j + 1 < k; ++j) {   // shift columns of R left
      memcpy(&R[(size_t)j * m], &R[(size_t)(j + 1) * m], sizeof(double) * (size_t)(j + 2));
      col[j] = col[j + 1];
    }
    --k;
    for (int i = pos; i < k; ++i) {   // zero R[i+1, i] with a rotation of rows i, i+1
      double& a = R[(size_t)i * m + i];
      double& bb = R[(size_t)i * m + i + 1];
      const double h = hypot(a, bb);
      if (h == 0.0) continue;
      const double c = a / h, s = bb / h;
      a = h;
      bb = 0.0;
      for (int j = i + 1; j < k; ++j) {
        double& x = R[(size_t)j * m + i];
        double& y = R[(size_t)j * m + i + 1];
        const double nx = c * x + s * y, ny = -s * x + c * y;
        x = nx;
        y = ny;
      }
      double* q0 = &Q[(size_t)i * m];
      double* q1 = &Q[(size_t)(i + 1) * m];
      for (int t = 0; t < m; ++t) {
        const double nx = c * q0[t] + s * q1[t], ny = -s * q0[t] + c * q1[t];
        q0[t] = nx;
        q1[t] = ny;
      }
      const double nb = c * qtb[i] + s * qtb[i + 1], nb1 = -s * qtb[i] + c * qtb[i + 1];
      qtb[i] = nb;
      qtb[i + 1] = nb1;
    }
  }

  void solve(double* s) const {   // R s = Q' b
    for (int i = k - 1; i >= 0; --i) {
      double acc = qtb[i];
      for (int j = i + 1; j < k; ++j) acc -= R[(size_t)j * m + i] * s[j];
      s[i] = acc / R[(size_t)i * m + i];
    }
  }
};

// min |A y - b|, y >= 0  (A: m x n, column j at A + j*m).  `qr` holds the factor of the columns flagged in `passive` when
// `factor_ready` (the previous solve of a growing problem left it there: same columns, same b); otherwise it is built
// from the flagged columns first.  a1 = max_j |A_j|_1 (the tolerance of the KKT test: 10 max(m,n) eps a1).
// Returns 1 when the KKT test of the original algorithm holds, 0 if maxiter steps were not enough.  On return `qr` is the
// factor of the final passive set and y the solution.
inline int nnls_core(const double* A, int m, int n, const double* b, uint8_t* passive, double* y, int maxiter, double a1,
                     ThinQR& qr, bool factor_ready, std::vector<double>& s, std::vector<double>& resid,
                     std::vector<uint8_t>& banned, int* changes) {
  if (maxiter <= 0) maxiter = 3 * n + 10;
  const double tol = 10.0 * (double)(m > n ? m : n) * 2.220446049250313e-16 * (a1 > 0.0 ? a1 : 1e-300);
  const int kmax = m < n ? m : n;
  s.resize((size_t)kmax + 1);
  resid.resize((size_t)m);
  banned.assign((size_t)n, 0);
  for (int j = 0; j < n; ++j) y[j] = 0.0;
  if (!factor_ready) {
    qr.reset(m);
    for (int j = 0; j < n; ++j)
      if (passive[j] && !qr.add(A + (size_t)j * m, b, j)) passive[j] = 0;
  }
  // warm start: keep the part of the guess whose least-squares solution is positive
  for (int guard = 0; guard <= n && qr.k > 0; ++guard) {
    qr.solve(s.data());
    bool all_pos = true;
    for (int i = qr.k - 1; i >= 0; --i)
      if (!(s[i] > 0.0)) {
        all_pos = false;
        passive[qr.col[i]] = 0;
        qr.remove(i);
        if (changes) ++*changes;
      }
    if (all_pos) {
      for (int i = 0; i < qr.k; ++i) y[qr.col[i]] = s[i];
      break;
    }
  }
  for (int j = 0; j < n; ++j) passive[j] = 0;
  for (int i = 0; i < qr.k; ++i) passive[qr.col[i]] = 1;
  if (qr.k == 0)
    for (int j = 0; j < n; ++j) y[j] = 0.0;

  for (int it = 0; it < maxiter; ++it) {
    for (int i = 0; i < m; ++i) resid[i] = b[i];
    for (int j = 0; j < n; ++j)
      if (y[j] != 0.0) {
        const double* a = A + (size_t)j * m;
        for (int i = 0; i < m; ++i) resid[i] -= a[i] * y[j];
      }
    int jbest = -1;
    double wbest = tol;
    for (int j = 0; j < n; ++j) {
      if (passive[j] || banned[j]) continue;
      const double* a = A + (size_t)j * m;
      double d = 0.0;
      for (int i = 0; i < m; ++i) d += a[i] * resid[i];
      if (d > wbest) {
        wbest = d;
        jbest = j;
      }
    }
    if (jbest < 0) return 1;
    if (!qr.add(A + (size_t)jbest * m, b, jbest)) {   // dependent column: it cannot improve the fit
      banned[jbest] = 1;
      continue;
    }
    if (changes) ++*changes;
    passive[jbest] = 1;
    for (int inner = 0; inner <= n; ++inner) {
      qr.solve(s.data());
      bool all_pos = true;
      for (int i = 0; i < qr.k; ++i)
        if (!(s[i] > 0.0)) all_pos = false;
      if (all_pos) {
        for (int j = 0; j < n; ++j) y[j] = 0.0;
        for (int i = 0; i < qr.k; ++i) y[qr.col[i]] = s[i];
        break;
      }
      double alpha = INFINITY;
      for (int i = 0; i < qr.k; ++i)
        if (!(s[i] > 0.0)) {
          const double yi = y[qr.col[i]], den = yi - s[i];
          const double a = den > 0.0 ? yi / den : 0.0;
          if (a < alpha) alpha = a;
        }
      if (!(alpha < INFINITY)) alpha = 0.0;
      bool dropped = false;
      for (int i = 0; i < qr.k; ++i) y[qr.col[i]] += alpha * (s[i] - y[qr.col[i]]);
      for (int i = qr.k - 1; i >= 0; --i)
        if (!(s[i] > 0.0) && !(y[qr.col[i]] > 1e-300)) {   // the variables that hit zero leave the passive set
          y[qr.col[i]] = 0.0;
          passive[qr.col[i]] = 0;
          qr.remove(i);
          if (changes) ++*changes;
          dropped = true;
        }
      if (!dropped) {   // rounding: force progress
        for (int i = qr.k - 1; i >= 0; --i)
          if (!(s[i] > 0.0)) {
            y[qr.col[i]] = 0.0;
            passive[qr.col[i]] = 0;
            qr.remove(i);
            if (changes) ++*changes;
          }
      }
      if (qr.k == 0) break;
    }
    for (int j = 0; j < n; ++j) banned[j] = 0;   // the passive set changed: dependencies may have, too
  }
  return 0;
}

}  // namespace rato_nnls

#endif  // RATO_NNLS_H
