// Host side of the boundary: the master QP of the cutting-plane loop (riskaversetrajopt_amd/dense_qp.py) is a
// least-distance problem solved through non-negative least squares.  The reference hands its subproblem to OSQP, a C
// library (drone_risk.py:433-457); this is the corresponding native piece here -- HOST code only (no kernel, no device
// memory), compiled into librato_saa.so so that it travels with the C ABI.
//
// Lawson & Hanson, "Solving Least Squares Problems", ch. 23:   min |A y - b|,  y >= 0,   started from a guess of the
// passive (positive) set -- a cutting-plane loop solves a sequence of problems that differ by one column, and the
// optimal passive set moves by a column or two.  The least-squares problems on the passive columns are solved through
// a thin QR factorisation (modified Gram-Schmidt with one re-orthogonalisation) that is UPDATED when a column enters
// (O(m k)) and DOWNDATED with Givens rotations when one leaves (O(m k)), instead of being recomputed (O(m k^2)) per
// step as the NumPy version did (numpy.linalg.lstsq, ~18 us of call overhead each, 4-5 calls per master solve: 43 % of
// the SCP wall-clock at M = 1e5).  Same iterates, same termination test.
#include <math.h>
#include <stdint.h>
#include <string.h>

#include <vector>

#include "rato_saa.h"

namespace {

struct ThinQR {
  int m, k = 0;
  std::vector<double> Q;    // m x kmax, column-major, orthonormal columns
  std::vector<double> R;    // kmax x kmax, column-major, upper triangular
  std::vector<double> qtb;  // Q' b
  std::vector<int> col;     // original column index of each factor column
  int kmax;
  ThinQR(int m_, int kmax_) : m(m_), Q((size_t)m_ * kmax_), R((size_t)kmax_ * kmax_), qtb(kmax_), col(kmax_), kmax(kmax_) {}

  // append column a (length m); false if it is numerically dependent on the columns already there
  bool add(const double* a, const double* b, int index) {
    if (k == kmax) return false;
    std::vector<double> v(a, a + m);
    double* r = &R[(size_t)k * kmax];
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
      memcpy(&R[(size_t)j * kmax], &R[(size_t)(j + 1) * kmax], sizeof(double) * kmax);
      col[j] = col[j + 1];
    }
    --k;
    for (int i = pos; i < k; ++i) {   // zero R[i+1, i] with a rotation of rows i, i+1
      double& a = R[(size_t)i * kmax + i];
      double& bb = R[(size_t)i * kmax + i + 1];
      const double h = hypot(a, bb);
      if (h == 0.0) continue;
      const double c = a / h, s = bb / h;
      a = h;
      bb = 0.0;
      for (int j = i + 1; j < k; ++j) {
        double& x = R[(size_t)j * kmax + i];
        double& y = R[(size_t)j * kmax + i + 1];
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
      for (int j = i + 1; j < k; ++j) acc -= R[(size_t)j * kmax + i] * s[j];
      s[i] = acc / R[(size_t)i * kmax + i];
    }
  }
};

}  // namespace

// A: m x n, COLUMN-major (column j at A + j*m); passive: n bytes, in = guess of the passive set, out = the passive set of
// the solution; y: n doubles out.  Returns 1 when the KKT test of the original algorithm holds (dual w = A'(b - A y) <=
// tol on the zero set, y > 0 on the passive set; tol = 10 max(m,n) eps |A|_1), 0 if maxiter steps were not enough.
extern "C" int rato_nnls_warm(const double* A, int32_t m, int32_t n, const double* b, uint8_t* passive, double* y,
                              int32_t maxiter) {
  if (!A || !b || !passive || !y || m <= 0 || n <= 0) return RATO_EINVAL;
  if (maxiter <= 0) maxiter = 3 * n + 10;
  double a1 = 0.0;
  for (int j = 0; j < n; ++j) {
    double sj = 0.0;
    for (int i = 0; i < m; ++i) sj += fabs(A[(size_t)j * m + i]);
    if (sj > a1) a1 = sj;
  }
  const double tol = 10.0 * (double)(m > n ? m : n) * 2.220446049250313e-16 * (a1 > 0.0 ? a1 : 1e-300);
  const int kmax = m < n ? m : n;
  ThinQR qr(m, kmax);
  std::vector<double> s(kmax), resid(m), w(n);
  std::vector<uint8_t> banned(n, 0);   // columns found dependent on the passive set in this step
  for (int j = 0; j < n; ++j) y[j] = 0.0;
  // warm start: keep the part of the guess whose least-squares solution is positive
  for (int j = 0; j < n; ++j)
    if (passive[j] && !qr.add(A + (size_t)j * m, b, j)) passive[j] = 0;
  for (int guard = 0; guard <= n && qr.k > 0; ++guard) {
    qr.solve(s.data());
    bool all_pos = true;
    for (int i = qr.k - 1; i >= 0; --i)
      if (!(s[i] > 0.0)) {
        all_pos = false;
        passive[qr.col[i]] = 0;
        qr.remove(i);
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
          dropped = true;
        }
      if (!dropped) {   // rounding: force progress
        for (int i = qr.k - 1; i >= 0; --i)
          if (!(s[i] > 0.0)) {
            y[qr.col[i]] = 0.0;
            passive[qr.col[i]] = 0;
            qr.remove(i);
          }
      }
      if (qr.k == 0) break;
    }
    for (int j = 0; j < n; ++j) banned[j] = 0;   // the passive set changed: dependencies may have, too
  }
  return 0;
}
