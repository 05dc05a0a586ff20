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

#include "rato_nnls.h"
#include "rato_saa.h"

// A: m x n, COLUMN-major (column j at A + j*m); passive: n bytes, in = guess of the passive set, out = the passive set of
// the solution; y: n doubles out.  Returns 1 when the KKT test of the original algorithm holds (dual w = A'(b - A y) <=
// tol on the zero set, y > 0 on the passive set; tol = 10 max(m,n) eps |A|_1), 0 if maxiter steps were not enough.
extern "C" int rato_nnls_warm(const double* A, int32_t m, int32_t n, const double* b, uint8_t* passive, double* y,
                              int32_t maxiter) {
  if (!A || !b || !passive || !y || m <= 0 || n <= 0) return RATO_EINVAL;
  double a1 = 0.0;
  for (int j = 0; j < n; ++j) {
    double sj = 0.0;
    for (int i = 0; i < m; ++i) sj += fabs(A[(size_t)j * m + i]);
    if (sj > a1) a1 = sj;
  }
  rato_nnls::ThinQR qr;
  std::vector<double> s, resid;
  std::vector<uint8_t> banned;
  return rato_nnls::nnls_core(A, m, n, b, passive, y, maxiter, a1, qr, false, s, resid, banned, nullptr);
}
