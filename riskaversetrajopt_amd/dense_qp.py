"""Exact dense solver for the small strictly convex master QP of the cutting-plane loop.

    min 1/2 x' P x + q' x   s.t.  A_eq x = b_eq,   A_in x <= b_in          (P symmetric positive definite)

Method (Lawson & Hanson, "Solving Least Squares Problems", ch. 23): eliminate the equalities with a
null-space basis, whiten with the Cholesky factor of the reduced Hessian so that the objective becomes
1/2 |v|^2, and solve the resulting least-distance problem  min |v| s.t. G v >= h  through one
non-negative least squares (scipy.optimize.nnls, an active-set method: exact up to rounding, no
tolerances to tune, ~ms at the sizes used here: <= 3S+1 variables, a few hundred rows).
"""
import numpy as np
import scipy.linalg as sla
from scipy.optimize import nnls


class InfeasibleError(RuntimeError):
    pass


def solve(P, q, A_eq, b_eq, A_in, b_in):
    """-> (x, lam) with lam >= 0 the multipliers of the inequality rows."""
    P = np.asarray(P, dtype=np.float64)
    q = np.asarray(q, dtype=np.float64)
    n = q.shape[0]
    A_in = np.asarray(A_in, dtype=np.float64).reshape(-1, n)
    b_in = np.asarray(b_in, dtype=np.float64)
    if A_eq is not None and len(A_eq):
        A_eq = np.asarray(A_eq, dtype=np.float64)
        x0 = np.linalg.lstsq(A_eq, np.asarray(b_eq, dtype=np.float64), rcond=None)[0]
        N = sla.null_space(A_eq)
    else:
        x0, N = np.zeros(n), np.eye(n)
    H = N.T @ P @ N
    c = N.T @ (P @ x0 + q)
    L = np.linalg.cholesky(H)                                   # H = L L'
    Linv_c = sla.solve_triangular(L, c, lower=True)
    # w = L^-T (v - L^-1 c);  rows: A_in (x0 + N w) <= b_in  ->  E v <= f
    E = sla.solve_triangular(L, (A_in @ N).T, lower=True).T     # A_in N L^-T
    f = b_in - A_in @ x0 + E @ Linv_c
    if E.shape[0] == 0:
        v = np.zeros(N.shape[1])
        lam = np.zeros(0)
    else:
        # scale rows (helps nnls): (E_i / s_i) v <= f_i / s_i
        sc = np.maximum(np.linalg.norm(E, axis=1), 1e-300)
        G, h = -(E / sc[:, None]), -(f / sc)
        A_n = np.vstack([G.T, h[None, :]])
        b_n = np.zeros(A_n.shape[0])
        b_n[-1] = 1.0
        y, _ = nnls(A_n, b_n, maxiter=20 * A_n.shape[1])
        r = A_n @ y - b_n
        if abs(r[-1]) < 1e-14:
            raise InfeasibleError("master QP infeasible")
        v = -r[:-1] / r[-1]
        lam = (y / (-r[-1])) / sc
    w = sla.solve_triangular(L.T, v - Linv_c, lower=False)
    return x0 + N @ w, lam


class Master:
    """The same solve with the equality elimination and the Cholesky whitening done ONCE, and inequality rows
    appended incrementally (a cutting-plane loop adds one row per iteration)."""

    def __init__(self, P, q, A_eq, b_eq):
        P = np.asarray(P, dtype=np.float64)
        q = np.asarray(q, dtype=np.float64)
        n = q.shape[0]
        if A_eq is not None and len(A_eq):
            A_eq = np.asarray(A_eq, dtype=np.float64)
            self.x0 = np.linalg.lstsq(A_eq, np.asarray(b_eq, dtype=np.float64), rcond=None)[0]
            self.N = sla.null_space(A_eq)
        else:
            self.x0, self.N = np.zeros(n), np.eye(n)
        H = self.N.T @ P @ self.N
        c = self.N.T @ (P @ self.x0 + q)
        self.L = np.linalg.cholesky(H)
        self.Linv_c = sla.solve_triangular(self.L, c, lower=True)
        self.NLinvT = sla.solve_triangular(self.L, self.N.T, lower=True).T      # N L^-T  (n x k)
        self.G = np.zeros((0, self.N.shape[1]))
        self.h = np.zeros(0)
        self.sc = np.zeros(0)

    def add_rows(self, A_in, b_in):
        A_in = np.atleast_2d(np.asarray(A_in, dtype=np.float64))
        b_in = np.atleast_1d(np.asarray(b_in, dtype=np.float64))
        E = A_in @ self.NLinvT
        f = b_in - A_in @ self.x0 + E @ self.Linv_c
        sc = np.maximum(np.linalg.norm(E, axis=1), 1e-300)
        self.G = np.vstack([self.G, -(E / sc[:, None])])
        self.h = np.concatenate([self.h, -(f / sc)])
        self.sc = np.concatenate([self.sc, sc])

    def solve(self):
        if self.G.shape[0] == 0:
            v = np.zeros(self.N.shape[1])
            lam = np.zeros(0)
        else:
            A_n = np.vstack([self.G.T, self.h[None, :]])
            b_n = np.zeros(A_n.shape[0])
            b_n[-1] = 1.0
            y, _ = nnls(A_n, b_n, maxiter=20 * A_n.shape[1])
            r = A_n @ y - b_n
            if abs(r[-1]) < 1e-14:
                raise InfeasibleError("master QP infeasible")
            v = -r[:-1] / r[-1]
            lam = (y / (-r[-1])) / self.sc
        return self.x0 + self.NLinvT @ (v - self.Linv_c), lam
